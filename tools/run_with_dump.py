#!/usr/bin/env python3
"""Run a script with a traceback dump (and exit) after N seconds: finds where a run hangs.  usage: run_with_dump.py SECONDS script.py [args...]"""
import faulthandler, runpy, sys
secs = float(sys.argv[1])
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
