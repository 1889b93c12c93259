"""python tools/efence/run_script.py script.py [args]: run a Python script under the electric-fence allocator (serialised launches)."""
import os, runpy, subprocess, sys
from pathlib import Path
here = Path(__file__).resolve().parent
so = here / "libefence.so"
if not so.exists():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-fPIC", "-shared", "-o", str(so), str(here / "efence.cpp")])
os.environ.setdefault("HIP_LAUNCH_BLOCKING", "1")
os.environ.setdefault("AMD_SERIALIZE_KERNEL", "3")
import faulthandler
faulthandler.enable()
import torch
torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(str(so), "ef_malloc", "ef_free"))
sys.argv = sys.argv[1:]
sys.path.insert(0, os.getcwd())
runpy.run_path(sys.argv[0], run_name="__main__")
