"""Run (a selection of) the GPU tests under the electric-fence allocator: python tools/efence/run_pytest.py [pytest args].
Builds tools/efence/libefence.so on first use.  Graph-capturing tests are not supported by pluggable allocators: deselect them."""
import os, subprocess, sys
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
alloc = torch.cuda.memory.CUDAPluggableAllocator(str(so), "ef_malloc", "ef_free")
torch.cuda.memory.change_current_allocator(alloc)
import pytest
sys.exit(pytest.main(sys.argv[1:]))
