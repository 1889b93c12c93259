# Measurement aid (GPU box): phase anatomy of tail_fwd_kernel's workgroups inside the replayed configs[1] step.  Needs a library
# built with TT_EXTRA_HIPCC_FLAGS=-DTT_TAIL_STAMPS (the stamps are compiled out of the shipped library).
import ctypes, sys
sys.path.insert(0, ".")
import numpy as np
import torch
import bench
from jodalrob_twotower_amd import _lib
args = bench.parse(["--no-cpu-baseline", "--no-h2d", "--steps", "30", "--warmup", "10", "--no-lookup-profile", "--no-extra-legs"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ctx = dict(dev=dev, world=1, rank=0, staged=False, comm=None, fence=torch.cuda.synchronize, max_over_ranks=lambda x: x)
leg = bench.Leg(args, ctx, 8192, 1_000_000, 1_000_000, False)
leg.run()
torch.cuda.synchronize()
lib = _lib.load()
fn = lib.tt_debug_tail_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_void_p]
buf = np.zeros(512 * 8, dtype=np.uint64)
assert fn(0, buf.ctypes.data) == 0
s = buf.reshape(512, 8).astype(np.int64)
live = s[:, 5] > 0
s = s[live]
t0 = s[:, 0].min()
print(f"tail_fwd_kernel: {int(live.sum())} stamped workgroups; span {(s[:, 5].max() - t0) / 100:.2f} us")
names = ["start", "loads + sub-chains + W_out -> LDS", "statistics finalised", "BN apply + dropout + act stored", "output MFMA", "normalise + stores (emb, y, operand images)"]
for i in range(1, 6):
    d = (s[:, i] - s[:, i - 1]) / 100.0
    print(f"  phase {i} {names[i]:45s} mean {d.mean():6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
st = (s[:, 0] - t0) / 100.0
print(f"  workgroup start: mean {st.mean():.2f} p90 {np.percentile(st, 90):.2f} max {st.max():.2f} us;  lifetime mean {((s[:, 5] - s[:, 0]) / 100).mean():.2f} us")
leg.close()
