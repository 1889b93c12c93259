# Measurement aid (GPU box): start / end of every workgroup of seg_reduce_chunk_slab_kernel inside the replayed configs[1] step, by role.
# Needs a library built with TT_EXTRA_HIPCC_FLAGS=-DTT_SEG_STAMPS (the stamps are compiled out of the shipped library).
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
fn = lib.tt_debug_seg_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_void_p]
buf = np.zeros(4096 * 4, dtype=np.uint64)
assert fn(0, buf.ctypes.data) == 0
s = buf.reshape(4096, 4).astype(np.int64)
s = s[s[:, 2] > 0]
t0 = s[:, 0].min()
print(f"seg_reduce_chunk_slab_kernel: {len(s)} workgroups stamped; span {(s[:, 1].max() - t0) / 100:.2f} us")
for role, name in ((1, "slab reduction (weight gradients)"), (2, "rows (one 8-lane group per distinct row)"), (3, "chunks of long rows")):
    r = s[s[:, 2] == role]
    if len(r) == 0:
        continue
    st, en = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0
    d = en - st
    print(f"  {name:42s} n = {len(r):5d}  start mean {st.mean():5.2f} p90 {np.percentile(st, 90):5.2f} max {st.max():5.2f} | "
          f"lifetime mean {d.mean():5.2f} p90 {np.percentile(d, 90):5.2f} max {d.max():5.2f} | end mean {en.mean():5.2f} max {en.max():5.2f} us")
full = buf.reshape(4096, 4).astype(np.int64)
ch = s[s[:, 2] == 3]
dch = (ch[:, 1] - ch[:, 0]) / 100.0
print(f"  chunk workgroups with work (> 2 us): {(dch > 2).sum()}, their lifetime mean {dch[dch > 2].mean() if (dch > 2).any() else 0:.2f} us")
for role in (2, 3):
    idx = np.nonzero(full[:, 2] == role)[0]
    r = full[idx]
    d = (r[:, 1] - r[:, 0]) / 100.0
    top = np.argsort(-d)[:6]
    print(f"  role {role}: longest workgroups (index within role, start, lifetime us):", [(int(idx[k] - idx[0]), round(float((r[k, 0] - t0) / 100), 2), round(float(d[k]), 2)) for k in top])
    print(f"          lifetime histogram (us, 0-1-2-3-4-6-8-12):", np.histogram(d, bins=[0, 1, 2, 3, 4, 6, 8, 12, 100])[0].tolist())
rows = s[s[:, 2] == 2]
st = np.sort((rows[:, 0] - t0) / 100.0)
print("  rows: start quantiles (10/50/90/99 %):", [round(float(np.percentile(st, q)), 2) for q in (10, 50, 90, 99)])
leg.close()
