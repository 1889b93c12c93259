# Measurement aid: what the eager hand-over launch in front of every replay costs beyond its own kernel time.
# Times (a) step(batch) = hand-over launch + replay, (b) the replay alone, (c) the hand-over launch alone, each back to back on
# bench.py's configs[1] task.  Run on the GPU box from the repository root.
import sys, time
sys.path.insert(0, '.')
import torch
import bench
args = bench.parse(["--no-cpu-baseline", "--no-h2d", "--steps", "20", "--warmup", "5", "--no-lookup-profile"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ctx = dict(dev=dev, world=1, rank=0, staged=False, comm=None, fence=torch.cuda.synchronize, max_over_ranks=lambda x: x)
leg = bench.Leg(args, ctx, 8192, 1_000_000, 1_000_000, False, negatives="local", sync_bn=False)
leg.run()
gs = leg.gstep
N = 300
def timed(fn):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(N): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / N * 1e6
i_ = [0]
a = timed(lambda i=0: gs.step(leg.pool[i % len(leg.pool)]))
b = timed(lambda i=0: gs._replay())
c = timed(lambda i=0: (gs._run_ingest([gs._fill_slot()], None), gs._mark_slot()))
print(f"hand-over + replay {a:.1f} us   replay alone {b:.1f} us   hand-over launch alone {c:.1f} us   (a - b - c = {a - b - c:.1f} us)")
leg.close()
