"""Per-workgroup anatomy of the lookup launch from bench.py's stamp ring: python bench.py --lookup-wg-dump path.npy ...; then
python tools/lookup_wg.py path.npy"""
import sys, numpy as np
r = np.load(sys.argv[1]).astype(np.int64)
MAX_WG = 4096
n = (len(r) - MAX_WG) // (MAX_WG * 2)
launches = int(r[0])
pairs = r[MAX_WG:].reshape(n, MAX_WG, 2)
for k in range(max(0, launches - 3), launches):
    blk = pairs[k % n]
    live = blk[:, 1] > 0
    st, en = blk[live, 0], blk[live, 1]
    t0 = st.min()
    dur = (en - st) / 100.0
    print(f"launch {k}: workgroups {int(live.sum())} span {(en.max() - t0) / 100:.2f} us | start: mean {(st.mean() - t0) / 100:.2f} p90 {(np.percentile(st, 90) - t0) / 100:.2f} max {(st.max() - t0) / 100:.2f}"
          f" | duration: mean {dur.mean():.2f} p10 {np.percentile(dur, 10):.2f} p90 {np.percentile(dur, 90):.2f} max {dur.max():.2f} | end: p50 {(np.percentile(en, 50) - t0) / 100:.2f} p90 {(np.percentile(en, 90) - t0) / 100:.2f}")
