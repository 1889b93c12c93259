# Debugging aid (round 3, fp8 gradient products): per-row error of the fp8 score backward against its oracle model.  Run from the repository root on the GPU box.
import sys, numpy as np, torch
sys.path.insert(0, '.')
import importlib
from oracle import oracle_np as O
import jodalrob_twotower_amd as m
from jodalrob_twotower_amd import ops, _lib as L
from jodalrob_twotower_amd.two_tower_train_task import _ScoreCEFn
B, D, T = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rng = np.random.default_rng(B * 3 + D)
n = rng.standard_normal((B, D)).astype(np.float32)
c = (0.5 * n + rng.standard_normal((B, D))).astype(np.float32)
n /= np.linalg.norm(n, axis=1, keepdims=True); c /= np.linalg.norm(c, axis=1, keepdims=True)
tn, tc = torch.from_numpy(n).cuda(), torch.from_numpy(c).cuda()
n8, c8 = O.score_operands_fp8(n.astype(np.float64), c.astype(np.float64), T)
nb, cb = O.score_operands_bf16(n.astype(np.float64), c.astype(np.float64), T)
_, _, S, lse = O.score_ce_fwd(n8, c8, T)
a, b = tn.clone().requires_grad_(True), tc.clone().requires_grad_(True)
l2, _, _ = _ScoreCEFn.apply(a, b, 1.0 / T, "fp8", False, False)
l2.backward()
rN, rC = O.score_ce_bwd(n8, c8, S, lse, T, prod_operands=(nb, cb), block_fp8=True)
for name, g, r in (("dN", a.grad.cpu().numpy(), rN), ("dC", b.grad.cpu().numpy(), rC)):
    err = np.linalg.norm(g - r, axis=1) / np.linalg.norm(r, axis=1)
    bad = np.argsort(-err)[:12]
    print(name, "total", np.linalg.norm(g - r) / np.linalg.norm(r), "rows>1e-3:", int((err > 1e-3).sum()), "worst", [(int(i), float(f"{err[i]:.2e}")) for i in bad])
    e2 = np.abs(g - r).max(axis=0)
    print("   per-column max err (first 8, then blocks of 32 max):", e2[:8], [float(f"{e2[i:i+32].max():.2e}") for i in range(0, D, 32)])
# which pair of b rows explains dN's error?  regress the error of every row on the oracle's per-pair contributions
W = np.exp(S - lse[0][:, None]) + np.exp(S - lse[1][None, :])
W[np.arange(B), np.arange(B)] = 0
Q = O.q_block_e4m3(W, 1)
g = a.grad.cpu().numpy().astype(np.float64) * (2 * B) * T
r = rN * (2 * B) * T
err = g - r
for p in range((B + 63) // 64):
    G = Q[:, 64 * p:64 * p + 64] @ c8[64 * p:64 * p + 64]
    G2 = W[:, 64 * p:64 * p + 64] @ c8[64 * p:64 * p + 64]
    al = (err * G).sum(1) / (G * G).sum(1)
    print("pair", p, "alpha median %.3f mean %.3f" % (np.median(al), al.mean()), "|err| explained:", float(np.linalg.norm(err - al[:, None] * G) / np.linalg.norm(err)),
          " quantisation noise of this pair vs err:", float(np.linalg.norm(G - G2) / np.linalg.norm(err)))
be = (err * cb).sum(1) / (cb * cb).sum(1)
print("diag-row regression: beta median %.4f min %.4f max %.4f" % (np.median(be), be.min(), be.max()), " residual:", float(np.linalg.norm(err - be[:, None] * cb) / np.linalg.norm(err)))
wdiag = np.exp(np.diagonal(S) - lse[0]) + np.exp(np.diagonal(S) - lse[1])
print("  beta / w_aa: median %.3f" % np.median(be / wdiag), " first rows:", (be / wdiag)[:8], " last rows:", (be / wdiag)[-8:])
if D >= 256:
    worst = int(np.argmax(np.linalg.norm(err, axis=1)))
    x = err[worst] @ np.linalg.pinv(c8)
    top = np.argsort(-np.abs(x))[:40]
    print("row", worst, "min-norm dW over b, top 40:", sorted((int(bb), float(f"{x[bb] / W[worst, bb]:.2f}")) for bb in top))
