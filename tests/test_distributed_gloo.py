"""world_size-2 gloo tests (CPU) of the multi-GPU routing logic: row-wise sharding, bucket-by-owner,
the three all-to-alls (ids, pooled rows, row gradients) and the un-permute, checked against a direct
gather / scatter-add on the unsharded table.  The HIP compute steps are replaced by a checker backend
defined HERE (tests may use CPU code as the checker; the product path has no CPU fallback)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class CheckerBackend:
    """Same interface as jodalrob_twotower_amd.distributed.HipBackend, plain torch/numpy on CPU."""

    def __init__(self):
        self.dense_grad = None

    def global_rows(self, sides, B, E, table_rows):
        out = []
        for s in sides:
            ids = s.ids.view(B, s.K)
            cl = torch.minimum(torch.clamp(ids, min=0), s.key_vocab[None, :] - 1)
            out.append((cl + s.key_row_offset[None, :]).reshape(-1))
        return torch.cat(out).to(torch.int32)

    def bucket_by_owner(self, rows, world):
        owners = torch.remainder(rows, world)
        order = torch.sort(owners, stable=True).indices.to(torch.int32)
        return order, torch.bincount(owners.long(), minlength=world)

    def owner_lookup(self, weight, local_ids, want_plan):
        return weight[local_ids].clone(), (local_ids.clone() if want_plan else None)

    def place_rows(self, pooled, inv, sides, B):
        base = 0
        E = pooled.shape[1]
        for s in sides:
            n = B * s.K
            s.out.copy_(pooled[inv[base:base + n]].view(B, s.K * E))
            base += n

    def collect_grads(self, srcs, order, B, E):
        flat = torch.cat([d.reshape(B * K, E) for d, K in srcs])
        return flat[order.long()]

    def owner_accumulate(self, store, plan, d_rows, max_per_row=0):
        g = torch.zeros((store.weight.shape[0] + 1, store.weight.shape[1]))      # + the pad row of the fixed-capacity exchange
        g.index_add_(0, plan.long(), d_rows)
        self.dense_grad = g[:-1]

    # ---- fixed-capacity exchange (PaddedRowExchange): same contracts as HipBackend's steps, plain numpy ----
    def local_plan(self, rows, side_K, B, table_rows=0):
        from types import SimpleNamespace
        r = rows.numpy()
        order = np.argsort(r, kind="stable").astype(np.int32)
        sr = r[order]
        heads = np.flatnonzero(np.concatenate([[True], sr[1:] != sr[:-1]])) if len(sr) else np.zeros(0, np.int64)
        M, U = len(r), len(heads)
        uniq = np.full(M, -1, np.int32); uniq[:U] = sr[heads]
        seg = np.full(M + 1, -1, np.int32); seg[:U] = heads; seg[U] = M
        return SimpleNamespace(sorted_src=torch.from_numpy(order), unique_rows=torch.from_numpy(uniq), seg_offsets=torch.from_numpy(seg),
                               n_unique=torch.tensor([U], dtype=torch.int32), M=M)

    def new_flag(self, device):
        return torch.zeros(1, dtype=torch.int32)

    def route_bucket(self, plan, G, C, pad_id, pad_u, overflow):
        U = int(plan.n_unique)
        send_ids = np.repeat(np.asarray(pad_id, np.int32), C)
        send_u = np.full(G * C, pad_u, np.int32)
        pos_u = np.full(plan.M, G * C, np.int32)            # rows that do not fit point at the zero row behind the buckets
        counts = np.zeros(G, np.int32)
        for u in range(U):
            row = int(plan.unique_rows[u]); g = row % G
            p = counts[g]; counts[g] += 1
            if p < C:
                send_ids[g * C + p] = row // G; send_u[g * C + p] = u; pos_u[u] = g * C + p
        if (counts > C).any():
            overflow[0] = 1
        return tuple(torch.from_numpy(a) for a in (send_ids, send_u, pos_u, counts))

    def route_expand(self, plan, pos_u):
        idx = torch.zeros(plan.M, dtype=torch.int64)
        U = int(plan.n_unique)
        for u in range(U):
            for p in range(int(plan.seg_offsets[u]), int(plan.seg_offsets[u + 1])):
                idx[int(plan.sorted_src[p])] = int(pos_u[u])
        return idx

    def gather_rows(self, table, idx, out_dtype=torch.float32):
        out = table[torch.clamp(idx.long(), 0, table.shape[0] - 1)].clone()
        out[idx < 0] = 0
        return out.to(out_dtype)

    def owner_plan(self, recv_ids, local_rows, G=1):
        runs = recv_ids.view(G, -1)
        assert bool((runs[:, 1:] >= runs[:, :-1]).all()), "every received bucket must be ascending (the owner merges runs)"
        return recv_ids.clone()

    def reduce_local(self, plan, srcs, B, E, counters=None):
        flat = torch.cat([d.reshape(B * K, E) for d, K in srcs])
        out = torch.zeros((max(plan.M, 1), E))
        for u in range(int(plan.n_unique)):
            sl = plan.sorted_src[int(plan.seg_offsets[u]):int(plan.seg_offsets[u + 1])].long()
            out[u] = flat[sl].sum(0)
        return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, kind="exact"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        from pathlib import Path
        root = Path(__file__).resolve().parents[1]
        sys.path.insert(0, str(root))
        from jodalrob_twotower_amd import ops
        from jodalrob_twotower_amd.distributed import PaddedRowExchange, RowExchange, ShardedStore
        E, B = 4, 9
        vocabs = [[5, 11, 3], [7, 2]]                      # two "towers"
        R = sum(map(sum, vocabs))
        rng = np.random.default_rng(123)
        table = torch.from_numpy(rng.standard_normal((R, E)).astype(np.float32))      # same on every rank
        store = ShardedStore(E, R, rank, world, "cpu", "dense" if kind == "exact" else "sparse")
        store.load_global(table)
        assert store.local_rows == len(range(rank, R, world))
        np.testing.assert_array_equal(store.gather_global().numpy(), table.numpy())   # shard <-> global round trip
        be = CheckerBackend()
        ex = RowExchange(store, backend=be) if kind == "exact" else PaddedRowExchange(store, backend=be)
        r2 = np.random.default_rng(1000 + rank)            # every rank has its own batch
        sides, outs, exp = [], [], []
        base = 0
        for v in vocabs:
            K = len(v)
            ids = np.stack([r2.integers(-2, vk + 2, B) for vk in v], axis=1).astype(np.int64)
            off = np.concatenate([[0], np.cumsum(v)[:-1]]) + base
            base += sum(v)
            out = torch.zeros(B, K * E)
            sides.append(ops.LookupSide(torch.from_numpy(ids.reshape(-1)), torch.from_numpy(off.astype(np.int64)),
                                        torch.tensor(v, dtype=torch.int64), out, K))
            outs.append(out)
            cl = np.minimum(np.maximum(ids, 0), np.array(v)[None, :] - 1) + off[None, :]
            exp.append((cl, table.numpy()[cl.reshape(-1)].reshape(B, K * E)))
        state = ex.forward(sides, B, True)
        for out, (_, e) in zip(outs, exp):
            assert np.array_equal(out.numpy(), e)          # pooled rows arrive bit-exact in slot order
        # backward: every rank sends gradients; owners must end up with the global scatter-add of ALL ranks
        srcs, contrib = [], []
        for (cl, _), s in zip(exp, sides):
            d = torch.from_numpy(r2.standard_normal((B, s.K * E)).astype(np.float32))
            srcs.append((d, s.K))
            contrib.append((cl.reshape(-1), d.numpy().reshape(B * s.K, E)))
        ex.backward(state, srcs, B)
        rows_all = np.concatenate([c[0] for c in contrib])
        vals_all = np.concatenate([c[1] for c in contrib])
        gathered = [None] * world
        dist.all_gather_object(gathered, (rows_all, vals_all))
        ref = np.zeros((R, E), np.float64)
        for rws, vls in gathered:
            np.add.at(ref, rws, vls.astype(np.float64))
        mine = ref[rank::world]
        np.testing.assert_allclose(be.dense_grad.numpy()[:mine.shape[0]], mine, rtol=1e-5, atol=1e-6)
        if kind == "padded":
            assert not ex.overflowed() and ex.C >= 256
            # a capacity that is too small must be flagged AND rejected by the product path, never silently wrong-and-quiet:
            # the rows that did not fit reach the towers as ZERO rows (not as some other row's embedding), the sticky flag
            # is set, and both the polling check (every forward / replay) and the synchronous one raise
            from jodalrob_twotower_amd.distributed import ExchangeOverflowError
            ex2 = PaddedRowExchange(store, backend=be, capacity=1)
            for o in outs:
                o.fill_(7.0)
            ex2.forward(sides, B, False)
            assert ex2.overflowed()
            n_zero = 0
            for out, (_, e) in zip(outs, exp):
                got = out.numpy().reshape(-1, E)
                want = e.reshape(-1, E)
                fit = (got == want).all(axis=1)
                assert ((got[~fit] == 0).all())             # every row that is not the right one is all-zero
                n_zero += int((~fit).sum())
            assert n_zero > 0
            with pytest.raises(ExchangeOverflowError):
                ex2.poll_overflow()
            with pytest.raises(ExchangeOverflowError):
                ex2.check_overflow()
            ex2.reset_capacity()
            ex2.forward(sides, B, False)                    # re-calibrated: clean again
            ex2.check_overflow()
            for out, (_, e) in zip(outs, exp):
                assert np.array_equal(out.numpy(), e)
        # dense-gradient reduction = SUM over ranks (the towers pre-scale by 1/world)
        g = [torch.full((5,), float(rank + 1))]
        ex.all_reduce_dense(g)
        assert torch.equal(g[0], torch.full((5,), float(sum(range(1, world + 1)))))
        q.put((rank, "ok"))
    except Exception as e:                                  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "exact"), (3, "exact"), (2, "padded"), (3, "padded")])
def test_row_exchange_gloo(world, kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
