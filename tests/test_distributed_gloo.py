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

    def owner_accumulate(self, store, plan, d_rows):
        g = torch.zeros_like(store.weight)
        g.index_add_(0, plan, d_rows)
        self.dense_grad = g


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        from pathlib import Path
        root = Path(__file__).resolve().parents[1]
        sys.path.insert(0, str(root))
        from jodalrob_twotower_amd import ops
        from jodalrob_twotower_amd.distributed import RowExchange, ShardedStore
        E, B = 4, 9
        vocabs = [[5, 11, 3], [7, 2]]                      # two "towers"
        R = sum(map(sum, vocabs))
        rng = np.random.default_rng(123)
        table = torch.from_numpy(rng.standard_normal((R, E)).astype(np.float32))      # same on every rank
        store = ShardedStore(E, R, rank, world, "cpu", "dense")
        store.load_global(table)
        assert store.local_rows == len(range(rank, R, world))
        np.testing.assert_array_equal(store.gather_global().numpy(), table.numpy())   # shard <-> global round trip
        be = CheckerBackend()
        ex = RowExchange(store, backend=be)
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


@pytest.mark.parametrize("world", [2, 3])
def test_row_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
