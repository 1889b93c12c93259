"""CPU ORACLE, torch-eager form -- the vectorised CPU restatement SURVEY.md section 8(d) asks for as the reported CPU
baseline (`bench.py`'s `cpu_baseline`, kind "port").

TEST INFRASTRUCTURE ONLY: imported by tests/ and by bench.py's cpu_baseline leg, never by the product path.

Same step as the reference (forward + autograd backward + torch.optim.Adam over EVERY parameter, dense [V_k, E] table
gradients: the reference's semantics), written against a plain state dict with torch.nn.functional ops on all host
cores.  It differs from the reference's own Python in exactly one place: `CategoricalEmbedder._kjt_to_dict`
(src/towers/cat_embed.py:98-123) loops over B*K ids with scalar tensor ops (~1.2 s per step at B = 8192: SURVEY section 6);
here the unpack + clamp is one vectorised expression -- so this is an OPTIMISTIC (faster) stand-in for the reference.
Pinned against the reference-generated golden vectors in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

NT, CT = "two_tower_model.notice_tower.", "two_tower_model.company_tower."


def _tower(state, prefix, keys, vocab_sizes, dense, values, train):
    """BaseTower.forward (src/towers/tower/base_tower.py:101-147) + CategoricalEmbedder.forward (cat_embed.py:126-178)."""
    K = len(keys)
    B = values.numel() // K                                                  # cat_embed.py:98
    ids = values[:B * K].view(B, K)
    hi = torch.as_tensor(vocab_sizes, dtype=torch.int64) - 1
    ids = torch.minimum(ids.clamp(min=0), hi[None, :])                       # cat_embed.py:114-117, vectorised
    parts = [F.linear(dense, state[prefix + "dense_projection.weight"], state[prefix + "dense_projection.bias"])]
    parts += [F.embedding(ids[:, i], state[f"{prefix}categorical_embedder.embeddings.{k}.weight"]) for i, k in enumerate(keys)]
    h = torch.cat(parts, dim=1)                                              # base_tower.py:139
    i = 0
    while f"{prefix}mlp.{4 * i + 2}.running_mean" in state:                  # Linear, ReLU, BatchNorm1d, Dropout(p = 0 here)
        h = F.relu(F.linear(h, state[f"{prefix}mlp.{4 * i}.weight"], state[f"{prefix}mlp.{4 * i}.bias"]))
        h = F.batch_norm(h, state[f"{prefix}mlp.{4 * i + 2}.running_mean"], state[f"{prefix}mlp.{4 * i + 2}.running_var"],
                         state[f"{prefix}mlp.{4 * i + 2}.weight"], state[f"{prefix}mlp.{4 * i + 2}.bias"], training=train,
                         momentum=0.1, eps=1e-5)
        if train:                                                            # nn.BatchNorm1d bumps its counter in forward
            state[f"{prefix}mlp.{4 * i + 2}.num_batches_tracked"] += 1
        i += 1
    y = F.linear(h, state[f"{prefix}mlp.{4 * i}.weight"], state[f"{prefix}mlp.{4 * i}.bias"])
    return F.normalize(y, p=2, dim=1)                                        # base_tower.py:145


def task_loss(state, batch, keys_n, keys_c, vocab_n, vocab_c, temperature=1.0, train=True):
    """TwoTowerTrainTask.forward (src/towers/two_tower_train_task.py:40-134): returns (loss, similarity matrix)."""
    n = _tower(state, NT, keys_n, vocab_n, batch["notice_dense"], batch["notice_ids"].reshape(-1), train)
    c = _tower(state, CT, keys_c, vocab_c, batch["company_dense"], batch["company_ids"].reshape(-1), train)
    S = torch.mm(n, c.t())
    if temperature != 1.0:
        S = S / temperature                                                  # :107-110
    labels = torch.arange(S.shape[0])
    return 0.5 * (F.cross_entropy(S, labels) + F.cross_entropy(S.t(), labels)), S


def make_state(state_np: dict):
    """numpy state dict -> torch tensors; parameters (everything but the BN running statistics) require grad."""
    out = {}
    for k, v in state_np.items():
        t = torch.as_tensor(v).clone()
        if t.is_floating_point() and "running_" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def parameters(state):
    return [v for v in state.values() if v.requires_grad]
