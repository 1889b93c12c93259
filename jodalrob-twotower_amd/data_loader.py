"""Device-resident feature store + pair loader -- the MI355X replacement for UnifiedBidDataset.__getitem__
and create_collate_fn (src/towers/pairs/unified_bid_data_loader.py:461-504, :630-684): the reference keeps
`dense_projected` / `categorical` in host numpy arrays, gathers them per batch with fancy indexing on a
2-thread pool, wraps the ids in a KJT (:827-841) and copies the batch to the GPU (scripts/train.py:261-273)
-- the README names this loader as the training bottleneck (23 it/s at 40 % GPU utilisation).

Here both stores live in HBM and a batch is assembled by tt_batch_gather from the pair indices: the only
per-step host->device traffic is 16 bytes per pair.  Batches have the reference's format
{"notice": {"dense", "kjt"}, "company": {"dense", "kjt"}} with sample-major ids.
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch

from . import ops
from .feature_preprocessor import FeaturePreprocessor
from .kjt import KeyedJaggedTensor
from .schema import TorchRecSchema


class DeviceFeatureStore:
    """dense_projected f32 [N, Din] and categorical i64 [N, K] of one tower, resident on the device."""

    def __init__(self, store: Dict, categorical_keys, device):
        self.device = torch.device(device)
        self.keys = list(categorical_keys)
        def resident(x, dtype):            # host arrays (the reference's stores: feature_store.py:76-79) or tensors already on a device
            t = x if torch.is_tensor(x) else torch.as_tensor(np.ascontiguousarray(x))
            return t.to(device=self.device, dtype=dtype).contiguous()
        self.dense = resident(store["dense_projected"], torch.float32)
        self.categorical = resident(store["categorical"], torch.int64)
        if self.dense.shape[0] != self.categorical.shape[0]:
            raise ValueError("dense_projected and categorical must have one row per entity")

    def __len__(self):
        return self.dense.shape[0]

    def gather(self, entity_idx: torch.Tensor) -> Dict:
        dense, ids = ops.batch_gather(entity_idx, self.dense, self.categorical)
        return {"dense": dense, "kjt": KeyedJaggedTensor(self.keys, ids)}


class DevicePairLoader:
    """Iterates over (notice_idx, company_idx) pairs in batches; len() = number of batches (drop_last=False,
    as torch's DataLoader default used by the reference: unified_bid_data_loader.py:1090-1110)."""

    def __init__(self, notice: DeviceFeatureStore, company: DeviceFeatureStore, pairs: np.ndarray, batch_size: int,
                 shuffle: bool, seed: int = 42):
        self.notice, self.company, self.batch_size, self.shuffle = notice, company, batch_size, shuffle
        pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        if len(pairs) and (pairs[:, 0].max() >= len(notice) or pairs[:, 1].max() >= len(company) or pairs.min() < 0):
            raise KeyError("pair refers to an entity that is not in the feature store")      # reference raises KeyError: :495-498
        self.pairs = torch.from_numpy(pairs).to(notice.device)
        # the epoch permutation is drawn where it is used: a host randperm of 1.6 M pairs + its upload cost ~20 ms per epoch,
        # a third of a 68-ms epoch at batch 8192 (profiles/NOTES.md, round 4)
        self._gen = torch.Generator(device=self.pairs.device)
        self._gen.manual_seed(seed)

    def __len__(self) -> int:
        return (self.pairs.shape[0] + self.batch_size - 1) // self.batch_size

    def epoch_order(self) -> Optional[torch.Tensor]:
        """This epoch's permutation of the pair list on the device (None when not shuffling); drawn from the loader's seeded
        generator, so an epoch iterated through __iter__ and one driven through step_batches() see the same batches."""
        n = self.pairs.shape[0]
        return torch.randperm(n, generator=self._gen, device=self.pairs.device) if self.shuffle else None

    def batch(self, order: Optional[torch.Tensor], lo: int) -> Dict:
        sel = self.pairs[lo:lo + self.batch_size] if order is None else self.pairs[order[lo:lo + self.batch_size]]
        return {"notice": self.notice.gather(sel[:, 0].contiguous()), "company": self.company.gather(sel[:, 1].contiguous())}

    def __iter__(self) -> Iterator[Dict]:
        order = self.epoch_order()
        for lo in range(0, self.pairs.shape[0], self.batch_size):
            yield self.batch(order, lo)

    def eval_batches(self, graphed_eval, eager_eval=None, max_batches: Optional[int] = None) -> int:
        """One pass over the pairs in order (no shuffling is applied here: the reference's test loader does not shuffle) through a
        GraphedEvalStep: full batches from the stores, the ragged last one through `eager_eval(batch) -> metrics dict` (skipped
        when None).  Returns the number of batches evaluated; the sums sit in `graphed_eval`."""
        n, B, done = self.pairs.shape[0], self.batch_size, 0
        order = self.epoch_order()
        for lo in range(0, n, B):
            if max_batches is not None and done >= max_batches:
                break
            if lo + B <= n:
                graphed_eval.step_from_store(self.notice, self.company, self.pairs, order, lo)
            elif eager_eval is not None:
                graphed_eval.add_eager(eager_eval(self.batch(order, lo)))
            else:
                continue
            done += 1
        return done

    def step_batches(self, graphed_step, eager_step=None, ragged_step=None):
        """One epoch on the fast path: every full batch is gathered out of the stores by the captured step's own hand-over launch
        (GraphedTrainStep.step_from_store) and replayed; the ragged last batch -- a captured step has one batch size -- goes
        through `ragged_step` (a second GraphedTrainStep captured at the remainder's size, sharing task and optimiser) or, without
        one, through `eager_step(batch)` (skipped when None).  Yields the step's result dict per batch (an unrolled step's U results
        arrive together, after its one launch: read them before the next launch overwrites them)."""
        order = self.epoch_order()
        n, B = self.pairs.shape[0], self.batch_size
        U = int(getattr(graphed_step, "unroll", 1))
        lo = 0
        while U > 1 and lo + U * B <= n:                      # unrolled.UnrolledTrainStep: U full batches per graph launch
            for r in graphed_step.steps_from_store(self.notice, self.company, self.pairs, order, [lo + j * B for j in range(U)]):
                yield r
            lo += U * B
        for lo in range(lo, n, B):
            if lo + B <= n:
                yield graphed_step.step_from_store(self.notice, self.company, self.pairs, order, lo)
            elif ragged_step is not None and ragged_step.static["notice"]["dense"].shape[0] == n - lo:
                yield ragged_step.step_from_store(self.notice, self.company, self.pairs, order, lo)
            elif eager_step is not None:
                yield eager_step(self.batch(order, lo))

    def ragged_example(self) -> Optional[Dict]:
        """A batch of the remainder's size (pairs % batch_size), or None when the epoch has no ragged batch: the example a second
        captured step is built from."""
        n, B = self.pairs.shape[0], self.batch_size
        r = n % B
        if r == 0 or n < B:
            return None
        sel = self.pairs[:r]
        return {"notice": self.notice.gather(sel[:, 0].contiguous()), "company": self.company.gather(sel[:, 1].contiguous())}


def sklearn_split_indices(n: int, test_size: float, seed: int) -> Tuple[np.ndarray, np.ndarray]:
    """(train, test) index arrays -- membership AND order -- of `sklearn.model_selection.train_test_split(range(n),
    test_size=test_size, random_state=seed)`, which is how the reference's test-mode loader splits its pair list
    (unified_bid_data_loader.py:1222-1226; scripts/train.py: test_split 0.2, shuffle_seed 42).  scikit-learn's ShuffleSplit draws
    `RandomState(seed).permutation(n)`, takes the first n_test = ceil(test_size * n) entries as the test set and the following
    n - n_test as the training set.  Index work: bit-exact against tests/golden/split_indices.json (generated with scikit-learn
    itself by oracle/gen_split_fixture.py); scikit-learn is not needed at run time."""
    if not (0.0 < test_size < 1.0):
        return np.arange(n, dtype=np.int64), np.empty(0, dtype=np.int64)        # reference: test_split == 0 keeps every pair (:1227-1228)
    n_test = int(np.ceil(test_size * n))
    n_train = n - n_test
    if n_train <= 0:
        raise ValueError(f"With n_samples={n}, test_size={test_size} the resulting train set will be empty")     # sklearn's ValueError
    perm = np.random.RandomState(seed).permutation(n)
    return perm[n_test:n_test + n_train].astype(np.int64), perm[:n_test].astype(np.int64)


def create_unified_bid_dataloaders(db_engine, schema: TorchRecSchema, batch_size: int = 32, limit: Optional[int] = None,
                                   test_split: float = 0.1, shuffle_seed: int = 42, num_workers: int = 4, pin_memory: bool = False,
                                   prefetch_factor: int = 2, persistent_workers: bool = False, streaming: bool = False,
                                   chunk_size: int = 1000, load_all_features: bool = True, feature_chunksize: int = 5000,
                                   feature_limit: Optional[int] = None, use_preprocessor: bool = True, test_mode: bool = False,
                                   pair_limit: Optional[int] = None, device="cuda:0") -> Tuple[DevicePairLoader, DevicePairLoader]:
    """Same signature as the reference factory (:971-990); `db_engine` is a feature/pair source object
    (see FeaturePreprocessor) offering additionally `load_pairs(pair_schema, limit)` -> list of
    ((bidntceno, bidntceord), bizno).  Worker / pinning / streaming knobs are accepted and ignored: there is
    no host-side batch assembly left to parallelise."""
    pre = FeaturePreprocessor(schema, device=str(device))
    stores = pre.preprocess_all(db_engine, feature_chunksize=feature_chunksize, feature_limit=feature_limit, show_progress=False)
    n2i, c2i = pre.build_id_mappings(stores)
    raw = db_engine.load_pairs(schema.pair, pair_limit if test_mode else limit)
    idx = np.empty((len(raw), 2), dtype=np.int64)
    for i, (nk, ck) in enumerate(raw):
        if tuple(nk) not in n2i:
            raise KeyError(f"Notice ID not found in features: {tuple(nk)}")
        if str(ck) not in c2i:
            raise KeyError(f"Company ID not found in features: {ck}")
        idx[i] = (n2i[tuple(nk)], c2i[str(ck)])
    train_sel, test_sel = sklearn_split_indices(len(idx), test_split, shuffle_seed)
    test_idx, train_idx = idx[test_sel], idx[train_sel]
    ns = DeviceFeatureStore(stores["notice"], schema.notice.categorical, device)
    cs = DeviceFeatureStore(stores["company"], schema.company.categorical, device)
    return (DevicePairLoader(ns, cs, train_idx, batch_size, shuffle=True, seed=shuffle_seed),
            DevicePairLoader(ns, cs, test_idx, batch_size, shuffle=False, seed=shuffle_seed))
