"""Drop-in shim: same import path as the reference's src/towers/pairs/unified_bid_data_loader.py; re-exports the MI355X implementation.
Put this directory's parent (`.../dropin`) and the repository root first on sys.path."""
from jodalrob_twotower_amd.data_loader import create_unified_bid_dataloaders, DevicePairLoader, DeviceFeatureStore  # noqa: F401
from jodalrob_twotower_amd.kjt import build_batch_kjt as _build_batch_kjt  # noqa: F401
