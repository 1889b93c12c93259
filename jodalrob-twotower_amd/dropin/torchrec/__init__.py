"""Drop-in shim: same import path as the reference's (the torchrec package; only its KeyedJaggedTensor container is used on this path); re-exports the MI355X implementation.
Put this directory's parent (`.../dropin`) and the repository root first on sys.path."""
from jodalrob_twotower_amd.kjt import KeyedJaggedTensor  # noqa: F401
