"""Drop-in shim: same import path as the reference's src/torchrec_preprocess/schema.py; re-exports the MI355X implementation.
Put this directory's parent (`.../dropin`) and the repository root first on sys.path."""
from jodalrob_twotower_amd.schema import (PairSchema, SideSchema, TorchRecSchema, build_side_schema_from_meta,  # noqa: F401
                                          build_torchrec_schema_from_meta)
