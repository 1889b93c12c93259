"""Drop-in shim: same import path as the reference's src/towers/two_tower_model.py; re-exports the MI355X implementation.
Put this directory's parent (`.../dropin`) and the repository root first on sys.path."""
from jodalrob_twotower_amd.two_tower_model import TwoTowerModel, create_two_tower_model  # noqa: F401
