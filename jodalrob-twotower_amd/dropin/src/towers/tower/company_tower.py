"""Drop-in shim: same import path as the reference's src/towers/tower/company_tower.py; re-exports the MI355X implementation.
Put this directory's parent (`.../dropin`) and the repository root first on sys.path."""
from jodalrob_twotower_amd.towers import CompanyTower  # noqa: F401
