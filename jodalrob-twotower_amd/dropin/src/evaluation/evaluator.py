"""Drop-in shim: same import path as the reference's src/evaluation/evaluator.py; re-exports the MI355X implementation.
Put this directory's parent (`.../dropin`) and the repository root first on sys.path."""
from jodalrob_twotower_amd.evaluator import TwoTowerEvaluator  # noqa: F401
