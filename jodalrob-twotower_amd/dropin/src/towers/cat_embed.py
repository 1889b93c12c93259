"""Drop-in shim: same import path as the reference's src/towers/cat_embed.py; re-exports the MI355X implementation.
Put this directory's parent (`.../dropin`) and the repository root first on sys.path."""
from jodalrob_twotower_amd.cat_embed import CategoricalEmbedder, create_categorical_embedder  # noqa: F401
