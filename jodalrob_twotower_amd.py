"""Import alias: the package directory is `jodalrob-twotower_amd/` (hyphen), which Python cannot name
in an import statement.  `import jodalrob_twotower_amd` loads that directory as a regular package."""
import importlib.util
import sys
from pathlib import Path

_dir = Path(__file__).resolve().parent / "jodalrob-twotower_amd"
_spec = importlib.util.spec_from_file_location(__name__, _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
