"""Import alias: `import maxtext_indextts2_amd as ixt` loads the package directory
`maxtext-indextts2_amd/` (whose name is not a Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
  sys.path.insert(0, _root)
_pkg = importlib.import_module("maxtext-indextts2_amd")
sys.modules[__name__] = _pkg
