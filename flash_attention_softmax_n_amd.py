"""Import shim: `import flash_attention_softmax_n_amd` loads the package kept in the directory
`flash-attention-softmax-n_amd/` (a hyphenated directory name is not importable by itself)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flash-attention-softmax-n_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
