"""Import alias: the product package lives in ``unsupervised-part-segmentation_amd/`` (a directory name
with hyphens, as the build contract asks) which Python cannot import by name; this module loads it
under the name ``upsparts_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "unsupervised-part-segmentation_amd")
_spec = importlib.util.spec_from_file_location("upsparts_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["upsparts_amd"] = _mod
_spec.loader.exec_module(_mod)
