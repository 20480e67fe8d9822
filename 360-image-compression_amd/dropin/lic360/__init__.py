"""Put THIS directory's parent (`360-image-compression_amd/dropin`) on PYTHONPATH -- and nothing else of this repository -- to
give the reference its `lic360` extension module while the reference's OWN `lic360_operator/`, `test/model_zoo.py` and
`test/lic360_demo.py` stay in charge above it.  This stub replaces itself by the real package
`360-image-compression_amd/lic360/` (the ctypes shim over liblic360_hip.so)."""
import importlib.util
import os
import sys

_real_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "lic360")
_spec = importlib.util.spec_from_file_location("lic360", os.path.join(_real_dir, "__init__.py"), submodule_search_locations=[_real_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lic360"] = _mod
_spec.loader.exec_module(_mod)
