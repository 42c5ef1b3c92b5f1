"""Import alias for the package directory `variational-self-organizing-maps_amd/`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("variational-self-organizing-maps_amd")
sys.modules[__name__] = _pkg
