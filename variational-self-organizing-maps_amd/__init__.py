"""MI355X-native VSOM training hot path (BMU search + neighbourhood mean/sigma^2 update).

The directory name is not a Python identifier; import it through the `vsom_amd` alias module
at the repository root (or importlib.import_module("variational-self-organizing-maps_amd")).
"""
from . import capi  # noqa: F401
from .capi import (BATCHMAP, CLR, EXPONENTIAL, INVERSE_PROPORTIONAL, MEDIAN, STANDARD,  # noqa: F401
                   Context, Group, VsomError)
