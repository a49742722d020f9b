"""MI355X-native RetinaNet detection hot path (librtn.so + a thin PyTorch-ROCm host).

The directory name is not a Python identifier; load it with
    importlib.import_module("retinanet-for-table-detection_amd")
or put this directory on sys.path to get the reference's `model` call surface
(`from model import anchors, losses, layers, defineModel, utils`).
Importing this package loads librtn.so and raises if it is missing: there is no CPU fallback.
"""
from . import _lib
from ._lib import lib, Handle, RtnError, LIB_PATH

__all__ = ["_lib", "lib", "Handle", "RtnError", "LIB_PATH"]
