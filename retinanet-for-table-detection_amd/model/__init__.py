"""Drop-in `model` package: the reference's call surface (model/anchors.py, losses.py, layers.py, defineModel.py, utils.py,
initializers.py, Parameters.py — same names, argument meaning and error behaviour) on top of librtn.so.

Use:  sys.path.insert(0, ".../retinanet-for-table-detection_amd")  then  `from model import anchors, losses, ...`
exactly as RetinaNet.py / csv_generator.py do.  Arrays come in and go out as NumPy (like Keras); all arithmetic runs in the
HIP library on cuda:0 — importing this package without librtn.so or without a GPU raises (no CPU fallback).
"""
from . import Parameters, _rt  # noqa: F401
