"""Runtime glue shared by the mirror modules: one library handle on the current device, NumPy <-> device copies."""
import importlib
import os
import sys

import numpy as np
import torch

_pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_root = os.path.dirname(_pkg_dir)
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module(os.path.basename(_pkg_dir))
L = _pkg._lib
engine = importlib.import_module(_pkg.__name__ + ".engine")
weights = importlib.import_module(_pkg.__name__ + ".weights")
trainer = importlib.import_module(_pkg.__name__ + ".trainer")
keras_h5 = importlib.import_module(_pkg.__name__ + ".keras_h5")

_handle = None


def handle():
    global _handle
    if _handle is None:
        if not torch.cuda.is_available():
            raise RuntimeError("the model package runs on a ROCm GPU through librtn.so: no CPU fallback exists")
        _handle = L.Handle(torch.cuda.current_device())
    _handle.set_stream(torch.cuda.current_stream().cuda_stream)
    return _handle


def dev(a, dtype):
    """NumPy / tensor -> contiguous device tensor of `dtype`."""
    if isinstance(a, torch.Tensor):
        return a.to(device="cuda", dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a)).to(device="cuda", dtype=dtype).contiguous()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()
