"""model/losses.py of the reference (focal :5-46, smooth_l1 :49-91) on the device.  The functors take (y_true, y_pred) as
NumPy arrays or tensors and return the scalar loss (NumPy float32), like evaluating the Keras loss tensor."""
import numpy as np
import torch

from . import _rt

L = _rt.L


def _sums(y_true_cls, y_true_reg, y_cls, y_reg, alpha, gamma, sigma):
    h = _rt.handle()
    B, N = (y_true_cls if y_true_cls is not None else y_true_reg).shape[:2]
    K = 1
    if y_true_cls is None:                       # smooth-L1 only: a state column that selects nothing for the focal half
        lab = torch.full((B, N, 2), -1.0, dtype=torch.float32, device="cuda")
        cls = torch.full((B, N, 1), 0.5, dtype=torch.float32, device="cuda")
    else:
        lab, cls = _rt.dev(y_true_cls, torch.float32), _rt.dev(y_cls, torch.float32)
        K = cls.shape[2]
    if y_true_reg is None:
        regt = torch.zeros(B, N, 5, dtype=torch.float32, device="cuda")
        reg = torch.zeros(B, N, 4, dtype=torch.float32, device="cuda")
    else:
        regt, reg = _rt.dev(y_true_reg, torch.float32), _rt.dev(y_reg, torch.float32)
    rows = B * N
    wsb = L.lib.rtn_retina_loss_workspace_bytes(rows)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    sums = torch.zeros(4, dtype=torch.float64, device="cuda")
    h.check(L.lib.rtn_retina_loss_fwd(h.raw, rows, K, lab.data_ptr(), regt.data_ptr(), cls.data_ptr(), reg.data_ptr(), alpha, gamma,
                                      sigma, sums.data_ptr(), ws.data_ptr(), wsb))
    return _rt.host(sums)


def focal(alpha=0.25, gamma=2.0):
    """ Create a functor for computing the focal loss (model/losses.py:5-46)."""
    def _focal(y_true, y_pred):
        s = _sums(y_true, None, y_pred, None, alpha, gamma, 3.0)
        return np.float32(s[0] / max(1.0, s[2]))
    return _focal


def smooth_l1(sigma=3.0):
    """ Create a smooth L1 loss functor (model/losses.py:49-91)."""
    def _smooth_l1(y_true, y_pred):
        s = _sums(None, y_true, None, y_pred, 0.25, 2.0, sigma)
        return np.float32(s[1] / max(1.0, s[3]))
    return _smooth_l1
