"""The three helpers of the reference's utils/graph_utils.py that the front-end path
uses, under their own names: `compute_err` (utils/graph_utils.py:7-11, called by
PnP.supporters_pair), `homogenous` / `non_homogenous` (utils/graph_utils.py:26-35,
called when the right camera's pose is composed).  The rest of that module serves the
GTSAM back-end, which stays the reference's."""
from __future__ import annotations

import numpy as np
import torch

from .. import kernels


def compute_err(meas_pts, proj_pts):
    """Pixel distance between measured and projected pixels, (n,) float64.  Computed on
    the GPU; a device tensor comes back if a device tensor came in, else a NumPy array."""
    err = kernels.compute_err(meas_pts, proj_pts)
    if torch.is_tensor(meas_pts) and meas_pts.is_cuda:
        return err
    return err.cpu().numpy()


def homogenous(Rt):
    """(3,4) -> (4,4) by appending the row [0, 0, 0, 1]."""
    if torch.is_tensor(Rt):
        last = torch.tensor([[0, 0, 0, 1]], dtype=Rt.dtype, device=Rt.device)
        return torch.cat([Rt, last], dim=0)
    return np.vstack([Rt, [0, 0, 0, 1]])


def non_homogenous(T):
    """(4,4) -> its upper (3,4) block (a view, as in the reference)."""
    return T[:3, :]
