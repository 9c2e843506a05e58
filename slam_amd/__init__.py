"""slam_amd: MI355X (gfx950) implementation of the GuyShmueli/SLAM stereo
front-end hot path - Hamming kNN matching, stereo row filter, DLT
triangulation, PnP-RANSAC consensus - behind the reference's class interface.

Importing the package loads libslamhip.so and fails if it has not been built.
"""
from . import _lib  # noqa: F401  (loads the shared library; raises if missing)
from .types import DMatch, KeyPoint, MatchList
from .frontend.io.camera_model import CameraModel
from .frontend.vision.descriptor_matcher import DescriptorMatcher
from .frontend.geometry.triangulation import Triangulation
from .frontend.geometry.pnp import PnP, SOLVEPNP_AP3P, SOLVEPNP_ITERATIVE

__all__ = ["DMatch", "KeyPoint", "MatchList", "CameraModel", "DescriptorMatcher",
           "Triangulation", "PnP", "SOLVEPNP_AP3P", "SOLVEPNP_ITERATIVE"]
