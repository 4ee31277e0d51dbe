"""Multiscale processing: coarse-to-fine disparity ranges."""
from .multiscale import AbstractMultiscale  # noqa: F401
from .fixed_zoom_pyramid import FixedZoomPyramid  # noqa: F401
from .pyramid import prepare_pyramid, get_pyramids, read_multiscale_params  # noqa: F401
