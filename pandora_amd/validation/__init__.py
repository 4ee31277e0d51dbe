"""Validation step: cross-checking and the interpolation of the rejected pixels, on the device."""
from .validation import AbstractValidation, CrossCheckingAccurate, allocate_confidence_map  # noqa: F401
from .interpolated_disparity import AbstractInterpolation, McCnnInterpolation, SgmInterpolation  # noqa: F401
