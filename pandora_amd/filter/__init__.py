"""Disparity-map filters."""
from .filter import AbstractFilter  # noqa: F401
from .median import MedianFilter  # noqa: F401
from .bilateral import BilateralFilter  # noqa: F401
from .median_for_intervals import MedianForIntervalsFilter  # noqa: F401
from .disparity_denoiser import DisparityDenoiser  # noqa: F401
