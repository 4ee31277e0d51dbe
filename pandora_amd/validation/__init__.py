"""Validation step: cross-checking on the device."""
from .validation import AbstractValidation, CrossCheckingAccurate, allocate_confidence_map  # noqa: F401
