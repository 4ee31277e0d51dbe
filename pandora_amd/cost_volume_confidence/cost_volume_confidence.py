"""AbstractCostVolumeConfidence (reference: cost_volume_confidence/cost_volume_confidence.py:36-250)."""
from abc import ABCMeta, abstractmethod

import numpy as np

from ..validation.validation import allocate_confidence_map as _allocate


class AbstractCostVolumeConfidence:
    __metaclass__ = ABCMeta

    confidence_methods_avail = {}
    cfg = None
    _indicator = ""

    def __new__(cls, **cfg):
        if cls is AbstractCostVolumeConfidence:
            method = cfg.get("confidence_method")
            if isinstance(method, str):
                try:
                    return super(AbstractCostVolumeConfidence, cls).__new__(cls.confidence_methods_avail[method])
                except KeyError:
                    raise KeyError("No confidence method named {} supported".format(method))
            raise KeyError("No confidence method named {} supported".format(method))
        return super(AbstractCostVolumeConfidence, cls).__new__(cls)

    @classmethod
    def register_subclass(cls, short_name):
        def decorator(subclass):
            cls.confidence_methods_avail[short_name] = subclass
            return subclass

        return decorator

    @abstractmethod
    def desc(self):
        """Describes the confidence method"""

    @abstractmethod
    def confidence_prediction(self, disp, img_left, img_right, cv):
        """Computes a confidence prediction; returns (disp, cv) with the confidence_measure updated."""

    @staticmethod
    def normalize_with_extremum(confidence, dataset, nbr_etas, subpix=1):
        """cost_volume_confidence.py:114-138"""
        gmin, gmax = dataset.attrs["global_disparity"][0], dataset.attrs["global_disparity"][1]
        return np.copy(confidence) / ((gmax - gmin) * nbr_etas * subpix)

    @staticmethod
    def allocate_confidence_map(name_confidence_measure, confidence_map, disp, cv):
        """cost_volume_confidence.py:141-246"""
        return _allocate(name_confidence_measure, confidence_map, disp, cv)
