"""SAD / SSD matching costs (reference: matching_cost/sad_ssd.py:39-368)."""
import numpy as np

from .. import runtime

from .matching_cost import AbstractMatchingCost, ConfigError


@AbstractMatchingCost.register_subclass("sad", "ssd")
class SadSsd(AbstractMatchingCost):
    def __init__(self, **cfg):
        super().instantiate_class(**cfg)

    def check_conf(self, **cfg):
        cfg = super().check_conf(**cfg)
        if cfg["matching_cost_method"] not in ("ssd", "sad"):
            raise ConfigError("matching_cost_method must be sad or ssd")
        if not isinstance(cfg["window_size"], int) or cfg["window_size"] <= 0 or cfg["window_size"] % 2 == 0:
            raise ConfigError("window_size must be an odd positive int")  # sad_ssd.py:69
        return cfg

    def compute_cost_volume(self, img_left, img_right, cost_volume):
        eng, dcv = self._bind_device_volume(img_left, img_right, cost_volume)
        left, right = runtime.select_band(img_left, self._band), runtime.select_band(img_right, self._band)  # sad_ssd.py:115-122
        min_left, max_left, min_right, max_right = np.amin(left), np.amax(left), np.amin(right), np.amax(right)
        if self._method == "sad":  # sad_ssd.py:132-137
            cmax = int(max(abs(max_left - min_right), abs(max_right - min_left)) * (self._window_size ** 2))
        else:
            cmax = int(max(abs(max_left - min_right) ** 2, abs(max_right - min_left) ** 2) * (self._window_size ** 2))
        cost_volume.attrs.update({"type_measure": "min", "cmax": cmax})
        eng.sad_ssd(dcv, self._window_size, self._method == "ssd")
        return cost_volume
