"""Census matching cost (reference: matching_cost/census.py:39-153 + cpp/src/census.cpp)."""
from .matching_cost import AbstractMatchingCost, ConfigError


@AbstractMatchingCost.register_subclass("census")
class Census(AbstractMatchingCost):
    def __init__(self, **cfg):
        super().instantiate_class(**cfg)

    def check_conf(self, **cfg):
        cfg = super().check_conf(**cfg)
        if not isinstance(cfg["window_size"], int) or cfg["window_size"] not in (3, 5, 7, 9, 11, 13):
            raise ConfigError("census window_size must be in (3, 5, 7, 9, 11, 13)")  # census.py:68
        return cfg

    def compute_cost_volume(self, img_left, img_right, cost_volume):
        eng, dcv = self._bind_device_volume(img_left, img_right, cost_volume)
        cost_volume.attrs.update({"type_measure": "min", "cmax": int(self._window_size ** 2)})  # census.py:116-122
        eng.census(dcv, self._window_size)
        return cost_volume
