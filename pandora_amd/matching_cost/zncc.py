"""ZNCC matching cost (reference: matching_cost/zncc.py:38-277)."""
from .matching_cost import AbstractMatchingCost, ConfigError


@AbstractMatchingCost.register_subclass("zncc")
class Zncc(AbstractMatchingCost):
    def __init__(self, **cfg):
        super().instantiate_class(**cfg)

    def check_conf(self, **cfg):
        cfg = super().check_conf(**cfg)
        if not isinstance(cfg["window_size"], int) or cfg["window_size"] <= 0 or cfg["window_size"] % 2 == 0:
            raise ConfigError("window_size must be an odd positive int")
        return cfg

    def point_interval(self, img_left, img_right, disp):
        """zncc.py:73-112: empty ranges as soon as abs(disp) > nb_col - 2 * (window_size // 2)"""
        point_p, point_q = super().point_interval(img_left, img_right, disp)
        nx_left, nx_right = int(img_left.sizes["col"]), int(img_right.sizes["col"])
        if abs(disp) > nx_right - (int(self._window_size / 2) * 2):
            point_p, point_q = (nx_left, nx_left), (nx_right, nx_right)
        return point_p, point_q

    def compute_cost_volume(self, img_left, img_right, cost_volume):
        eng, dcv = self._bind_device_volume(img_left, img_right, cost_volume)
        cost_volume.attrs.update({"type_measure": "max", "cmax": 1})  # zncc.py:171-176
        eng.zncc(dcv, self._window_size)
        return cost_volume
