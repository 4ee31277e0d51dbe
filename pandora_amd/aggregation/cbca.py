"""Cross-Based Cost Aggregation (reference: aggregation/cbca.py:39-295)."""
from .. import runtime
from ..matching_cost.matching_cost import ConfigError
from .aggregation import AbstractAggregation


@AbstractAggregation.register_subclass("cbca")
class CrossBasedCostAggregation(AbstractAggregation):
    _CBCA_INTENSITY = 30.0
    _CBCA_DISTANCE = 5

    def __init__(self, **cfg):
        self.cfg = self.check_conf(**cfg)
        self._cbca_intensity = self.cfg["cbca_intensity"]
        self._cbca_distance = self.cfg["cbca_distance"]

    def check_conf(self, **cfg):
        cfg.setdefault("cbca_intensity", self._CBCA_INTENSITY)
        cfg.setdefault("cbca_distance", self._CBCA_DISTANCE)
        if cfg.get("aggregation_method") != "cbca":
            raise ConfigError("aggregation_method must be cbca")
        if not isinstance(cfg["cbca_intensity"], float) or cfg["cbca_intensity"] <= 0:
            raise ConfigError("cbca_intensity must be a float > 0")  # cbca.py:74-77
        if not isinstance(cfg["cbca_distance"], int) or cfg["cbca_distance"] <= 0:
            raise ConfigError("cbca_distance must be an int > 0")
        return cfg

    def desc(self):
        print("CrossBasedCostAggregation method")

    def cost_volume_aggregation(self, img_left, img_right, cv, **cfg):
        subpix = cv.attrs["subpixel"]
        eng = runtime.ensure_pair(img_left, img_right, subpix, band=cv.attrs.get("band_correl"))
        dcv = cv["cost_volume"].device_cv
        eng.cbca(dcv, int(cv.attrs["offset_row_col"]), float(self._cbca_intensity), int(self._cbca_distance))
        cv.attrs["aggregation"] = "cbca"
        cv.attrs["cmax"] = cv.attrs["cmax"] * ((self._cbca_distance * 2) - 1) ** 2  # cbca.py:181-182

    def computes_cross_supports(self, img_left, img_right, cv):
        """cbca.py:184-295: the cross support regions (left, right, top, bottom arms, int16) of the left image and of every
        sub-pixel phase of the right image, on the images cropped by the matching-cost offset -> (cross_left, [cross_right...])"""
        subpix = cv.attrs["subpixel"]
        eng = runtime.ensure_pair(img_left, img_right, subpix, band=cv.attrs.get("band_correl"))
        off = int(cv.attrs["offset_row_col"])
        args = (off, float(self._cbca_intensity), int(self._cbca_distance))
        return eng.cross_support(0, *args), [eng.cross_support(1 + k, *args) for k in range(subpix)]
