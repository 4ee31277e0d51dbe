"""8-path Semi-Global Matching plugin, registered as "sgm".

In the reference this step is the EXTERNAL plugin pandora_plugin_libsgm==1.5.7 (pyproject.toml:59-61)
wrapping CNES libSGM; only its configuration surface is documented in the reference tree
(docs/source/userguide/plugins/plugin_libsgm.rst:88-209).  This class keeps that surface
(overcounting, min_cost_paths, penalty{penalty_method, P1, P2, p2_method}) for the constant-penalty
"sgm_penalty" method and runs this build's own SGM definition (oracle/oracle.c orc_sgm) as
HIP kernels.  Numerical parity with libSGM is UNPINNED (no reference test pins it).
"""
import numpy as np

from ..matching_cost.matching_cost import ConfigError
from .optimization import AbstractOptimization


@AbstractOptimization.register_subclass("sgm")
class Sgm(AbstractOptimization):
    _P1 = 8
    _P2 = 32
    _OVERCOUNTING = False
    _MIN_COST_PATH = False

    def __init__(self, _img=None, **cfg):
        self.cfg = self.check_conf(**cfg)
        pen = self.cfg["penalty"]
        self._p1, self._p2 = float(pen["P1"]), float(pen["P2"])
        self._overcounting = bool(self.cfg["overcounting"])
        self._use_confidence = self.cfg.get("use_confidence") or None

    def check_conf(self, **cfg):
        cfg.setdefault("overcounting", self._OVERCOUNTING)
        cfg.setdefault("min_cost_paths", self._MIN_COST_PATH)
        pen = dict(cfg.get("penalty") or {})
        pen.setdefault("penalty_method", "sgm_penalty")
        pen.setdefault("p2_method", "constant")
        pen.setdefault("P1", self._P1)
        pen.setdefault("P2", self._P2)
        cfg["penalty"] = pen
        if cfg.get("optimization_method") != "sgm":
            raise ConfigError("optimization_method must be sgm")
        if pen["penalty_method"] != "sgm_penalty" or pen["p2_method"] != "constant":
            raise ConfigError("pandora_amd implements the constant-penalty sgm_penalty method only")
        if not isinstance(pen["P1"], (int, float)) or not isinstance(pen["P2"], (int, float)) or pen["P1"] <= 0 or pen["P2"] <= pen["P1"]:
            raise ConfigError("penalties must satisfy 0 < P1 < P2 (plugin_libsgm.rst:170-185)")
        if cfg["min_cost_paths"]:
            raise ConfigError("min_cost_paths is not implemented")
        if cfg.get("geometric_prior") not in (None, False, {"source": "internal"}):
            raise ConfigError("geometric_prior (piecewise optimisation) is out of scope of pandora_amd")
        use = cfg.get("use_confidence")
        if use not in (None, False) and not (isinstance(use, str) and use.split(".")[0] == "cost_volume_confidence"):
            raise ConfigError("use_confidence names the cost_volume_confidence step whose ambiguity is applied, "
                              "e.g. 'cost_volume_confidence' or 'cost_volume_confidence.before'")
        return cfg

    def desc(self):
        print("Semi-global matching (8 paths) optimization method")

    def optimize_cv(self, cv, img_left, img_right):
        arr = cv["cost_volume"]
        if not hasattr(arr, "device_cv"):
            raise TypeError("optimize_cv needs a device-resident cost volume (pandora_amd has no CPU path)")
        dcv = arr.device_cv
        is_max = cv.attrs["type_measure"] == "max"
        cmax = float(cv.attrs["cmax"])
        invalid_cost = cmax + 1.0  # this build's convention: NaN cells cost "worse than the worst"
        if self._use_confidence:
            # plugin_libsgm.rst:38-47: E(D) = sum_p C(p, D_p) * Confidence(p) + ...; the ambiguity confidence computed by the
            # named step (indicator suffix = what follows "cost_volume_confidence" in the step's name).  "If not [computed],
            # default confidence values equal to 1 will be used"
            suffix = self._use_confidence[len("cost_volume_confidence"):]
            name = "confidence_from_ambiguity" + suffix
            if "confidence_measure" in cv.data_vars and name in list(cv.coords.get("indicator", [])):
                layer = list(cv.coords["indicator"]).index(name)
                dcv.engine.scale_pixels(dcv, np.asarray(cv["confidence_measure"].data)[:, :, layer])
        dcv.engine.sgm(dcv, self._p1, self._p2, is_max, invalid_cost, self._overcounting)
        cv.attrs["optimization"] = "sgm"
        cv.attrs["cmax"] = 8.0 * (cmax + self._p2)  # upper bound of the 8-path sum
        return cv
