"""8-path Semi-Global Matching plugin, registered as "sgm".

In the reference this step is the EXTERNAL plugin pandora_plugin_libsgm==1.5.7 (pyproject.toml:59-61)
wrapping CNES libSGM; only its configuration surface is documented in the reference tree
(docs/source/userguide/plugins/plugin_libsgm.rst:88-209).  This class keeps that surface
(overcounting, min_cost_paths, penalty{penalty_method, P1, P2, p2_method, alpha, beta, gamma}) for the
"sgm_penalty" method and runs this build's own SGM definition (oracle/oracle.c orc_sgm) as
HIP kernels.  Numerical parity with libSGM is UNPINNED (no reference test pins it).

p2_method (plugin_libsgm.rst:20-27, 168-290): "constant"; "negativeGradient": P2 = -alpha * |I(p) - I(p-r)| + gamma;
"inverseGradient": P2 = alpha / (|I(p) - I(p-r)| + beta) + gamma, with I the left image and p-r the pixel before p on the
path.  This build's reading of what the documentation leaves open: the configured P2 is the floor of the adaptive value
(P2 > P1 keeps the recurrence's order of penalties), the gradient of a path's first pixel is 0 (its update ignores P2).
The maps are 2-D host work (numpy), the recurrence runs on the device (pmx_sgm_p2maps).

geometric_prior (3SGM's piecewise optimisation, plugin_libsgm.rst:49-78): "For each segment, optimization will only be applied
inside this segment"; "when using edges, the optimization paths will stop at the first edge found".  Read as: a path that would
step from p-r to p across a segment border starts again at p (L_r(p, d) = C(p, d), as at the image border) - for "segm" when the
two pixels carry different values, for "classif" when they differ in the selected classes, for "edges" when either of them is an
edge pixel (value > 0).  No kernel of its own: a restart IS the update with P2 = 0 (min(L(p-r, d), ..., M + 0) = M, so
L = C + (M - M)), bit for bit, so the cut (pixel, direction) pairs become zeros of the penalty maps.  "internal" (the default) cuts
nothing, as in the plugin.  UNPINNED like everything SGM.
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
        self._p2_method = pen["p2_method"]
        self._alpha, self._beta, self._gamma = float(pen.get("alpha", 1.0)), float(pen.get("beta", 1)), float(pen.get("gamma", 1))
        self._overcounting = bool(self.cfg["overcounting"])
        self._use_confidence = self.cfg.get("use_confidence") or None
        prior = self.cfg.get("geometric_prior") or {"source": "internal"}
        self._prior_source, self._prior_classes = prior["source"], list(prior.get("classes", []))

    def check_conf(self, **cfg):
        cfg.setdefault("overcounting", self._OVERCOUNTING)
        cfg.setdefault("min_cost_paths", self._MIN_COST_PATH)
        pen = dict(cfg.get("penalty") or {})
        pen.setdefault("penalty_method", "sgm_penalty")
        pen.setdefault("p2_method", "constant")
        pen.setdefault("P1", self._P1)
        pen.setdefault("P2", self._P2)
        if pen["p2_method"] in ("negativeGradient", "inverseGradient"):  # defaults of plugin_libsgm.rst:215-290
            pen.setdefault("alpha", 1.0)
            pen.setdefault("gamma", 1)
            if pen["p2_method"] == "inverseGradient":
                pen.setdefault("beta", 1)
        cfg["penalty"] = pen
        if cfg.get("optimization_method") != "sgm":
            raise ConfigError("optimization_method must be sgm")
        if pen["penalty_method"] != "sgm_penalty":
            raise ConfigError("pandora_amd implements the sgm_penalty method only (mc_cnn_fast_penalty belongs to the MC-CNN plugin)")
        if pen["p2_method"] not in ("constant", "negativeGradient", "inverseGradient"):
            raise ConfigError("p2_method must be constant, negativeGradient or inverseGradient (plugin_libsgm.rst:163-166)")
        for key in ("alpha", "beta", "gamma"):
            if key in pen and (not isinstance(pen[key], (int, float)) or isinstance(pen[key], bool)):
                raise ConfigError(f"penalty {key} must be a number")
        if pen.get("beta", 1) <= 0:
            raise ConfigError("penalty beta must be > 0 (it keeps alpha / (gradient + beta) finite)")
        if not isinstance(pen["P1"], (int, float)) or not isinstance(pen["P2"], (int, float)) or pen["P1"] <= 0 or pen["P2"] <= pen["P1"]:
            raise ConfigError("penalties must satisfy 0 < P1 < P2 (plugin_libsgm.rst:170-185)")
        if cfg["min_cost_paths"]:
            raise ConfigError("min_cost_paths is not implemented")
        prior = cfg.get("geometric_prior")
        if prior not in (None, False):  # plugin_libsgm.rst:122-137
            if not isinstance(prior, dict) or prior.get("source") not in ("internal", "classif", "segm", "edges"):
                raise ConfigError("geometric_prior is {'source': 'internal' | 'classif' | 'segm' | 'edges'[, 'classes': [...]]}")
            if prior["source"] == "classif":
                if not isinstance(prior.get("classes"), (list, tuple)) or not prior["classes"]:
                    raise ConfigError("geometric_prior with source 'classif' needs the list of 'classes' to use")
            elif set(prior) - {"source"}:
                raise ConfigError("geometric_prior: 'classes' goes with source 'classif' only")
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
        p2_top = self._p2
        cuts = self.path_cuts(img_left) if self._prior_source != "internal" else None
        if self._p2_method == "constant" and cuts is None:
            dcv.engine.sgm(dcv, self._p1, self._p2, is_max, invalid_cost, self._overcounting)
        else:
            if self._p2_method == "constant":
                maps = np.full((8,) + cuts.shape[1:], np.float32(self._p2), np.float32)
            else:
                maps = self.p2_maps(self._band_of(img_left, cv))
            p2_top = float(maps.max())
            if cuts is not None:
                maps[cuts] = 0.0  # the path starts again here: the update with P2 = 0 is L = C (module docstring)
            dcv.engine.sgm_p2maps(dcv, self._p1, maps, is_max, invalid_cost, self._overcounting)
        cv.attrs["optimization"] = "sgm"
        cv.attrs["cmax"] = 8.0 * (cmax + p2_top)  # upper bound of the 8-path sum
        return cv

    # (drow, dcol) of the step from p-r to p, in the definition's order (oracle.c orc_sgm_dirs)
    DIRECTIONS = ((0, 1), (0, -1), (1, 0), (1, 1), (1, -1), (-1, 0), (-1, 1), (-1, -1))

    @staticmethod
    def _band_of(img_left, cv):
        """the band the matching cost was computed on (band_correl), else the image / its first band"""
        im = np.asarray(img_left["im"].data)
        if im.ndim == 2:
            return im
        band = cv.attrs.get("band_correl")
        names = [str(b) for b in np.asarray(img_left.coords["band_im"])]
        return im[names.index(band)] if band in names else im[0]

    def path_cuts(self, img):
        """bool [8][H][W]: True where the path of direction k that arrives at pixel p crosses a border of the geometric prior
        between p - r and p (it then starts again at p).  Needs the layer the configuration names in the image dataset
        (img_tools.py:165-231: "segm" / "edges" int16 (row, col), "classif" int16 (band_classif, row, col))."""
        src = self._prior_source
        if src not in img.data_vars:
            raise AttributeError(f"geometric_prior source '{src}' is not in the image dataset (no {src} layer was given)")
        layer = np.asarray(img[src].data)
        if src == "classif":
            names = [str(b) for b in np.asarray(img.coords["band_classif"])]
            missing = [c for c in self._prior_classes if str(c) not in names]
            if missing:
                raise AttributeError(f"geometric_prior classes {missing} are not bands of the classification {names}")
            label = np.zeros(layer.shape[1:], np.int64)  # which of the selected classes a pixel belongs to, as one number
            for i, c in enumerate(self._prior_classes):
                label |= (layer[names.index(str(c))] != 0).astype(np.int64) << i
        elif src == "edges":
            label = layer > 0
        else:
            label = layer
        H, W = label.shape
        cuts = np.zeros((8, H, W), bool)
        for k, (dr, dc) in enumerate(self.DIRECTIONS):
            r0, r1 = max(dr, 0), H + min(dr, 0)
            c0, c1 = max(dc, 0), W + min(dc, 0)
            here, before = label[r0:r1, c0:c1], label[r0 - dr:r1 - dr, c0 - dc:c1 - dc]
            cuts[k, r0:r1, c0:c1] = (here | before) if src == "edges" else (here != before)
        return cuts

    def p2_maps(self, image):
        """float32 [8][H][W]: the P2 that enters pixel p's update on each path, from the left image's gradient along the path"""
        img = np.asarray(image, np.float32)
        H, W = img.shape
        maps = np.empty((8, H, W), np.float32)
        for k, (dr, dc) in enumerate(self.DIRECTIONS):
            grad = np.zeros((H, W), np.float32)  # |I(p) - I(p - r)|, 0 where p - r is outside the image
            r0, r1 = max(dr, 0), H + min(dr, 0)
            c0, c1 = max(dc, 0), W + min(dc, 0)
            grad[r0:r1, c0:c1] = np.abs(img[r0:r1, c0:c1] - img[r0 - dr:r1 - dr, c0 - dc:c1 - dc])
            if self._p2_method == "negativeGradient":
                adaptive = np.float32(-self._alpha) * grad + np.float32(self._gamma)
            else:
                adaptive = np.float32(self._alpha) / (grad + np.float32(self._beta)) + np.float32(self._gamma)
            maps[k] = np.maximum(adaptive, np.float32(self._p2))
        return maps
