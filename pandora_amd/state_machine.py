"""PandoraMachine - the reference's step sequencer (state_machine.py:70-1072) for the hot path.

A dependency-free finite state machine (the reference builds on the `transitions` package, absent
here): three states ``begin -> cost_volume -> disp_map``; triggers are the pipeline keys' prefixes
(``"filter.after"`` -> ``filter``, state_machine.py:706-717).  Hot-path triggers are implemented
(matching_cost, aggregation, optimization, disparity, refinement) plus the validation step (SURVEY 8f
N1: cross_checking_accurate / cross_checking_fast, with the left/right duplication of every step the
reference performs, state_machine.py:311-364, :379-380, :418-419, :436-448, :490-491, :493-519); the
median / bilateral / disparity_denoiser filters (N2; state_machine.py:449-473) and the multiscale loop (N3;
fixed_zoom_pyramid, state_machine.py:521-556, images without masks); the others of the reference
(semantic_segmentation) are outside this
build's scope: an unknown filter raises the reference's KeyError, an unknown step ``MachineError``.
"""
import logging

import numpy as np

from . import (aggregation, cost_volume_confidence, disparity, filter, matching_cost, multiscale, optimization, refinement,
               validation)
from .criteria import validity_mask
from .margins import GlobalMargins
from .dataset import DataArray, Dataset


class MachineError(Exception):
    """An illegal transition (same role as transitions.MachineError)."""


class PandoraMachine:
    _transitions_run = {
        # trigger: (source, dest, prepare, before)
        "matching_cost": ("begin", "cost_volume", "matching_cost_prepare", "matching_cost_run"),
        "aggregation": ("cost_volume", "cost_volume", None, "aggregation_run"),
        "optimization": ("cost_volume", "cost_volume", None, "optimization_run"),
        "cost_volume_confidence": ("cost_volume", "cost_volume", None, "cost_volume_confidence_run"),
        "disparity": ("cost_volume", "disp_map", None, "disparity_run"),
        "refinement": ("disp_map", "disp_map", None, "refinement_run"),
        "validation": ("disp_map", "disp_map", None, "validation_run"),
        "filter": ("disp_map", "disp_map", None, "filter_run"),
        # conditional: on the last scale the trigger does nothing and the state stays disp_map (state_machine.py:125-133)
        "multiscale": ("disp_map", "begin", None, "run_multiscale"),
    }
    _transitions_check = {
        "check_matching_cost": ("begin", "cost_volume", "matching_cost_check_conf"),
        "check_aggregation": ("cost_volume", "cost_volume", "aggregation_check_conf"),
        "check_optimization": ("cost_volume", "cost_volume", "optimization_check_conf"),
        "check_cost_volume_confidence": ("cost_volume", "cost_volume", "cost_volume_confidence_check_conf"),
        "check_disparity": ("cost_volume", "disp_map", "disparity_check_conf"),
        "check_refinement": ("disp_map", "disp_map", "refinement_check_conf"),
        "check_validation": ("disp_map", "disp_map", "validation_check_conf"),
        "check_filter": ("disp_map", "disp_map", "filter_check_conf"),
        "check_multiscale": ("disp_map", "disp_map", "multiscale_check_conf"),  # state_machine.py:191-198
    }
    _out_of_scope = ("semantic_segmentation",)

    def __init__(self):
        self.left_img = None
        self.right_img = None
        self.disp_min = None
        self.disp_max = None
        self.right_disp_min = None
        self.right_disp_max = None
        self.scale_factor = 1
        self.num_scales = 1
        self.current_scale = 0
        self.left_cv = None
        self.margins = GlobalMargins()
        self.right_cv = None
        self.left_disparity = None
        self.right_disparity = None
        self.step = 1
        self.pipeline_cfg = {"pipeline": {}}
        self.right_disp_map = None
        self.matching_cost_ = None
        self.state = "begin"
        self._mode = None  # "run" | "check"

    # -- FSM core ------------------------------------------------------------------------------
    def trigger(self, name, cfg, input_step):
        table = self._transitions_run if self._mode == "run" else self._transitions_check
        if name not in table:
            base = name[len("check_"):] if name.startswith("check_") else name
            if base in self._out_of_scope:
                raise MachineError(f"step '{base}' is outside the hot path implemented by pandora_amd (SURVEY 8)")
            raise MachineError(f"Can't trigger event {name}: unknown step")
        t = table[name]
        if self.state != t[0]:
            raise MachineError(f"Can't trigger event {name} from state {self.state}!")
        if self._mode == "run" and name == "multiscale" and not self.is_not_last_scale():
            return  # condition not met: no callback, no state change
        if self._mode == "run":
            if t[2]:
                getattr(self, t[2])(cfg, input_step)
            getattr(self, t[3])(cfg, input_step)
        else:
            getattr(self, t[2])(cfg, input_step)
        self.state = t[1]

    def is_not_last_scale(self):
        """state_machine.py:1027-1039"""
        return self.current_scale != 0

    def run_prepare(self, cfg, left_img, right_img, scale_factor=None, num_scales=None):
        """state_machine.py:589-692"""
        if num_scales is None or scale_factor is None:
            self.num_scales, self.scale_factor = 1, 1
        else:
            self.num_scales, self.scale_factor = num_scales, scale_factor
        self.dmin_user = self.dmax_user = self.dmin_user_right = self.dmax_user_right = None
        if self.num_scales > 1:
            # coarse-to-fine: pyramids (coarsest first), user ranges divided down to the coarsest scale; every
            # matching_cost_prepare multiplies them back by scale_factor (state_machine.py:635-657)
            self.img_left_pyramid, self.img_right_pyramid = multiscale.prepare_pyramid(left_img, right_img, self.num_scales,
                                                                                      self.scale_factor)
            self.left_img = self.img_left_pyramid.pop(0)
            self.right_img = self.img_right_pyramid.pop(0)
            self.current_scale = self.num_scales - 1
            shrink = float(self.scale_factor ** self.num_scales)
            self.disp_min = np.asarray(left_img["disparity"].sel(band_disp="min").data) / shrink
            self.disp_max = np.asarray(left_img["disparity"].sel(band_disp="max").data) / shrink
            self.dmin_user, self.dmax_user = self.disp_min, self.disp_max
            self.right_disp_min, self.right_disp_max = -self.disp_max, -self.disp_min
            self.dmin_user_right, self.dmax_user_right = self.right_disp_min, self.right_disp_max
            self.left_disparity, self.right_disparity = Dataset(), Dataset()
            self.right_cv = None
            self.right_disp_map = None
            if "validation" in cfg["pipeline"]:
                self.right_disp_map = cfg["pipeline"]["validation"]["validation_method"]
            self.state = "begin"
            self._mode = "run"
            return
        self.current_scale = 0
        self.left_img, self.right_img = left_img, right_img
        self.disp_min = np.asarray(left_img["disparity"].sel(band_disp="min").data)
        self.disp_max = np.asarray(left_img["disparity"].sel(band_disp="max").data)
        # right-side ranges (state_machine.py:659-675): the right image's own grids, else reversed from the left
        if "disparity" in right_img.data_vars:
            self.right_disp_min = np.asarray(right_img["disparity"].sel(band_disp="min").data)
            self.right_disp_max = np.asarray(right_img["disparity"].sel(band_disp="max").data)
        elif "validation" in cfg["pipeline"]:
            self.right_disp_min, self.right_disp_max = matching_cost.AbstractMatchingCost.reverse_disp_range(self.disp_min, self.disp_max)
            right_img.coords["band_disp"] = np.array(["min", "max"])
            right_img["disparity"] = DataArray(np.stack([self.right_disp_min, self.right_disp_max], axis=0),
                                               ("band_disp", "row", "col"), {"band_disp": ["min", "max"]})
        self.left_disparity = Dataset()
        self.right_disparity = Dataset()
        self.right_cv = None
        self.right_disp_map = None
        if "validation" in cfg["pipeline"]:
            self.right_disp_map = cfg["pipeline"]["validation"]["validation_method"]
        self.state = "begin"
        self._mode = "run"

    def run(self, input_step, cfg):
        """state_machine.py:694-720"""
        try:
            trig = input_step.split(".")[0] if len(input_step.split(".")) != 1 else input_step
            self.trigger(trig, cfg, input_step)
        except (MachineError, KeyError, AttributeError):
            logging.error("A problem occurs during Pandora running %s. Be sure of your sequencing", input_step)
            raise

    def run_exit(self):
        """state_machine.py:722-730"""
        self.state = "begin"
        self._mode = None

    # -- run callbacks (state_machine.py:292-490) ----------------------------------------------
    def matching_cost_prepare(self, cfg, input_step):
        self.matching_cost_ = matching_cost.AbstractMatchingCost(**cfg["pipeline"][input_step])
        self.matching_cost_.prefetch(self.left_img, self.right_img)  # the images travel while the grids are scanned
        # matching_cost_prepare and matching_cost_run are the two callbacks of ONE trigger: no user code runs between them, the
        # disparity grids cannot change, one scan of each serves both the sizing of the volume and cv_masked
        self.matching_cost_._grid_memo = {}
        if self.scale_factor != 1:  # (a 4 M-pixel grid times one is 8 ms of host time for nothing)
            self.disp_min = self.disp_min * self.scale_factor
            self.disp_max = self.disp_max * self.scale_factor
        self.left_cv = self.matching_cost_.allocate_cost_volume(self.left_img, (self.disp_min, self.disp_max), cfg)
        self.left_cv = validity_mask(self.left_img, self.right_img, self.left_cv)
        if self.right_disp_map is not None:  # state_machine.py:311-331
            if self.scale_factor != 1:
                self.right_disp_min = self.right_disp_min * self.scale_factor
                self.right_disp_max = self.right_disp_max * self.scale_factor
            if self.right_disp_map == "cross_checking_accurate":
                grids = (self.right_disp_min, self.right_disp_max)
            else:  # fast: sized from the left range so that it matches the reversed left volume
                grids = (-self.disp_max, -self.disp_min)
            self.right_cv = self.matching_cost_.allocate_cost_volume(self.right_img, grids, cfg)
            self.right_cv = validity_mask(self.right_img, self.left_img, self.right_cv)

    def matching_cost_run(self, _, __):
        logging.info("Matching cost computation...")
        try:
            self.left_cv = self.matching_cost_.compute_cost_volume(self.left_img, self.right_img, self.left_cv)
            self.matching_cost_.cv_masked(self.left_img, self.right_img, self.left_cv, self.disp_min, self.disp_max)
            if self.right_disp_map == "cross_checking_accurate":
                self.right_cv = self.matching_cost_.compute_cost_volume(self.right_img, self.left_img, self.right_cv)
                self.matching_cost_.cv_masked(self.right_img, self.left_img, self.right_cv, self.right_disp_min, self.right_disp_max)
        finally:
            self.matching_cost_._grid_memo = None

    def aggregation_run(self, cfg, input_step):
        logging.info("Aggregation computation...")
        aggregation_ = aggregation.AbstractAggregation(**cfg["pipeline"][input_step])
        aggregation_.cost_volume_aggregation(self.left_img, self.right_img, self.left_cv)
        if self.right_disp_map == "cross_checking_accurate":
            aggregation_.cost_volume_aggregation(self.right_img, self.left_img, self.right_cv)

    def optimization_run(self, cfg, input_step):
        logging.info("Cost optimization...")
        optimization_ = optimization.AbstractOptimization(self.left_img, **cfg["pipeline"][input_step])
        self.left_cv = optimization_.optimize_cv(self.left_cv, self.left_img, self.right_img)
        if self.right_disp_map == "cross_checking_accurate":
            self.right_cv = optimization_.optimize_cv(self.right_cv, self.right_img, self.left_img)

    def cost_volume_confidence_run(self, cfg, input_step):
        """state_machine.py:558-587 (N4: ambiguity, risk and interval_bounds on the device, std_intensity on the host)"""
        logging.info("Cost volume confidence computation...")
        cfg["pipeline"][input_step]["indicator"] = ""
        if len(input_step.split(".")) == 2:
            cfg["pipeline"][input_step]["indicator"] = "." + input_step.split(".")[1]
        confidence_ = cost_volume_confidence.AbstractCostVolumeConfidence(**cfg["pipeline"][input_step])
        self.left_disparity, self.left_cv = confidence_.confidence_prediction(self.left_disparity, self.left_img, self.right_img,
                                                                              self.left_cv)
        if self.right_disp_map == "cross_checking_accurate":
            self.right_disparity, self.right_cv = confidence_.confidence_prediction(self.right_disparity, self.right_img,
                                                                                    self.left_img, self.right_cv)

    def disparity_run(self, cfg, input_step):
        logging.info("Disparity computation...")
        disparity_ = disparity.AbstractDisparity(**cfg["pipeline"][input_step])
        self.left_disparity = disparity_.to_disp(self.left_cv, self.left_img, self.right_img)
        if self.right_disp_map == "cross_checking_accurate":
            self.right_disparity = disparity_.to_disp(self.right_cv, self.right_img, self.left_img)
        elif self.right_disp_map == "cross_checking_fast":
            # state_machine.py:438-448: the right volume is the re-indexed left one, built on the device at WTA time
            self.right_cv.data_vars["cost_volume"] = matching_cost.AbstractMatchingCost.reverse_cost_volume(
                self.left_cv["cost_volume"], np.nanmin(-self.disp_max))
            self.right_cv.attrs["type_measure"] = self.left_cv.attrs["type_measure"]
            self.right_cv.attrs["cmax"] = self.left_cv.attrs["cmax"]
            self.right_disparity = disparity_.to_disp(self.right_cv, self.right_img, self.left_img)

    def run_multiscale(self, cfg, input_step):
        """state_machine.py:521-556: disparity ranges of the next (finer) scale from this scale's disparity maps."""
        logging.info("Disparity range computation...")
        multiscale_ = multiscale.AbstractMultiscale(self.left_img, self.right_img, **cfg["pipeline"][input_step])
        self.dmin_user = self.dmin_user * self.scale_factor
        self.dmax_user = self.dmax_user * self.scale_factor
        self.disp_min, self.disp_max = multiscale_.disparity_range(self.left_disparity, self.dmin_user, self.dmax_user)
        self.left_disparity = None
        if self.right_disp_map is not None:
            self.dmin_user_right = self.dmin_user_right * self.scale_factor
            self.dmax_user_right = self.dmax_user_right * self.scale_factor
            self.right_disp_min, self.right_disp_max = multiscale_.disparity_range(self.right_disparity, self.dmin_user_right,
                                                                                   self.dmax_user_right)
            self.right_disparity = None
        self.left_img = self.img_left_pyramid.pop(0)
        self.right_img = self.img_right_pyramid.pop(0)
        self.current_scale = self.current_scale - 1

    def multiscale_check_conf(self, cfg, input_step):
        """state_machine.py:924-935"""
        m = multiscale.AbstractMultiscale(self.left_img, self.right_img, **cfg[input_step])
        self.pipeline_cfg["pipeline"][input_step] = m.cfg

    def _image_shape(self):
        return (self.left_img.sizes["row"], self.left_img.sizes["col"]) if self.left_img is not None else None

    def filter_run(self, cfg, input_step):
        """state_machine.py:449-473"""
        logging.info("Disparity filtering...")
        filter_ = filter.AbstractFilter(cfg=cfg["pipeline"][input_step], image_shape=self._image_shape(), step=self.step)
        filter_.filter_disparity(self.left_disparity, self.left_img)
        if self.right_disp_map == "cross_checking_accurate" or (
                self.right_disp_map == "cross_checking_fast"
                and cfg["pipeline"][input_step]["filter_method"] != "median_for_intervals"):
            filter_.filter_disparity(self.right_disparity, self.right_img)

    def refinement_run(self, cfg, input_step):
        logging.info("Subpixel refinement...")
        refinement_ = refinement.AbstractRefinement(**cfg["pipeline"][input_step])
        refinement_.subpixel_refinement(self.left_cv, self.left_disparity)
        if self.right_disp_map is not None:
            refinement_.subpixel_refinement(self.right_cv, self.right_disparity)

    def validation_run(self, cfg, input_step):
        """state_machine.py:493-519"""
        logging.info("Validation...")
        validation_ = validation.AbstractValidation(**cfg["pipeline"][input_step])
        self.left_disparity = validation_.disparity_checking(self.left_disparity, self.right_disparity)
        if self.right_disp_map is not None:
            self.right_disparity = validation_.disparity_checking(self.right_disparity, self.left_disparity)
            if "interpolated_disparity" in cfg["pipeline"][input_step]:  # mismatches and occlusions get a value
                interpolate_ = validation.AbstractInterpolation(**cfg["pipeline"][input_step])
                interpolate_.interpolated_disparity(self.left_disparity)
                interpolate_.interpolated_disparity(self.right_disparity)
        if self.right_disp_map == "cross_checking_fast":
            # do not hand incomplete right-side data to the user
            self.right_disparity = Dataset()
            self.right_cv = None

    # -- configuration pass (state_machine.py:732-1008) ----------------------------------------
    def matching_cost_check_conf(self, cfg, input_step):
        m = matching_cost.AbstractMatchingCost(**cfg[input_step])
        self.pipeline_cfg["pipeline"][input_step] = m.cfg
        self.step = m._step_col
        self.margins.add_cumulative(input_step, m.margins)
        for img in (self.left_img, self.right_img):  # state_machine.py:748-759
            if img is not None:
                bands = list(img.coords["band_im"]) if "band_im" in img.coords else [None]
                self.check_band_pipeline(bands, cfg[input_step]["matching_cost_method"], m.cfg["band"])

    @staticmethod
    def check_band_pipeline(band_list, step, band_used):
        """state_machine.py:1042-1072: a step's band parameter against the bands of an input image"""
        if not band_used:
            if len(band_list) != 1:
                raise AttributeError(f"Missing band instantiate on {step} step : input image is multiband")
        elif isinstance(band_used, (list, dict)):
            for band in (band_used.values() if isinstance(band_used, dict) else band_used):
                if band not in band_list:
                    raise AttributeError(f"Wrong band instantiate on {step} step: {band} not in input image")
        elif isinstance(band_used, str):
            if band_used not in band_list:
                raise AttributeError(f"Wrong band instantiate on {step} step: {band_used} not in input image")
        else:
            raise TypeError(f"Wrong type for band {band_used} used in {step}")

    def aggregation_check_conf(self, cfg, input_step):
        a = aggregation.AbstractAggregation(**cfg[input_step])
        self.pipeline_cfg["pipeline"][input_step] = a.cfg
        self.margins.add_cumulative(input_step, a.margins)

    def optimization_check_conf(self, cfg, input_step):
        if self.step != 1:  # state_machine.py:868-870
            raise AttributeError("For performing the SGM optimization step, step attribute must be equal to 1")
        o = optimization.AbstractOptimization(self.left_img, **cfg[input_step])
        self.pipeline_cfg["pipeline"][input_step] = o.cfg
        self.margins.add_cumulative(input_step, o.margins)

    def cost_volume_confidence_check_conf(self, cfg, input_step):
        """state_machine.py:937-948"""
        c = cost_volume_confidence.AbstractCostVolumeConfidence(**cfg[input_step])
        self.pipeline_cfg["pipeline"][input_step] = c.cfg

    def disparity_check_conf(self, cfg, input_step):
        d = disparity.AbstractDisparity(**cfg[input_step])
        self.pipeline_cfg["pipeline"][input_step] = d.cfg
        self.margins.add_cumulative(input_step, d.margins)

    def filter_check_conf(self, cfg, input_step):
        """state_machine.py:775-792"""
        f = filter.AbstractFilter(cfg=dict(cfg[input_step]), image_shape=self._image_shape(), step=self.step)
        self.pipeline_cfg["pipeline"][input_step] = f.cfg
        self.margins.add_non_cumulative(input_step, f.margins)

    def refinement_check_conf(self, cfg, input_step):
        r = refinement.AbstractRefinement(**cfg[input_step])
        self.pipeline_cfg["pipeline"][input_step] = r.cfg
        self.margins.add_cumulative(input_step, r.margins)

    def validation_check_conf(self, cfg, input_step):
        """state_machine.py:894-922"""
        v = validation.AbstractValidation(**cfg[input_step])
        self.pipeline_cfg["pipeline"][input_step] = v.cfg
        if "interpolated_disparity" in v.cfg:
            validation.AbstractInterpolation(**cfg[input_step])
        self.right_disp_map = v.cfg["validation_method"]
        if self.left_img is not None and self.right_img is not None:
            ds_left = self.left_img.attrs.get("disparity_source")
            ds_right = self.right_img.attrs.get("disparity_source")
            if isinstance(ds_left, list) and isinstance(ds_right, list):
                if ds_left[0] != -ds_right[1] or ds_left[1] != -ds_right[0]:
                    raise AttributeError("disp_min != -disp_right_max or disp_max != -disp_right_min")
            elif isinstance(ds_left, str) and isinstance(ds_right, str):
                logging.warning("The right disp will be ignored, and instead computed from the left disp.")

    def check_conf(self, cfg, img_left=None, img_right=None, right_left_img_check=False):
        """state_machine.py:950-1008: dry-run the FSM with the check_* triggers; returns the checked
        pipeline configuration (defaults filled in)."""
        self.left_img, self.right_img = img_left, img_right
        self.state = "begin"
        self._mode = "check"
        self.pipeline_cfg = {"pipeline": {}}
        self.margins = GlobalMargins()  # state_machine.py:260: rebuilt by every check
        for input_step in list(cfg["pipeline"]):
            trig = "check_" + input_step.split(".")[0]
            try:
                self.trigger(trig, cfg["pipeline"], input_step)
            except (MachineError, KeyError, AttributeError) as err:  # state_machine.py:992-993: one error type, one message
                logging.error("Problem during Pandora checking configuration steps sequencing. "
                              "Check your configuration file.")
                raise MachineError(f"A problem occurs during Pandora checking. Be sure of your sequencing ({err})") from err
        self.state = "begin"
        self._mode = None
        return self.pipeline_cfg
