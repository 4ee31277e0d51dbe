"""pandora_amd - MI355X-native stereo cost-volume engine behind Pandora's plugin API.

Host side (Python, mirrors the reference's interface for the hot path only):
    matching_cost / aggregation / optimization / disparity / refinement plugin registries,
    PandoraMachine step sequencing, criteria.validity_mask, constants.
Device side: hand-written HIP kernels for gfx950 in ``csrc/`` behind the C ABI of
``include/pandora_amd.h`` (``libpandora_amd.so``, loaded with ctypes; no PyTorch in the data path).
There is NO CPU fallback: every plugin raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"


def run(pandora_machine, img_left, img_right, cfg):
    """The reference's pandora.run (__init__.py:50-124): every key of cfg["pipeline"] in order on the machine, once per
    scale (coarse to fine when a multiscale step is configured); returns (left, right) disparity datasets."""
    from .multiscale import read_multiscale_params

    num_scales, scale_factor = read_multiscale_params(img_left, img_right, cfg)
    pandora_machine.run_prepare(cfg, img_left, img_right, scale_factor, num_scales)
    for _ in range(pandora_machine.num_scales):
        for step in list(cfg["pipeline"]):
            pandora_machine.run(step, cfg)
            if pandora_machine.state == "begin":  # the multiscale step moved the machine to the next scale
                break
    pandora_machine.run_exit()
    return pandora_machine.left_disparity, pandora_machine.right_disparity


def import_plugin():
    """The reference's pandora.import_plugin (__init__.py:141-148): load every entry point of the
    group "pandora.plugin" so that external plugins register themselves."""
    from importlib.metadata import entry_points

    eps = entry_points()
    group = eps.select(group="pandora.plugin") if hasattr(eps, "select") else eps.get("pandora.plugin", [])
    for ep in group:
        ep.load()


def check_datasets(left, right):
    """check_configuration.py:112-167: images present, a disparity range on the left with min <= max, same shapes."""
    import numpy as np

    for ds in (left, right):
        if "im" not in ds.data_vars:
            raise AttributeError("User must provide an image im")
        if "msk" in ds.data_vars and np.asarray(ds["im"].data).shape[-2:] != np.asarray(ds["msk"].data).shape[-2:]:
            raise ValueError(" im and msk must have the same shape")
        missing = {"no_data_img", "valid_pixels", "no_data_mask", "crs", "transform"} - set(ds.attrs)
        if missing:
            raise AttributeError(f"User must provide the {missing} attribute(s)")
    if "disparity" not in left.data_vars:
        raise AttributeError("left dataset must have disparity DataArray")
    grids = np.asarray(left["disparity"].data)
    if (grids[0] > grids[1]).any():
        raise AttributeError("Disp_max grid must be bigger than Disp_min grid for each pixel")
    if np.asarray(left["im"].data).shape[-2:] != np.asarray(right["im"].data).shape[-2:]:
        raise AttributeError("left and right datasets must have the same shape")


def main(cfg_path, output, verbose=False):
    """The reference's pandora.main (__init__.py:151-202) for single-band inputs: read the JSON configuration, build the
    two image datasets, check, run the pipeline on the GPU, write left_/right_ disparity.tif, validity_mask.tif,
    confidence_measure.tif and cfg/config.json into ``output``."""
    import json
    import logging

    from . import common
    from .img_tools import create_dataset_from_inputs
    from .state_machine import PandoraMachine

    with open(cfg_path) as f:
        user_cfg = json.load(f)
    logging.basicConfig(level=logging.INFO if verbose else logging.WARNING)
    import_plugin()
    inputs = {"left": {"nodata": -9999, "mask": None, "classif": None, "segm": None, "edges": None},
              "right": {"nodata": -9999, "mask": None, "classif": None, "segm": None, "edges": None, "disp": None}}
    for side in inputs:  # check_configuration.py:634-652 default_short_configuration_input
        inputs[side].update(user_cfg["input"][side])
    if inputs["right"]["disp"] is None and not isinstance(inputs["left"]["disp"], str):
        inputs["right"]["disp"] = [-inputs["left"]["disp"][1], -inputs["left"]["disp"][0]]
    img_left = create_dataset_from_inputs(input_config=inputs["left"])
    img_right = create_dataset_from_inputs(input_config=inputs["right"])
    check_datasets(img_left, img_right)
    machine = PandoraMachine()
    cfg = {"input": inputs, "pipeline": user_cfg["pipeline"]}
    cfg["pipeline"] = machine.check_conf(cfg, img_left, img_right)["pipeline"]
    left, right = run(machine, img_left, img_right, cfg)
    common.save_results(left, right, output)
    cfg["margins"] = machine.margins.to_dict()  # __init__.py:197-198
    common.save_config(output, cfg)
    return left, right
