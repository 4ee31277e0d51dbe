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
