"""pandora_amd - MI355X-native stereo cost-volume engine behind Pandora's plugin API.

Host side (Python, mirrors the reference's interface for the hot path only):
    matching_cost / aggregation / optimization / disparity / refinement plugin registries,
    PandoraMachine step sequencing, criteria.validity_mask, constants.
Device side: hand-written HIP kernels for gfx950 in ``csrc/`` behind the C ABI of
``include/pandora_amd.h`` (``libpandora_amd.so``, loaded with ctypes; no PyTorch in the data path).
There is NO CPU fallback: every plugin raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
