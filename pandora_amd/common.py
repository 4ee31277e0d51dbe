"""Results on disk (reference: common.py:37-181 write_data_array / save_results / save_config,
output_tree_design.py).  Same file names and pixel types as the reference (float32 disparity and confidence, uint16
validity mask); written as plain uncompressed TIFF (tiff_reader.write_tiff) with the input image's georeferencing tags (attrs["crs"])
copied through and the band names as GDAL band descriptions."""
import json
import os

import numpy as np

OTD = {"left_disparity.tif": ".", "right_disparity.tif": ".", "left_confidence_measure.tif": ".", "right_confidence_measure.tif": ".",
       "left_validity_mask.tif": ".", "right_validity_mask.tif": ".", "config.json": "./cfg"}  # output_tree_design.py:27-38


def get_out_file_path(key):
    return os.path.join(OTD[key], key)


def mkdir_p(path):
    os.makedirs(path, exist_ok=True)


def write_data_array(data_array, filename, dtype=np.float32, band_names=None, crs=None, transform=None):
    """common.py:37-96: a (row, col) array is one band, a (row, col, indicator) array one band per indicator, with the band
    names as band descriptions (GDAL's metadata tag, which rasterio's ``descriptions`` reads back)."""
    from .tiff_reader import write_tiff

    data = np.asarray(data_array.data if hasattr(data_array, "data") and not isinstance(data_array, np.ndarray) else data_array)
    mkdir_p(os.path.dirname(os.path.abspath(filename)))
    out_t = np.uint16 if np.dtype(dtype) == np.uint16 else np.float32
    geo = crs.get("geotiff_tags") if isinstance(crs, dict) else None  # (transform is derived from the same tags: nothing more to write)
    if data.ndim == 2:
        write_tiff(filename, data.astype(out_t), geo=geo)
    else:
        write_tiff(filename, np.moveaxis(data, 2, 0).astype(out_t), None if band_names is None else [str(b) for b in band_names], geo=geo)


def save_results(left, right, output):
    """common.py:112-181"""
    mkdir_p(output)
    for side, ds in (("left", left), ("right", right)):
        if side == "right" and len(ds.sizes) == 0:  # no validation step: nothing on the right
            continue
        geo = {"crs": ds.attrs.get("crs"), "transform": ds.attrs.get("transform")}  # common.py:141-181: the image's, passed through
        write_data_array(ds["disparity_map"], os.path.join(output, get_out_file_path(f"{side}_disparity.tif")), **geo)
        if "confidence_measure" in ds.data_vars:
            write_data_array(ds["confidence_measure"], os.path.join(output, get_out_file_path(f"{side}_confidence_measure.tif")),
                             band_names=list(ds.coords["indicator"]), **geo)
        write_data_array(ds["validity_mask"], os.path.join(output, get_out_file_path(f"{side}_validity_mask.tif")), dtype=np.uint16, **geo)


def save_config(output, user_cfg):
    """common.py:184-200"""
    path = os.path.join(output, get_out_file_path("config.json"))
    mkdir_p(os.path.dirname(path))

    def plain(v):
        if isinstance(v, dict):
            return {k: plain(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        if isinstance(v, np.generic):
            return v.item()
        return v

    with open(path, "w") as f:
        json.dump(plain(user_cfg), f, indent=2)
