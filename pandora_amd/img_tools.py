"""Image datasets from files (reference: img_tools.py:101-162 add_disparity / add_disparity_grid, :234-316
add_no_data / add_mask, :345-437 create_dataset_from_inputs).  Host-side I/O: the reference reads with rasterio, which is
not in this image; Pillow reads the single-band PNG / TIFF files, tiff_reader.py the multi-sample TIFFs (multiband images,
two-band disparity grids); a GeoTIFF's georeferencing tags ride along in attrs["crs"] / attrs["transform"] and are written back
by common.save_results.  ROI windows (get_window) are cut out of the decoded raster.  Classification / segmentation / edge layers
(img_tools.py:165-231) ride along as int16 variables: the SGM step's geometric_prior reads them."""
from collections import namedtuple

import numpy as np

from .dataset import DataArray, Dataset


def _read_raster(path):
    """-> (array (row, col) or (band, row, col), band descriptions or None): Pillow for the single-band PNG / TIFF files,
    the small baseline-TIFF reader of tiff_reader.py for multi-sample rasters as GDAL writes them."""
    from PIL import Image, UnidentifiedImageError

    try:
        with Image.open(path) as im:
            if getattr(im, "n_frames", 1) == 1 and im.mode in ("F", "L", "I", "I;16", "I;16B", "1", "P"):
                return np.array(im), None
    except (UnidentifiedImageError, ValueError, OSError, SyntaxError):  # a layout Pillow cannot decode: try the TIFF reader below
        pass
    from .tiff_reader import read_tiff

    return read_tiff(path)


def _read_band(path):
    data, _ = _read_raster(path)
    if data.ndim != 2:
        raise ValueError(f"{path}: a single-band raster is expected")
    return data


Window = namedtuple("Window", ["col_off", "row_off", "width", "height"])  # rasterio.windows.Window's four numbers


def get_window(roi, width, height):
    """img_tools.py:61-98: the window of a region of interest {"col": {"first", "last"}, "row": {"first", "last"},
    "margins": [left, up, right, down]} with its margins, clipped to the image; ValueError when it lies outside."""
    col_off = max(roi["col"]["first"] - roi["margins"][0], 0)
    row_off = max(roi["row"]["first"] - roi["margins"][1], 0)
    roi_width = roi["col"]["last"] - col_off + roi["margins"][2] + 1
    roi_height = roi["row"]["last"] - row_off + roi["margins"][3] + 1
    if col_off > width or row_off > height or (col_off + roi_width) < 0 or (row_off + roi_height) < 0:
        raise ValueError("Roi specified is outside the image")
    if col_off + roi_width > width:
        roi_width = width - col_off
    if row_off + roi_height > height:
        roi_height = height - row_off
    return Window(col_off, row_off, roi_width, roi_height)


def _cut(data, window):
    """rows / columns of the window out of a (row, col) or (band, row, col) raster (the whole raster was decoded: windowed
    decoding is an I/O optimisation this reader does without)"""
    if window is None:
        return data
    return np.ascontiguousarray(data[..., window.row_off:window.row_off + window.height, window.col_off:window.col_off + window.width])


def add_disparity_grid(dataset, disparity_grid=None, disparity_source="xr.Dataset"):
    """img_tools.py:138-162: ``disparity_grid`` is (2, row, col) = [min, max]."""
    if disparity_grid is not None:
        data = disparity_grid.data if hasattr(disparity_grid, "data") and not isinstance(disparity_grid, np.ndarray) else disparity_grid
        dataset.coords["band_disp"] = np.array(["min", "max"])
        dataset["disparity"] = DataArray(np.asarray(data), ("band_disp", "row", "col"), {"band_disp": ["min", "max"]})
        dataset.attrs["disparity_source"] = disparity_source
    return dataset


def add_disparity(dataset, disparity, window=None):
    """img_tools.py:101-135: a [min, max] pair becomes two constant grids; a path names a two-band grid file (read inside `window`)."""
    if disparity is None:
        dataset.attrs["disparity_source"] = None
        return dataset
    if isinstance(disparity, str):  # a two-band raster: band 1 = minimum, band 2 = maximum (img_tools.py:124-125)
        grids, _ = _read_raster(disparity)
        grids = _cut(grids, window)
        if grids.ndim != 3 or grids.shape[0] != 2 or grids.shape[1:] != (dataset.sizes["row"], dataset.sizes["col"]):
            raise ValueError(f"{disparity}: a disparity grid file holds two bands (min, max) of the image's size")
        return add_disparity_grid(dataset, grids.astype(np.float32), disparity)
    H, W = dataset.sizes["row"], dataset.sizes["col"]
    grids = np.array([np.full((H, W), disparity[0]), np.full((H, W), disparity[1])])
    return add_disparity_grid(dataset, grids, disparity)


def add_no_data(dataset, no_data, no_data_pixels):
    """img_tools.py:234-254: NaN / inf no-data values are replaced by -9999 in the image (their positions live in msk)."""
    if no_data_pixels[0].size != 0 and (np.isnan(no_data) or np.isinf(no_data)):
        dataset["im"].data[no_data_pixels] = -9999
        no_data = -9999
    dataset.attrs.update({"no_data_img": no_data})
    return dataset


def add_mask(dataset, mask, no_data_pixels, width, height, window=None):
    """img_tools.py:257-316: msk = valid_pixels everywhere, valid_pixels + no_data_mask + 1 where the input mask is not
    valid_pixels, no_data_mask on the no-data pixels (which win over the input mask); no mask at all when there is neither
    an input mask nor a no-data pixel."""
    if mask is None and no_data_pixels[0].size == 0:
        return dataset
    valid, nodata = dataset.attrs["valid_pixels"], dataset.attrs["no_data_mask"]
    msk = np.full((height, width), valid).astype(np.int16)
    if mask is not None:
        input_mask = _cut(_read_band(mask), window) if isinstance(mask, str) else np.asarray(mask)
        msk[np.where(input_mask != valid)] = valid + nodata + 1
    msk[(no_data_pixels[-2], no_data_pixels[-1])] = int(nodata)
    dataset["msk"] = DataArray(msk, ("row", "col"))
    return dataset


def create_dataset_from_inputs(input_config, roi=None):
    """img_tools.py:345-437 for single-band images: {"img": path, "nodata": value, "mask": path or None, "disp": [min, max]}
    -> Dataset{im float32, msk int16 (when needed), disparity} with attrs crs / transform (the file's georeferencing, None
    without any) / valid_pixels 0 / no_data_mask 1 / no_data_img / disparity_source.  `roi`: only the window of get_window is kept
    and the row / col coordinates start at its offset (img_tools.py:377-398)."""
    params = {"mask": None, "classif": None, "segm": None, "edges": None}
    params.update(input_config)
    data, names = _read_raster(params["img"])
    data = data.astype(np.float32)
    window = get_window(roi, data.shape[-1], data.shape[-2]) if roi else None
    col_off, row_off = (window.col_off, window.row_off) if roi else (0, 0)
    data = _cut(data, window)
    ny_, nx_ = data.shape[-2:]
    rows, cols = np.arange(row_off, ny_ + row_off), np.arange(col_off, nx_ + col_off)
    if data.ndim == 2:
        dataset = Dataset({"im": (("row", "col"), data)}, coords={"row": rows, "col": cols})
    else:  # img_tools.py:388-398: band names come from the image metadata
        dataset = Dataset({"im": (("band_im", "row", "col"), data)},
                          coords={"band_im": np.asarray(names if names else [None] * data.shape[0], dtype=object), "row": rows, "col": cols})
    from .tiff_reader import read_georeferencing

    crs, transform = read_georeferencing(params["img"])  # (the whole image's, also for a window: img_tools.py:400-403)
    dataset.attrs.update({"crs": crs, "transform": transform if crs is not None else None, "valid_pixels": 0, "no_data_mask": 1})
    dataset.attrs["disparity_source"] = None
    if "disp" in params:
        add_disparity(dataset, params["disp"], window)
    no_data = params["nodata"]
    if np.isnan(no_data):
        no_data_pixels = np.where(np.isnan(data))
    elif np.isinf(no_data):
        no_data_pixels = np.where(np.isinf(data))
    else:
        no_data_pixels = np.where(data == no_data)
    add_no_data(dataset, no_data, no_data_pixels)
    add_layers(dataset, params, window)
    return add_mask(dataset, params["mask"], no_data_pixels, nx_, ny_, window)


def add_layers(dataset, params, window=None):
    """img_tools.py:165-231 add_classif / add_segm / add_edges: "classif" int16 (band_classif, row, col) with the band names of
    the file as the band_classif coordinate, "segm" and "edges" int16 (row, col) from the first band."""
    ny_, nx_ = np.asarray(dataset["im"].data).shape[-2:]
    if params.get("classif") is not None:
        data, names = _read_raster(params["classif"])
        data = _cut(data if data.ndim == 3 else data[np.newaxis], window).astype(np.int16)
        if data.shape[-2:] != (ny_, nx_):
            raise ValueError("the classification must have the image's dimensions (plugin_libsgm.rst:58)")
        dataset.coords["band_classif"] = np.asarray(names if names else [None] * data.shape[0], dtype=object)
        dataset["classif"] = DataArray(data, ("band_classif", "row", "col"))
    for layer in ("segm", "edges"):
        if params.get(layer) is not None:
            data, _ = _read_raster(params[layer])
            data = _cut(data if data.ndim == 2 else data[0], window).astype(np.int16)
            if data.shape != (ny_, nx_):
                raise ValueError(f"the {layer} layer must have the image's dimensions (plugin_libsgm.rst:58)")
            dataset[layer] = DataArray(data, ("row", "col"))
    return dataset
