"""Image datasets from files (reference: img_tools.py:101-162 add_disparity / add_disparity_grid, :234-316
add_no_data / add_mask, :345-437 create_dataset_from_inputs).  Host-side I/O: the reference reads with rasterio, which is
not in this image; Pillow reads the single-band PNG / TIFF files, tiff_reader.py the multi-sample TIFFs (multiband images,
two-band disparity grids).  Classification / segmentation layers, ROI windows and georeferencing (crs / transform stay None)
are outside this build and refused loudly."""
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


def add_disparity_grid(dataset, disparity_grid=None, disparity_source="xr.Dataset"):
    """img_tools.py:138-162: ``disparity_grid`` is (2, row, col) = [min, max]."""
    if disparity_grid is not None:
        data = disparity_grid.data if hasattr(disparity_grid, "data") and not isinstance(disparity_grid, np.ndarray) else disparity_grid
        dataset.coords["band_disp"] = np.array(["min", "max"])
        dataset["disparity"] = DataArray(np.asarray(data), ("band_disp", "row", "col"), {"band_disp": ["min", "max"]})
        dataset.attrs["disparity_source"] = disparity_source
    return dataset


def add_disparity(dataset, disparity, window=None):
    """img_tools.py:101-135: a [min, max] pair becomes two constant grids; a path to a two-band grid file needs rasterio."""
    if window is not None:
        raise NotImplementedError("ROI windows are out of scope of pandora_amd")
    if disparity is None:
        dataset.attrs["disparity_source"] = None
        return dataset
    if isinstance(disparity, str):  # a two-band raster: band 1 = minimum, band 2 = maximum (img_tools.py:124-125)
        grids, _ = _read_raster(disparity)
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
    if window is not None:
        raise NotImplementedError("ROI windows are out of scope of pandora_amd")
    if mask is None and no_data_pixels[0].size == 0:
        return dataset
    valid, nodata = dataset.attrs["valid_pixels"], dataset.attrs["no_data_mask"]
    msk = np.full((height, width), valid).astype(np.int16)
    if mask is not None:
        input_mask = _read_band(mask) if isinstance(mask, str) else np.asarray(mask)
        msk[np.where(input_mask != valid)] = valid + nodata + 1
    msk[(no_data_pixels[-2], no_data_pixels[-1])] = int(nodata)
    dataset["msk"] = DataArray(msk, ("row", "col"))
    return dataset


def create_dataset_from_inputs(input_config, roi=None):
    """img_tools.py:345-437 for single-band images: {"img": path, "nodata": value, "mask": path or None, "disp": [min, max]}
    -> Dataset{im float32, msk int16 (when needed), disparity} with attrs crs / transform (None: no georeferencing without
    rasterio) / valid_pixels 0 / no_data_mask 1 / no_data_img / disparity_source."""
    if roi is not None:
        raise NotImplementedError("ROI windows are out of scope of pandora_amd")
    params = {"mask": None, "classif": None, "segm": None, "edges": None}
    params.update(input_config)
    for layer in ("classif", "segm", "edges"):
        if params[layer] is not None:
            raise NotImplementedError(f"the '{layer}' layer is out of scope of pandora_amd")
    data, names = _read_raster(params["img"])
    data = data.astype(np.float32)
    ny_, nx_ = data.shape[-2:]
    if data.ndim == 2:
        dataset = Dataset({"im": (("row", "col"), data)}, coords={"row": np.arange(ny_), "col": np.arange(nx_)})
    else:  # img_tools.py:388-398: band names come from the image metadata
        dataset = Dataset({"im": (("band_im", "row", "col"), data)},
                          coords={"band_im": np.asarray(names if names else [None] * data.shape[0], dtype=object), "row": np.arange(ny_), "col": np.arange(nx_)})
    dataset.attrs.update({"crs": None, "transform": None, "valid_pixels": 0, "no_data_mask": 1})
    dataset.attrs["disparity_source"] = None
    if "disp" in params:
        add_disparity(dataset, params["disp"])
    no_data = params["nodata"]
    if np.isnan(no_data):
        no_data_pixels = np.where(np.isnan(data))
    elif np.isinf(no_data):
        no_data_pixels = np.where(np.isinf(data))
    else:
        no_data_pixels = np.where(data == no_data)
    add_no_data(dataset, no_data, no_data_pixels)
    return add_mask(dataset, params["mask"], no_data_pixels, nx_, ny_)
