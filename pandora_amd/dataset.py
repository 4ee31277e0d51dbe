"""A dependency-free stand-in for the handful of xarray idioms Pandora's plugin API shows
(``ds["im"].data``, ``ds.attrs``, ``ds.coords["disp"]``, ``ds.sizes``, ``"msk" in ds.data_vars``).
xarray itself is not required (and is absent on the GPU box); ``from_xarray`` / ``to_xarray`` adapt
real xarray datasets when it is importable.

The cost volume variable is special: its ``.data`` lives on the GPU (pandora_amd.engine
.DeviceCostVolume) and is only copied to the host when somebody reads ``.data``.
"""
import numpy as np


class DataArray:
    def __init__(self, data, dims, coords=None):
        self._data = data
        self.dims = tuple(dims)
        self.coords = dict(coords or {})

    @property
    def data(self):
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    @property
    def values(self):
        return self.data

    @property
    def shape(self):
        return self.data.shape

    def sel(self, indexers=None, **kw):
        """Label selection along one coordinate (img["disparity"].sel(band_disp="min"), or xarray's dict form
        cv["confidence_measure"].sel({"indicator": name}))."""
        (dim, label), = {**(indexers or {}), **kw}.items()
        axis = self.dims.index(dim)
        labels = list(self.coords[dim])
        idx = labels.index(label)
        data = self.data[(slice(None),) * axis + (idx,)]  # a view, like xarray's .sel
        dims = tuple(d for d in self.dims if d != dim)
        return DataArray(data, dims, {k: v for k, v in self.coords.items() if k != dim})

    def copy(self, deep=True):
        return DataArray(np.array(self.data, copy=True) if deep else self.data, self.dims, dict(self.coords))


class DeviceVolumeArray(DataArray):
    """``cv["cost_volume"]``: float32 (row, col, disp) resident in HBM; ``.data`` downloads it,
    assigning ``.data`` uploads."""

    def __init__(self, device_cv, coords=None):
        self.device_cv = device_cv
        self.dims = ("row", "col", "disp")
        self.coords = dict(coords or {})

    @property
    def data(self):
        return self.device_cv.to_host()

    @data.setter
    def data(self, value):
        self.device_cv.from_host(value)

    @property
    def shape(self):
        return self.device_cv.shape

    def copy(self, deep=True):
        return DataArray(self.data, self.dims, dict(self.coords))


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.coords = {k: np.asarray(v) for k, v in (coords or {}).items()}
        self.attrs = dict(attrs or {})
        self.data_vars = {}
        for name, v in (data_vars or {}).items():
            self[name] = v

    def __getitem__(self, name):
        return self.data_vars[name]

    def __setitem__(self, name, value):
        # (engine.DeviceMapArray, criteria.LazyValidity: asked of the TYPE - hasattr on the instance would run the ``data``
        #  property, i.e. download the map)
        if isinstance(value, DataArray) or (hasattr(type(value), "data") and hasattr(value, "dims")):
            self.data_vars[name] = value
        else:
            dims, data = value
            self.data_vars[name] = DataArray(np.asarray(data), dims, {d: self.coords[d] for d in dims if d in self.coords})

    def __contains__(self, name):
        return name in self.data_vars or name in self.coords

    @property
    def sizes(self):
        out = {k: len(v) for k, v in self.coords.items()}
        for v in self.data_vars.values():
            for d, n in zip(v.dims, v.shape):
                out.setdefault(d, n)
        return out

    def copy(self, deep=True):
        ds = Dataset(coords=dict(self.coords), attrs=dict(self.attrs))
        for k, v in self.data_vars.items():
            ds.data_vars[k] = v.copy(deep)
        return ds


def make_image(data, disparity=None, msk=None, valid_pixels=0, no_data_mask=1, disparity_grids=None, band_names=None, segm=None,
               edges=None, classif=None):
    """Image dataset as produced by img_tools.create_dataset_from_inputs (img_tools.py:345-437):
    ``im`` float32 (row, col) - or (band_im, row, col) with ``band_names`` as the band_im coordinate -, optional ``msk``
    int16, ``disparity`` (band_disp=[min,max], row, col)."""
    data = np.asarray(data)
    if data.ndim == 3:
        if band_names is None or len(band_names) != data.shape[0]:
            raise ValueError("a multiband image (band_im, row, col) needs one band name per band")
        H, W = data.shape[1:]
        ds = Dataset({"im": (("band_im", "row", "col"), data.astype(np.float32))},
                     coords={"band_im": np.asarray(band_names), "row": np.arange(H), "col": np.arange(W)})
    elif data.ndim == 2:
        H, W = data.shape
        ds = Dataset({"im": (("row", "col"), data.astype(np.float32))}, coords={"row": np.arange(H), "col": np.arange(W)})
    else:
        raise ValueError("an image is (row, col) or (band_im, row, col)")
    ds.attrs.update({"valid_pixels": valid_pixels, "no_data_mask": no_data_mask, "crs": None, "transform": None, "no_data_img": None})
    if msk is not None:
        ds["msk"] = (("row", "col"), np.asarray(msk, np.int16))
    if segm is not None:  # img_tools.py:190-231: layers the SGM step's geometric_prior reads
        ds["segm"] = (("row", "col"), np.asarray(segm, np.int16))
    if edges is not None:
        ds["edges"] = (("row", "col"), np.asarray(edges, np.int16))
    if classif is not None:  # (bands, names): img_tools.py:165-187
        bands, names = classif
        ds.coords["band_classif"] = np.asarray(list(names), dtype=object)
        ds["classif"] = DataArray(np.asarray(bands, np.int16), ("band_classif", "row", "col"))
    ds.attrs["disparity_source"] = None
    if disparity is not None:
        dmin, dmax = disparity
        ds.attrs["disparity_source"] = [dmin, dmax]  # img_tools.py:406-437: the [min, max] list or the grid's file name
        grids = np.empty((2, H, W), np.int64)
        grids[0], grids[1] = dmin, dmax
        ds.coords["band_disp"] = np.array(["min", "max"])
        ds["disparity"] = DataArray(grids, ("band_disp", "row", "col"), {"band_disp": ["min", "max"]})
    if disparity_grids is not None:
        gmin, gmax = disparity_grids
        ds.attrs["disparity_source"] = "disparity_grids"
        ds.coords["band_disp"] = np.array(["min", "max"])
        ds["disparity"] = DataArray(np.stack([np.asarray(gmin), np.asarray(gmax)]), ("band_disp", "row", "col"),
                                    {"band_disp": ["min", "max"]})
    return ds


def from_xarray(xds):
    """Adapt a real xarray.Dataset (when xarray is installed) to the shim."""
    ds = Dataset(coords={k: np.asarray(v.values) for k, v in xds.coords.items()}, attrs=dict(xds.attrs))
    for name, var in xds.data_vars.items():
        ds.data_vars[name] = DataArray(np.asarray(var.values), var.dims,
                                       {d: list(np.asarray(xds.coords[d].values)) for d in var.dims if d in xds.coords})
    return ds
