"""Image pyramids for the multiscale run (reference: img_tools.py:479-712 prepare_pyramid / get_pyramids / masks_pyramid
/ convert_pyramid_to_dataset, check_configuration.py:558-582 read_multiscale_params).  Host-side 2-D preparation.

PARITY UNPINNED for the pyramid itself: the reference calls skimage.transform.pyramid_gaussian(sigma=1.2, order=1,
mode="reflect"), and scikit-image is not installed here.  `_pyramid_reduce` restates what that function does with the
scipy.ndimage calls scikit-image itself makes (gaussian_filter in 'reflect' mode, then resize = ndimage.zoom with
grid_mode=True and the 'mirror' boundary that skimage maps 'reflect' to, anti-aliasing off).

What the documented definition of pyramid_gaussian fixes for ANY faithful implementation is tested in
tests/test_pyramid_properties.py: layer shapes ceil(n / downscale), layers stop when they no longer shrink, constants are kept,
a linear ramp is sampled at the pixel-centre grid x_j = (j + 0.5) * downscale - 0.5, linearity, mirror symmetry, the 1.2-pixel
gaussian, and on even sizes the 2 x 2 block mean of the smoothed image.  A vector from scikit-image itself would pin the rest
(border handling of the resize); until then the row stays "parity unpinned" in DESIGN.md."""
import math

import numpy as np
from scipy import ndimage as ndi

from ..dataset import DataArray, Dataset


def _pyramid_reduce(image, downscale, sigma=1.2, order=1, cval=0):
    out_shape = tuple(math.ceil(d / float(downscale)) for d in image.shape)
    smoothed = ndi.gaussian_filter(image, sigma, mode="reflect", cval=cval)
    factors = [o / i for i, o in zip(image.shape, out_shape)]
    return ndi.zoom(smoothed, factors, order=order, mode="mirror", cval=cval, grid_mode=True)


def get_pyramids(data, num_scales, scale_factor):
    """img_tools.py:479-496: [full resolution, ..., coarsest]"""
    layers = [np.asarray(data, np.float32)]
    for _ in range(num_scales - 1):
        prev = layers[-1]
        nxt = _pyramid_reduce(prev, scale_factor)
        if nxt.shape == prev.shape:  # skimage stops when a layer no longer shrinks
            break
        layers.append(nxt)
    return layers


def masks_pyramid(msk, scale_factor, num_scales):
    """img_tools.py:617-635: decimation."""
    out, tmp = [msk], msk
    for _ in range(num_scales - 1):
        tmp = tmp[::scale_factor, ::scale_factor]
        out.append(tmp)
    return out


def _convert(img_orig, images, masks, disps):
    """img_tools.py:638-712 (mono-band)"""
    pyramid = []
    for index, image in enumerate(images):
        if index == 0:
            pyramid.append(img_orig)
            continue
        ds = Dataset({"im": (("row", "col"), image.astype(np.float32))},
                     coords={"row": np.arange(image.shape[0]), "col": np.arange(image.shape[1])})
        ds["msk"] = (("row", "col"), np.full(image.shape, masks[index]).astype(np.int16))
        if disps is not None:
            ds.coords["band_disp"] = np.array(["min", "max"])
            ds["disparity"] = DataArray(np.array([disps[0][index].astype(np.int64), disps[1][index].astype(np.int64)]),
                                        ("band_disp", "row", "col"), {"band_disp": ["min", "max"]})
        ds.attrs = img_orig.attrs  # shared, as in the reference
        pyramid.append(ds)
    return pyramid


def fill_nodata_image(ds):
    """img_tools.py:578-613 (mono-band): masked images get their no-data / invalid pixels interpolated on the device
    (pmx_interpolate_nodata replaces img_tools_cpp.interpolate_nodata_sgm) so that the Gaussian reduction does not smear
    them; images without a mask get an all-valid one."""
    if "msk" in ds.data_vars:
        from .. import runtime
        from ..constants import PANDORA_MSK_PIXEL_FILLED_NODATA, PANDORA_MSK_PIXEL_INVALID

        if np.asarray(ds["im"].data).ndim != 2:
            raise NotImplementedError("multiscale on multiband images is outside pandora_amd's scope")
        return runtime.get_engine().interpolate_nodata(ds["im"].data, ds["msk"].data, PANDORA_MSK_PIXEL_INVALID,
                                                       PANDORA_MSK_PIXEL_FILLED_NODATA)
    return ds["im"].data, np.full((ds.sizes["row"], ds.sizes["col"]), int(ds.attrs.get("valid_pixels", 0)))


def prepare_pyramid(img_left, img_right, num_scales, scale_factor):
    """img_tools.py:499-572.  Returns the two pyramids, coarsest first; the full-resolution level is the original
    dataset (image and mask untouched), the coarser ones come from the filled image and the decimated filled mask."""
    out = []
    for ds in (img_left, img_right):
        img, msk = fill_nodata_image(ds)
        images = get_pyramids(img, num_scales, scale_factor)
        disps = None
        if "disparity" in ds.data_vars:
            d = np.asarray(ds["disparity"].data)
            disps = [get_pyramids(d[0].astype(np.float32), num_scales, scale_factor),
                     get_pyramids(d[1].astype(np.float32), num_scales, scale_factor)]
        out.append(_convert(ds, images, masks_pyramid(msk, scale_factor, num_scales), disps)[::-1])
    return out[0], out[1]


def read_multiscale_params(left_img, right_img, cfg):
    """check_configuration.py:558-582 -> (num_scales, scale_factor)"""
    from .multiscale import AbstractMultiscale

    if "multiscale" in cfg["pipeline"]:
        m = AbstractMultiscale(left_img, right_img, **cfg["pipeline"]["multiscale"])
        return m.cfg["num_scales"], m.cfg["scale_factor"]
    return 1, 1
