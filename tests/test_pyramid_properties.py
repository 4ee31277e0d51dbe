"""CPU: what can be pinned of the image pyramid without scikit-image (N3; the reference calls
skimage.transform.pyramid_gaussian(downscale, sigma=1.2, order=1, mode="reflect", cval=0), img_tools.py:479-505, and holds no
pyramid vector).  pyramid_gaussian's documented definition - layer k+1 = resize(gaussian(layer k, sigma, mode), ceil(shape /
downscale), order=1, no anti-aliasing), stop when a layer no longer shrinks - fixes a set of properties that ANY faithful
implementation has; pandora_amd/multiscale/pyramid.py is tested against them.  The pyramid itself stays "parity unpinned"."""
import numpy as np
import pytest

from pandora_amd.multiscale import pyramid as pyr


@pytest.mark.parametrize("shape,scale", [((375, 450), 2), ((64, 64), 2), ((61, 97), 2), ((100, 90), 3), ((33, 20), 4)])
def test_shape_rule_and_constant_images(shape, scale):
    img = np.full(shape, 37.25, np.float32)
    layers = pyr.get_pyramids(img, 4, scale)
    assert layers[0].shape == shape
    for a, b in zip(layers, layers[1:]):
        assert b.shape == tuple(int(np.ceil(n / scale)) for n in a.shape)  # pyramid_reduce: ceil(d / downscale)
        np.testing.assert_allclose(b, 37.25, rtol=0, atol=1e-5)           # a normalised kernel and an interpolating resize keep constants
    assert len(layers) <= 4


def test_layers_stop_when_they_no_longer_shrink():
    layers = pyr.get_pyramids(np.ones((3, 2), np.float32), 8, 2)
    assert [x.shape for x in layers] == [(3, 2), (2, 1), (1, 1)]


@pytest.mark.parametrize("scale", [2, 4])
def test_linear_ramp_is_sampled_at_the_resize_grid(scale):
    """A gaussian leaves a linear ramp unchanged away from the border (symmetric kernel); order-1 resize without anti-aliasing of
    n -> n / scale samples samples the ramp at the pixel-centre grid x_j = (j + 0.5) * scale - 0.5."""
    n = 64 * scale
    ramp = np.tile(np.arange(n, dtype=np.float32), (n, 1))
    out = pyr._pyramid_reduce(ramp, scale)
    j = np.arange(out.shape[1])
    expect = (j + 0.5) * scale - 0.5
    inner = slice(4, -4)  # 4 sigma of the 1.2-pixel kernel is 5 input pixels
    np.testing.assert_allclose(out[10, inner], expect[inner], rtol=0, atol=1e-3)
    np.testing.assert_allclose(pyr._pyramid_reduce(ramp.T, scale)[inner, 10], expect[inner], rtol=0, atol=1e-3)


def test_mirror_symmetry_and_linearity():
    rng = np.random.default_rng(4)
    a = rng.random((48, 60)).astype(np.float32) * 255
    b = rng.random((48, 60)).astype(np.float32) * 255
    ra, rb = pyr._pyramid_reduce(a, 2), pyr._pyramid_reduce(b, 2)
    # both steps are linear ...
    np.testing.assert_allclose(pyr._pyramid_reduce(2 * a + 3 * b, 2), 2 * ra + 3 * rb, rtol=0, atol=1e-3)
    # ... and commute with flipping an even-sized image ("reflect" borders are symmetric)
    np.testing.assert_allclose(pyr._pyramid_reduce(a[::-1, ::-1], 2), ra[::-1, ::-1], rtol=0, atol=1e-4)
    # the smoothing is a separable gaussian of sigma 1.2: an impulse far from the border spreads with that variance
    imp = np.zeros((65, 65), np.float32)
    imp[32, 32] = 1.0
    from scipy import ndimage as ndi

    sm = ndi.gaussian_filter(imp, 1.2, mode="reflect")
    x = np.arange(65) - 32
    assert abs(float((sm.sum(0) * x ** 2).sum()) - 1.2 ** 2) < 0.02 and abs(float(sm.sum()) - 1.0) < 1e-6


def test_coarse_layer_of_cones_is_a_plausible_half_resolution_image():
    """The only data the reference's multiscale tests see (tests/test_pandora.py:328-394 run it on cones): the coarse layer stays
    within the image's range and close to the 2 x 2 block mean of the smoothed image."""
    import os

    from PIL import Image

    cones = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cones", "left.png")
    img = np.array(Image.open(cones)).astype(np.float32)
    layers = pyr.get_pyramids(img, 2, 2)
    assert layers[1].shape == (188, 225) and layers[1].dtype == np.float32
    assert layers[1].min() >= img.min() - 1e-3 and layers[1].max() <= img.max() + 1e-3
    from scipy import ndimage as ndi

    # on an even-sized crop the resize grid falls on the centres of the 2 x 2 blocks of the smoothed image
    crop = img[:374]
    sm = ndi.gaussian_filter(crop, 1.2, mode="reflect")
    block = (sm[0::2, 0::2] + sm[1::2, 0::2] + sm[0::2, 1::2] + sm[1::2, 1::2]) / 4
    np.testing.assert_allclose(pyr.get_pyramids(crop, 2, 2)[1], block, rtol=0, atol=1e-3)
