"""GPU parity of the integer path's direction-family form (k_sgmfam8.hip + sgm_u8_hpair_kernel): three byte volumes (horizontal
pair, downward family, upward family) instead of eight path volumes.  Forced onto small pairs with PMX_SGM8_FAM=1 (by default it
takes images from 480 rows and 2048 columns on whose rows hold enough cells, W D >= 26500 KPL - 50000; shorter ones from 192 rows with 1.8 times the cells per row) and compared with the oracle's 8-path SGM bit for bit: the summed volume, the WTA and the
refinement.  Shapes exercise every lane map (KPL 4 ... 20), both window widths (16 / 32 columns), images narrower than a window,
images a window does not divide, and windows that enter and leave the image during the march."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pandora_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.fixture
def forced_families(hooks):
    hooks.setenv("PMX_SGM8_FAM", "1")
    yield hooks


def run_both(eng, oracle, L, R, dmin, dmax, win, P1, P2, family=True):
    D = dmax - dmin + 1
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, dmin)
    eng.census(cv, win)
    eng.sgm(cv, P1, P2, False, float(win * win + 1), False)
    eng.set_validity(None)
    eng.wta(cv, False, -9999.0)
    eng.refine(cv, "vfit", False)
    disp, val, itp = eng.get_disparity(want_itp=True)
    if family:
        with pytest.raises(Exception):
            eng.debug_path_costs(cv, raw=True)  # (there are no eight path volumes: the family form ran)
    vol = cv.to_host()
    cv.free()
    ref = oracle.sgm(oracle.census_cost(L, R, D, dmin, 1, win), P1, P2, False, float(win * win + 1), False)
    rdisp, rval = oracle.wta(ref, dmin, 1, False, -9999.0)
    ritp, rdisp2, rval2 = oracle.refine(ref, rdisp, rval, dmin, dmax, 1, False, "vfit")
    np.testing.assert_array_equal(vol, ref)
    np.testing.assert_array_equal(disp, rdisp2)
    np.testing.assert_array_equal(val, rval2)
    np.testing.assert_array_equal(itp, ritp)


@pytest.mark.parametrize("H,W,dmin,dmax,win,P1,P2", [
    (24, 37, -6, 3, 5, 8, 32),        # KPL 4, two windows of 16 columns + the ones that slide in
    (33, 70, -60, 0, 5, 8, 32),       # KPL 4 (61 disparities)
    (40, 100, 0, 64, 5, 8, 32),       # KPL 8
    (21, 90, 0, 128, 3, 4, 20),       # KPL 12, census 3x3
    (50, 45, -100, 100, 5, 8, 32),    # KPL 16 (201 disparities), more rows than columns
    (30, 130, 0, 256, 5, 8, 32),      # KPL 20 (257 disparities): the headline's lane map
    (9, 11, -2, 2, 3, 8, 32),         # narrower than one window
    (2, 64, 0, 10, 5, 8, 32),         # two rows
    (70, 16, -5, 5, 5, 1, 2),         # exactly one window wide, tall
    (45, 67, -20, 20, 7, 8, 30),      # census 7x7: byte costs (invalid cost 50: 3 * 80 = 240 fits a byte)
    (20, 600, 0, 256, 5, 8, 32),      # 257 disparities on a row wider than the range: the row walk's 64 x 4 + 1 form, blocks of
                                      # 16 columns with and without cells that are not numbers
    (14, 420, -256, 0, 5, 8, 32),     # ... to the other side
    (12, 340, -255, 0, 5, 8, 32),     # 256 disparities: the row walk's 64 lanes all busy
    (12, 340, 0, 257, 5, 8, 32),      # 258: eight per lane
    (8, 1400, -1150, -1100, 5, 8, 32),  # a range further from the pixel than the code images' guard (1024 words): the code-word
    (8, 1400, 1100, 1150, 5, 8, 32),    # kernels must not be chosen (their per-lane offsets would wrap / leave the allocation)
])
@pytest.mark.parametrize("nw,hpair,codes", [("4", "2", "0"), ("8", "1", "0"), ("8", "2", "0"),
                                            ("8", "1", "1"),   # both kernels from the census words, four rows per wavefront
                                            ("8", "3", "1"), ("4", "3", "1"),  # ... one row per wavefront (short images' default until round 6)
                                            ("8", "3", None), ("4", "3", None),  # row walk from the words, marching kernel from the cost volume (their default since)
                                            ("8", "3", "0")])  # the same, asked for
def test_family_form_equals_the_oracle(eng, oracle, forced_families, H, W, dmin, dmax, win, P1, P2, nw, hpair, codes):
    forced_families.setenv("PMX_SGM8_FAM_NW", nw)
    # the horizontal pair's one-sided (tall images) / two-sided walk on the cost volume, or the row-per-wavefront walk from the
    # census words (one-word windows; others fall back to the two-sided walk)
    forced_families.setenv("PMX_SGM8_HPAIR", hpair)
    if codes is not None:
        forced_families.setenv("PMX_SGM8_CODES", codes)
    L, R = pair(H, W, seed=3 * H + W)
    run_both(eng, oracle, L, R, dmin, dmax, win, P1, P2)


def test_family_form_is_not_taken_when_a_sum_would_not_fit_a_byte(eng, oracle, forced_families):
    """census 9x9: invalid cost 82, 3 * (82 + 32) > 255 -> the eight path volumes stay"""
    L, R = pair(30, 50, seed=4)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(9, -4)
    eng.census(cv, 9)
    eng.sgm(cv, 8, 32, False, 82.0, False)
    raw, _, _ = eng.debug_path_costs(cv, raw=True)
    assert raw.shape[0] == 8
    ref = oracle.sgm(oracle.census_cost(L, R, 9, -4, 1, 9), 8, 32, False, 82.0, False)
    np.testing.assert_array_equal(cv.to_host(), ref)
    cv.free()


def test_family_form_with_disparity_grids(eng, oracle, forced_families):
    """cv_masked on the integer path (per-pixel ranges): invalid cells carry invalid_cost in the byte costs, as on the 8-path route"""
    H, W, dmin, dmax, win = 40, 90, -12, 12, 5
    L, R = pair(H, W, seed=77)
    rng = np.random.default_rng(2)
    gmin = rng.integers(dmin, -2, (H, W)).astype(np.float64)
    gmax = rng.integers(2, dmax + 1, (H, W)).astype(np.float64)
    D = dmax - dmin + 1
    eng.set_images(L, R, 1)
    eng.set_disparity_grids(gmin, gmax)
    cv = eng.alloc_cv(D, dmin)
    eng.census(cv, win)
    eng.cv_masked(cv, win)
    eng.sgm(cv, 8, 32, False, float(win * win + 1), False)
    vol = cv.to_host()
    cv.free()
    eng.set_disparity_grids(None, None)
    cpu = oracle.census_cost(L, R, D, dmin, 1, win)
    oracle.cv_masked(cpu, dmin, 1, win, dmin=gmin, dmax=gmax)
    np.testing.assert_array_equal(vol, oracle.sgm(cpu, 8, 32, False, float(win * win + 1), False))


def test_family_form_mid_size_against_the_eight_volumes(eng):
    """Pairs large enough for a hundred windows in flight and for the default route (>= 480 rows, >= 2048 columns, W D >= 26500 KPL - 50000) - a short one
    (the two-sided horizontal walk) and a tall one (the one-sided walk): the family form and the eight-volume form give the same
    maps bit for bit."""
    from bench import synthetic_pair

    for H, W, dmin, dmax in ((500, 2600, 0, 64), (2600, 2600, -30, 10)):
        _family_against_eight_volumes(eng, synthetic_pair, H, W, dmin, dmax)


def test_short_wide_images_take_the_families_by_default(eng):
    """Round 6's rule for images below 480 rows (profiles/r06_fam_rows.txt): from 192 rows when a row holds 1.8 times the tall images'
    bound of cells.  300 rows x 4096 x 257 take the families by default (there are no eight path volumes afterwards) and give the
    eight-volume form's maps bit for bit; 300 x 2600 x 65 stay with the eight volumes."""
    from bench import synthetic_pair

    _family_against_eight_volumes(eng, synthetic_pair, 300, 4096, 0, 256)
    L, R = synthetic_pair(300, 2600, 0, 64, seed=5)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(65, 0)
    eng.census(cv, 5)
    eng.sgm(cv, 8, 32, False, 26.0, False)
    eng.debug_path_costs(cv, raw=True)  # (raises when the family form ran)
    cv.free()


def _family_against_eight_volumes(eng, synthetic_pair, H, W, dmin, dmax):
    L, R = synthetic_pair(H, W, dmin, dmax, seed=5)
    maps = {}
    for mode in ("0", "auto"):
        eng.set_option("SGM8_FAM", None if mode == "auto" else mode)
        try:
            eng.set_images(L, R, 1)
            cv = eng.alloc_cv(dmax - dmin + 1, dmin)
            eng.census(cv, 5)
            eng.sgm(cv, 8, 32, False, 26.0, False)
            eng.set_validity(None)
            eng.wta(cv, False, -9999.0)
            eng.refine(cv, "vfit", False)
            maps[mode] = eng.get_disparity(want_itp=True)
            if mode == "auto":
                with pytest.raises(Exception):
                    eng.debug_path_costs(cv, raw=True)
            cv.free()
        finally:
            eng.set_option("SGM8_FAM", None)
    for a, b in zip(maps["0"], maps["auto"]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("H,W,dmin,dmax", [
    (7, 33, -3, 1),        # one quad more than eight: the last quad of a row holds one pixel
    (5, 66, 0, 63),        # 64 disparities, a width that is no multiple of 4
    (9, 257, -128, 128),   # 257 disparities on a row of 257 columns: every block of a row touches the image's borders
    (3, 1001, 0, 40),      # 16 wavefronts' worth of quads per row and a ragged end
    (4, 64, 5, 9),         # a range that starts to the right of every pixel's own column
])
@pytest.mark.parametrize("wta3", [None, "0"])
def test_family_form_wta_kernels_agree_with_the_oracle(eng, oracle, forced_families, H, W, dmin, dmax, wta3):
    """The WTA of the three byte volumes: the kernel that stays in one image row (sum3_wta_kernel, default) and the general one
    (PMX_WTA3=0) on widths that are no multiple of a quad, of a wavefront's 16 quads, of a workgroup's 64"""
    if wta3 is not None:
        forced_families.setenv("PMX_WTA3", wta3)
    L, R = pair(H, W, seed=11 * H + W)
    run_both(eng, oracle, L, R, dmin, dmax, 5, 8, 32)


@pytest.mark.parametrize("H,W,dmin,dmax,win", [
    (20, 300, 0, 256, 13),        # six code words per pixel: (256 + 20) x 6 words lie behind a pixel of the last row
    (16, 1500, 1100, 1160, 11),   # four words, a range a thousand columns away
    (12, 40, 2000, 2010, 9),      # a range that never meets the image, further away than the image is large
    (12, 14, -3000, -2990, 5),    # ... to the other side: in front of the code images
])
@pytest.mark.parametrize("fam", ["0", "1"])
def test_code_guards_cover_the_whole_range(eng, oracle, hooks, H, W, dmin, dmax, win, fam):
    """The integer path's cost kernels read a pixel's right words at (pixel + d) x words-per-pixel through raw pointers for every d
    of the range, cells that are not numbers included: the guards around the code images are as long as the range is far (round 6;
    1024 dwords until then, whatever the window and the range).  The results never depended on it - this runs the shapes whose
    reads used to leave the allocation."""
    hooks.setenv("PMX_SGM8_FAM", fam)
    L, R = pair(H, W, seed=H + W)
    run_both(eng, oracle, L, R, dmin, dmax, win, 8, 32, family=False)
