"""float32 SGM: the three schedules of pmx_sgm (one launch per path "seq", eight paths side by side "par", the horizontal pair +
two fused three-path marching passes "fam", csrc/k_sgmfam.hip) against the CPU oracle, bit for bit, through the C ABI.
The family schedule hands path costs from one workgroup window to the next inside a launch: shapes are chosen so that windows
start inside / outside the image, finish early, are a single column wide at the image border, and so that both lane maps
(16 and 32 lanes per pixel) and every instantiated disparities-per-lane count run; costs are non-integers so that the float32
summation order of the definition is visible."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pandora_amd.engine import Engine

    e = Engine(0)
    e.set_lazy(False)
    yield e
    e.close()


def volume(rng, H, W, D, is_max=False, nan_frac=0.05):
    cv = (rng.random((H, W, D)).astype(np.float32) * 3 - 1) if is_max else (rng.random((H, W, D)) * 40).astype(np.float32)
    if nan_frac:
        cv[rng.random(cv.shape) < nan_frac] = np.nan
        cv[H // 2, W // 3] = np.nan
    return cv


def run(eng, cvh, P1, P2, is_max, inv, over, mask=0xFF):
    H, W, D = cvh.shape
    z = np.zeros((H, W), np.float32)
    eng.set_images(z, z, 1)
    cv = eng.alloc_cv(D, 0)
    cv.from_host(cvh)
    eng.sgm(cv, P1, P2, is_max, inv, over, dir_mask=mask)
    out = cv.to_host()
    cv.free()
    return out


# (H, W, D, lane map "lanes per pixel, disparities per lane, compute waves per workgroup" forced through PMX_SGM_FAM_SHAPE; None =
# the library's own choice, which for images this narrow is always a 32-lane map with 4 waves)
SHAPES = [
    (40, 70, 30, "16,3,4"),
    (33, 100, 61, "16,5,8"),    # cones' D
    (70, 41, 100, "16,7,4"),    # taller than wide: most windows start below the first row
    (25, 130, 129, "16,9,4"),   # C3 / C5 D: the last lane owns 3 disparities
    (25, 130, 129, "16,9,8"),
    (25, 130, 129, "16,9,10"),  # C5's map: 40-column windows (250 of them over 10000 columns: one per CU)
    (31, 95, 61, "16,5,10"),
    (37, 50, 257, "32,9,10"),
    (12, 45, 300, "32,12,10"),
    (9, 33, 512, "32,16,10"),   # (the launcher only picks it when the exchange buffers fit: they do not, 181 KB; the hook must not crash)
    (25, 130, 129, None),       # C3's map: 32 x 5
    (30, 64, 144, "16,9,4"),    # no tail
    (28, 60, 90, "32,3,8"),
    (20, 90, 150, "32,6,4"),
    (37, 50, 257, None),        # C4 D: 32 x 9
    (37, 50, 257, "32,9,8"),    # C4's map
    (12, 45, 300, "32,12,8"),
    (9, 33, 512, None),         # 32 x 16, the largest D
    (9, 33, 512, "32,16,8"),
    (2, 17, 20, None),          # two rows
    (50, 3, 40, "16,3,8"),      # narrower than any window
]


@pytest.mark.parametrize("beside", ["0", "1"])  # SGM_FAM_PAR: the downward family in line (default) / beside the horizontal pair
@pytest.mark.parametrize("H,W,D,lanes", SHAPES)
def test_family_schedule_equals_oracle(eng, oracle, hooks, H, W, D, lanes, beside):
    rng = np.random.default_rng(H * 1000 + W)
    cvh = volume(rng, H, W, D)
    exp = oracle.sgm(cvh, 1.5, 7.25, False, 45.0, False)
    hooks.setenv("PMX_SGM_SCHED", "fam")
    hooks.setenv("PMX_SGM_FAM_PAR", beside)
    if lanes:
        hooks.setenv("PMX_SGM_FAM_SHAPE", lanes)
    np.testing.assert_array_equal(run(eng, cvh, 1.5, 7.25, False, 45.0, False), exp)


@pytest.mark.parametrize("is_max,over", [(False, True), (True, False), (True, True)])
def test_family_schedule_max_measures_and_overcounting(eng, oracle, hooks, is_max, over):
    rng = np.random.default_rng(5)
    cvh = volume(rng, 31, 77, 129, is_max)
    exp = oracle.sgm(cvh, 0.3, 1.7, is_max, 45.0, over)
    for sched, beside in (("seq", "0"), ("par", "0"), ("fam", "0"), ("fam", "1")):
        hooks.setenv("PMX_SGM_SCHED", sched)
        hooks.setenv("PMX_SGM_FAM_PAR", beside)
        np.testing.assert_array_equal(run(eng, cvh, 0.3, 1.7, is_max, 45.0, over), exp)


@pytest.mark.parametrize("mask", [0x01, 0x02, 0x03, 0x04, 0x08, 0x10, 0x1C, 0x20, 0x40, 0x80, 0xE0, 0xFC, 0x5A, 0xA5, 0x1F])
def test_direction_masks_every_schedule(eng, oracle, hooks, mask):
    """pmx_debug_sgm_directions: any subset of the eight paths, same bits from every schedule (the first path of a subset starts the
    sum, the last one applies the epilogue, whatever kernel it runs in)."""
    rng = np.random.default_rng(mask)
    cvh = volume(rng, 27, 53, 129)
    exp = oracle.sgm(cvh, 2.5, 9.0, False, 45.0, False, dir_mask=mask)
    for sched in ("seq", "par", "fam"):
        hooks.setenv("PMX_SGM_SCHED", sched)
        np.testing.assert_array_equal(run(eng, cvh, 2.5, 9.0, False, 45.0, False, mask), exp)


def test_family_schedule_many_windows(eng, oracle, hooks):
    """Wide enough for 8 compute waves per workgroup (32-column windows) and hundreds of windows in flight: the hand-off chain is
    as long as at full size.  Integer costs keep the oracle fast; the result must still be exact."""
    rng = np.random.default_rng(77)
    H, W, D = 96, 7200, 33
    cvh = rng.integers(0, 30, (H, W, D)).astype(np.float32)
    exp = oracle.sgm(cvh, 8.0, 32.0, False, 45.0, False)
    hooks.setenv("PMX_SGM_SCHED", "fam")
    np.testing.assert_array_equal(run(eng, cvh, 8.0, 32.0, False, 45.0, False), exp)  # the library's choice: 16 x 3, 8 waves
    hooks.setenv("PMX_SGM_FAM_SHAPE", "32,3,4")                                  # 8-column windows: 900 of them
    np.testing.assert_array_equal(run(eng, cvh, 8.0, 32.0, False, 45.0, False), exp)


def test_family_schedule_repeated_launches(eng, oracle, hooks):
    """The hand-off buffer is recycled from launch to launch (tags = launch epochs): a second and third volume through the same
    context must not see the first one's granules."""
    hooks.setenv("PMX_SGM_SCHED", "fam")
    for seed in (1, 2, 3):
        rng = np.random.default_rng(seed)
        cvh = volume(rng, 45, 200, 129)
        exp = oracle.sgm(cvh, 3.0, 11.0, False, 45.0, False)
        np.testing.assert_array_equal(run(eng, cvh, 3.0, 11.0, False, 45.0, False), exp)


@pytest.mark.parametrize("H,W,D", [(20, 64, 129), (9, 37, 257), (5, 8, 60), (3, 7, 30), (6, 131, 384)])
def test_fused_horizontal_pair_equals_the_two_line_passes(eng, oracle, hooks, H, W, D):
    """The family schedule runs (0,+1) and (0,-1) as a checkpoint pass + a backward pass that re-computes the forward path segment by
    segment (k_sgm.hip): same bits as the oracle and as the two separate line passes (PMX_SGM_HFUSED=0); widths that are / are not
    multiples of the 8-column segment, narrower than one segment, max / overcounting epilogue in the backward kernel."""
    rng = np.random.default_rng(W)
    hooks.setenv("PMX_SGM_SCHED", "fam")
    for is_max, over in ((False, False), (True, True)):
        cvh = volume(rng, H, W, D, is_max)
        for mask in (0x03, 0xFF):
            exp = oracle.sgm(cvh, 1.25, 6.5, is_max, 45.0, over, dir_mask=mask)
            for fused in ("1", "0"):
                hooks.setenv("PMX_SGM_HFUSED", fused)
                np.testing.assert_array_equal(run(eng, cvh, 1.25, 6.5, is_max, 45.0, over, mask), exp)


@pytest.mark.parametrize("beside", ["0", "1"])
@pytest.mark.parametrize("is_max,over,D", [(False, False, 129), (True, False, 257), (False, True, 61), (True, True, 40)])
def test_deferred_last_pass_with_fused_wta(oracle, monkeypatch, is_max, over, D, beside):
    """Lazy mode + family schedule: pmx_sgm leaves the upward family pending; pmx_wta runs it in WTA mode (the optimised volume is
    never written) and pmx_refine works from the winner's three values; reading the volume instead runs the pass in store mode.
    Every route gives the oracle's bits: disparity, validity (all-NaN pixels included), interpolated coefficient, volume."""
    from pandora_amd.engine import Engine

    monkeypatch.setenv("PMX_SGM_SCHED", "fam")
    monkeypatch.setenv("PMX_SGM_FAM_PAR", beside)  # (two partial-sum volumes pending instead of one)
    rng = np.random.default_rng(D)
    H, W, dmin = 21, 150, -7
    cvh = volume(rng, H, W, D, is_max, nan_frac=0.1)
    cvh[5, 7] = np.nan      # a pixel without any cost
    cvh[9, 100:103] = np.nan
    inv = 45.0
    exp = oracle.sgm(cvh, 1.5, 7.25, is_max, inv, over)
    odisp, oval = oracle.wta(exp, dmin, 1, is_max, -9999.0)
    e = Engine(0)
    try:
        e.set_lazy(True)
        z = np.zeros((H, W), np.float32)
        e.set_images(z, z, 1)
        for method in ("vfit", "quadratic"):
            oitp, ordisp, orval = oracle.refine(exp, odisp.copy(), oval.copy(), dmin, dmin + D - 1, 1, is_max, method)
            # route 1: sgm -> wta (fused) -> refine (from the winner cache)
            cv = e.alloc_cv(D, dmin)
            cv.from_host(cvh)
            e.sgm(cv, 1.5, 7.25, is_max, inv, over)
            e.set_validity(None)
            e.wta(cv, is_max, -9999.0)
            d, v = e.get_disparity()
            np.testing.assert_array_equal(d, odisp)
            np.testing.assert_array_equal(v, oval)
            e.refine(cv, method, is_max)
            d, v, t = e.get_disparity(want_itp=True)
            np.testing.assert_array_equal(d, ordisp)
            np.testing.assert_array_equal(v, orval)
            np.testing.assert_array_equal(t, oitp)
            # ... and the volume is still there for whoever asks (store mode runs now)
            np.testing.assert_array_equal(cv.to_host(), exp)
            # route 2: an edited map between WTA and refinement: the cache is not trusted, the volume is materialised
            cv.from_host(cvh)
            e.sgm(cv, 1.5, 7.25, is_max, inv, over)
            e.set_validity(None)
            e.wta(cv, is_max, -9999.0)
            d, v = e.get_disparity()
            e.set_disparity(d, v)
            e.refine(cv, method, is_max)
            d, v, t = e.get_disparity(want_itp=True)
            np.testing.assert_array_equal(d, ordisp)
            np.testing.assert_array_equal(t, oitp)
            # route 3: the handle is reused for another volume while a pass is pending
            cv.from_host(cvh)
            e.sgm(cv, 1.5, 7.25, is_max, inv, over)
            cv.from_host(cvh[::-1].copy())
            np.testing.assert_array_equal(cv.to_host(), cvh[::-1])
            cv.free()
    finally:
        e.close()


def test_every_window_of_a_marching_launch_is_taken(eng, oracle, hooks):
    """pmx_debug_fam_windows after a family-schedule run with hundreds of windows: every window carries the XCD it ran on (1..8),
    the chunk counters reached their chunks' sizes - and the result is the oracle's (tickets: csrc/pmx_buf.h pmx_take_window)."""
    import ctypes

    from pandora_amd import _lib

    rng = np.random.default_rng(11)
    H, W, D = 64, 5000, 33
    cvh = rng.integers(0, 30, (H, W, D)).astype(np.float32)
    exp = oracle.sgm(cvh, 8.0, 32.0, False, 45.0, False)
    hooks.setenv("PMX_SGM_SCHED", "fam")
    for g in (None, "4", "1"):  # the library's chunk size (an XCD's CUs), small chunks, every neighbour on another XCD
        if g:
            hooks.setenv("PMX_SGM_FAM_XCD", g)
        np.testing.assert_array_equal(run(eng, cvh, 8.0, 32.0, False, 45.0, False), exp)
        buf = (ctypes.c_uint * 4096)()
        n = _lib.lib().pmx_debug_fam_windows(eng.ctx, buf, 4096)
        assert n > 8
        words = np.frombuffer(buf, np.uint32)[:n]
        # 16 lanes x 3 disparities, 8 or 10 compute wavefronts: windows of 32 / 40 columns
        for cw in (32, 40):
            nwin = (W + H - 2) // cw + 1
            flags = words[8:8 + nwin]
            if (flags >= 1).all() and (flags <= 8).all() and (words[8 + nwin:8 + nwin + 8] == 0).all():
                break
        else:
            raise AssertionError(f"window table: not every window started exactly once: {words[:64]}")
        G = int(g) if g else None
        if G:
            assert (words[:min(8, -(-nwin // G))] >= min(G, nwin)).all() or nwin < G
