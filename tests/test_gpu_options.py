"""The context's kernel-route options (include/pandora_amd.h pmx_set_option): set / get / clear, unknown names refused, the
environment read once - when the context is created - and never again, and a forced route really taken."""
import numpy as np
import pytest

from tests.test_gpu_parity import pair

pytestmark = pytest.mark.gpu


def test_options_set_get_clear_and_unknown_names(monkeypatch):
    from pandora_amd.engine import Engine, PmxError

    monkeypatch.setenv("PMX_SGM_SCHED", "seq")  # seeds the option of contexts created from now on
    e = Engine(0)
    try:
        assert "SGM8_FAM" in Engine.option_names() and "CBCA_FAST" in Engine.option_names()
        assert e.get_option("SGM_SCHED") == "seq" and e.get_option("SGM8_FAM") is None
        monkeypatch.setenv("PMX_SGM8_FAM", "1")  # after pmx_create: the library does not look
        assert e.get_option("SGM8_FAM") is None
        e.set_option("PMX_SGM8_FAM", "1")        # (with or without the prefix)
        assert e.get_option("SGM8_FAM") == "1"
        e.set_option("SGM8_FAM", None)
        assert e.get_option("SGM8_FAM") is None
        e.options_from_env()                     # what a script does that changes its environment between steps
        assert e.get_option("SGM8_FAM") == "1"
        with pytest.raises(PmxError):
            e.set_option("NO_SUCH_OPTION", "1")
    finally:
        e.close()


def test_a_forced_route_is_the_one_that_runs(oracle):
    """SGM8_FAM=1 on a small pair: three byte volumes instead of eight (pmx_debug_path_costs has nothing to show), same bits."""
    from pandora_amd.engine import Engine

    L, R = pair(40, 100, seed=11)
    e = Engine(0)
    try:
        e.set_images(L, R, 1)
        ref = oracle.sgm(oracle.census_cost(L, R, 17, -8, 1, 5), 8, 32, False, 26.0, False)
        for forced in (None, "1"):
            e.set_option("SGM8_FAM", forced)
            cv = e.alloc_cv(17, -8)
            e.census(cv, 5)
            e.sgm(cv, 8, 32, False, 26.0, False)
            if forced:
                with pytest.raises(Exception):
                    e.debug_path_costs(cv, raw=True)
            else:
                assert e.debug_path_costs(cv, raw=True)[0].shape[0] == 8
            np.testing.assert_array_equal(cv.to_host(), ref)
            cv.free()
    finally:
        e.close()


def test_release_caches_returns_the_memory_and_changes_nothing():
    """pmx_release_caches: after a float32 SGM run the context holds an accumulator volume, freed volumes and the marching kernels'
    hand-off buffer; releasing them gives the bytes back to the driver, and the next run (which allocates them again, hand-off
    epochs starting over) gives the same result bit for bit."""
    import numpy as np

    from pandora_amd.engine import Engine

    eng = Engine(0)
    eng.set_lazy(False)
    try:
        rng = np.random.default_rng(5)
        H, W, D = 600, 3600, 40  # wide enough for the family schedule (hand-off buffer)
        cvh = (rng.random((H, W, D)) * 40).astype(np.float32)
        z = np.zeros((H, W), np.float32)

        def run():
            eng.set_images(z, z, 1)
            cv = eng.alloc_cv(D, 0)
            cv.from_host(cvh)
            eng.set_option("SGM_SCHED", "fam")
            eng.sgm(cv, 1.5, 7.25, False, 45.0, False)
            eng.set_option("SGM_SCHED", None)
            out = cv.to_host()
            cv.free()
            return out

        first = run()
        eng.sync()
        held, total = eng.release_caches()
        again, _ = eng.release_caches()  # (idempotent)
        assert again >= held - (64 << 20)
        second = run()
        before, _ = eng.release_caches()
        np.testing.assert_array_equal(first, second)
        # the volume (346 MB), its accumulator and the hand-off buffer were held between the calls: releasing must have freed at least
        # the two volumes' worth
        import ctypes as C

        f, t = C.c_size_t(0), C.c_size_t(0)
        third = run()
        np.testing.assert_array_equal(first, third)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemGetInfo(C.byref(f), C.byref(t))
        busy = f.value
        freed, _ = eng.release_caches()
        assert freed - busy >= 2 * cvh.nbytes, (freed, busy, cvh.nbytes)
        assert total > 0 and abs(freed - before) < (256 << 20)
    finally:
        eng.close()


def test_out_of_memory_is_an_error_once_and_the_context_stays_usable(oracle):
    """A pair whose buffers do not fit: pmx_set_images / pmx_census / pmx_sgm answer with an error - and that is all.  The context holds
    no half-allocated pair (the same shape again, with memory back, works), the runtime's sticky "last error" does not fail the next
    call's launch check, and the next results are the oracle's.  (Round 6: found when four processes shared one GPU.)"""
    import ctypes

    from pandora_amd._lib import PmxError
    from pandora_amd.engine import Engine

    hip = ctypes.CDLL("libamdhip64.so")
    free_b, total_b = ctypes.c_size_t(), ctypes.c_size_t()
    eng = Engine(0)
    hog = ctypes.c_void_p()
    try:
        eng.set_lazy(False)
        assert hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) == 0
        # leave ~1.2 GB: enough for the 2048 x 2048 images and maps (0.6 GB), not for a 129-disparity float32 volume (2.2 GB)
        assert hip.hipMalloc(ctypes.byref(hog), ctypes.c_size_t(free_b.value - (1200 << 20))) == 0
        H, W, dmin, dmax = 2048, 2048, 0, 128
        L, R = pair(H, W, seed=3)
        failed = 0
        for _ in range(2):  # the same shape twice: the second call must not find "same shape, buffers kept" on missing buffers
            try:
                eng.set_images(L, R, 1)
                cv = eng.alloc_cv(dmax - dmin + 1, dmin)
                eng.census(cv, 5)
                eng.sgm(cv, 8.0, 32.0, False, 26.0, False)
                eng.sync()
                cv.free()
            except PmxError as err:
                failed += 1
                assert "memory" in str(err).lower(), str(err)
        assert failed == 2
    finally:
        if hog.value:
            hip.hipFree(hog)
    try:
        # memory is back: a small pair through the same context, against the oracle
        Ls, Rs = pair(40, 64, seed=5)
        eng.set_images(Ls, Rs, 1)
        cv = eng.alloc_cv(17, -8)
        eng.census(cv, 5)
        eng.sgm(cv, 8.0, 32.0, False, 26.0, False)
        got = cv.to_host()
        cv.free()
        exp = oracle.sgm(oracle.census_cost(Ls, Rs, 17, -8, 1, 5), 8.0, 32.0, False, 26.0, False)
        np.testing.assert_array_equal(got, exp)
        # ... and the big shape now fits
        eng.set_images(L, R, 1)
        cv = eng.alloc_cv(dmax - dmin + 1, dmin)
        eng.census(cv, 5)
        eng.sync()
        cv.free()
    finally:
        eng.close()


def test_maps_of_the_wrong_shape_are_refused_before_the_library_reads_them():
    """The C ABI takes a pointer and reads H x W elements behind it; Engine checks masks, grids and maps against the resident pair (a
    mask shorter than the image is a read past its end - a GPU memory fault when the runtime pins the caller's pages)."""
    from pandora_amd.engine import Engine

    eng = Engine(0)
    try:
        L, R = pair(30, 50, seed=1)
        eng.set_images(L, R, 1)
        small = np.zeros((20, 50), np.int16)
        with pytest.raises(ValueError, match="shape"):
            eng.set_masks(small, None, 0, 1)
        with pytest.raises(ValueError, match="shape"):
            eng.set_masks(None, np.zeros((30, 49), np.int16), 0, 1)
        with pytest.raises(ValueError, match="shape"):
            eng.set_disparity_grids(np.zeros((29, 50)), np.zeros((30, 50)))
        with pytest.raises(ValueError, match="shape"):
            eng.set_disparity(np.zeros((30, 51), np.float32), np.zeros((30, 50), np.int64))
        with pytest.raises(ValueError, match="shape"):
            eng.set_validity(np.zeros((30, 25), np.int64))
        with pytest.raises(ValueError, match="validity mask"):
            eng.cross_checking(np.zeros((30, 50), np.float32), np.zeros((15, 50), np.int64), np.zeros((30, 50), np.float32), -3, 3, 1.0)
        with pytest.raises(ValueError, match="set_full_rows"):
            eng.set_full_rows(60, 0, 30, np.zeros((30, 50), np.float32), np.zeros((29, 50), np.int64))
        eng.set_full_rows(60, 0, 30, np.zeros((30, 50), np.float32), np.zeros((30, 50), np.int64))
        with pytest.raises(ValueError, match="get_full_maps"):
            eng.get_full_maps(90)
        assert eng.get_full_maps(60)[0].shape == (60, 50)
        with pytest.raises(ValueError, match="xbuf_upload"):
            eng.xbuf_upload("keys", np.zeros(10, np.uint64))
        eng.set_masks(np.zeros((30, 50), np.int16), None, 0, 1)  # the right shape goes through
        eng.set_disparity_grids(np.full((30, 50), -3.0), np.full((30, 50), 3.0))
    finally:
        eng.close()
