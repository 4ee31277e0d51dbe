"""Shared by the oracle and GPU tests: cbca.py:184-295 (computes_cross_supports) composed from the oracle's pieces."""
import numpy as np


def oracle_cross_supports(oracle, L, R, msk_left, msk_right, subpix, offset, distance, intensity, valid=0):
    """-> (left arms, [right arms per sub-pixel phase]); masked pixels (msk != valid_pixels) become NaN before the 3x3
    nanmedian, a half-pixel sample is masked when either neighbour is (cbca.py:246-262)."""
    def arms(im, msk, shifted):
        m = np.array(im, np.float32, copy=True)
        if msk is not None:
            bad = np.asarray(msk) != valid
            if shifted:
                bad = bad[:, :-1] | bad[:, 1:]
            m[bad] = np.nan
        m = np.nan_to_num(oracle.median3(m), nan=np.inf)
        if offset > 0:
            m = m[offset:-offset, offset:-offset]
        return oracle.cross_support(np.ascontiguousarray(m), distance, intensity)

    return arms(L, msk_left, False), [arms(im, msk_right, k > 0) for k, im in enumerate(oracle.shift_right(R, subpix))]


def oracle_sad_cbca(oracle, L, R, msk_left, msk_right, win, subpix, dmin, dmax, distance, intensity):
    """SAD -> cv_masked -> CBCA, the pipeline of tests/test_aggregation.py."""
    L, R = np.asarray(L, np.float32), np.asarray(R, np.float32)
    D = (dmax - dmin) * subpix + 1
    cv = oracle.sad_ssd(L, R, D, dmin, subpix, win, False)
    oracle.cv_masked(cv, dmin, subpix, win, msk_left, msk_right, 0, 1)
    cl, crs = oracle_cross_supports(oracle, L, R, msk_left, msk_right, subpix, win // 2, distance, intensity)
    oracle.cbca(cv, dmin, subpix, win // 2, cl, crs)
    return cv
