"""ctypes bindings to liboracle.so (oracle.c).  TEST INFRASTRUCTURE ONLY - see oracle/__init__.py."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # idle OpenMP threads sleep instead of spinning
        _LIB = C.CDLL(path)
        _LIB.orc_set_threads(1)  # serial, like the reference, unless a caller asks for more (set_threads)
    return _LIB


def usable_cores(cap=64):
    """Cores this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith("cpu.max"):
                if fields[0] != "max":
                    n = min(n, max(1, int(int(fields[0]) / int(fields[1]))))
            elif int(fields[0]) > 0:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    n = min(n, max(1, int(int(fields[0]) / int(f.read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, cap))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def shift_right(R, subpix):
    """List of the subpix right images (img_tools.py:713-752); image k>0 is one column narrower."""
    R = _f32(R)
    H, W = R.shape
    out = [R]
    for k in range(1, subpix):
        o = np.empty((H, W - 1), np.float32)
        lib().orc_shift_right(_p(R), H, W, subpix, k, _p(o))
        out.append(o)
    return out


def pack_shifted(imgs):
    """Back-to-back buffer of the shifted right images, the layout orc_* functions expect."""
    return np.concatenate([_f32(i).ravel() for i in imgs])


def census_transform(img, win):
    img = _f32(img)
    H, W = img.shape
    nb = lib().orc_census_nb_chars(win)
    out = np.empty((H, W, nb), np.uint8)
    lib().orc_census_transform(_p(img), H, W, win, _p(out, C.c_uint8))
    return out


def _cv_call(fn, L, R, D, d0, subpix, win, *extra, fill_nan=True, shifted=None):
    """shifted: the subpix-1 shifted right images to use instead of the linear ones (spline_order > 1)"""
    L = _f32(L)
    H, W = L.shape
    Rs = pack_shifted(shift_right(R, subpix) if shifted is None else [R] + list(shifted))
    cv = np.full((H, W, D), np.nan, np.float32) if fill_nan else np.empty((H, W, D), np.float32)
    fn(_p(L), _p(Rs), H, W, D, int(d0), subpix, win, *extra, _p(cv))
    return cv


def census_cost(L, R, D, d0, subpix, win, shifted=None):
    return _cv_call(lib().orc_census_cost, L, R, D, d0, subpix, win, shifted=shifted)


def sad_ssd(L, R, D, d0, subpix, win, squared, shifted=None):
    return _cv_call(lib().orc_sad_ssd, L, R, D, d0, subpix, win, int(squared), shifted=shifted)


def zncc(L, R, D, d0, subpix, win, shifted=None):
    return _cv_call(lib().orc_zncc, L, R, D, d0, subpix, win, shifted=shifted)


def mask_dilatation(msk, win, valid=0, nodata=1):
    msk = np.ascontiguousarray(msk, np.int16)
    H, W = msk.shape
    bad = np.empty((H, W), np.uint8)
    lib().orc_mask_dilatation(_p(msk, C.c_int16), H, W, win, valid, nodata, _p(bad, C.c_uint8))
    return bad


def cv_masked(cv, d0, subpix, win, mskL=None, mskR=None, valid=0, nodata=1, dmin=None, dmax=None):
    """In place."""
    assert cv.dtype == np.float32 and cv.flags.c_contiguous
    H, W, D = cv.shape
    mL = None if mskL is None else np.ascontiguousarray(mskL, np.int16)
    mR = None if mskR is None else np.ascontiguousarray(mskR, np.int16)
    gmin = None if dmin is None else np.ascontiguousarray(dmin, np.float64)
    gmax = None if dmax is None else np.ascontiguousarray(dmax, np.float64)
    lib().orc_cv_masked(_p(cv), H, W, D, int(d0), subpix, win, _p(mL, C.c_int16), _p(mR, C.c_int16),
                        valid, nodata, _p(gmin, C.c_double), _p(gmax, C.c_double))
    return cv


def median3(img):
    img = _f32(img)
    H, W = img.shape
    out = np.empty_like(img)
    lib().orc_median3(_p(img), H, W, _p(out))
    return out


def cross_support(img, len_arms, intensity):
    img = _f32(img)
    H, W = img.shape
    cross = np.empty((H, W, 4), np.int16)
    lib().orc_cross_support(_p(img), H, W, int(len_arms), C.c_float(intensity), _p(cross, C.c_int16))
    return cross


def cbca(cv, d0, subpix, offset, crossL, crossRs):
    """In place.  crossRs: list of int16 arrays (one per sub-pixel phase)."""
    assert cv.dtype == np.float32 and cv.flags.c_contiguous
    H, W, D = cv.shape
    cl = np.ascontiguousarray(crossL, np.int16)
    cr = np.concatenate([np.ascontiguousarray(c, np.int16).ravel() for c in crossRs])
    lib().orc_cbca(_p(cv), H, W, D, int(d0), subpix, offset, _p(cl, C.c_int16), _p(cr, C.c_int16))
    return cv


# the definition's path order, (drow, dcol) of the step from p-r to p; bit k of a direction mask = SGM_DIRECTIONS[k]
SGM_DIRECTIONS = ((0, 1), (0, -1), (1, 0), (1, 1), (1, -1), (-1, 0), (-1, 1), (-1, -1))


def sgm_p2maps(cv, P1, p2maps, is_max, invalid_cost, overcounting=False):
    """orc_sgm with P2 given per pixel and direction: p2maps float32 [8][H][W], definition order of the directions."""
    assert cv.dtype == np.float32 and cv.flags.c_contiguous
    H, W, D = cv.shape
    maps = np.ascontiguousarray(p2maps, np.float32)
    assert maps.shape == (8, H, W)
    out = np.empty_like(cv)
    lib().orc_sgm_p2maps(_p(cv), H, W, D, C.c_float(P1), _p(maps), int(is_max), C.c_float(invalid_cost), int(overcounting), _p(out))
    return out


def sgm(cv, P1, P2, is_max=False, invalid_cost=None, overcounting=False, dir_mask=0xFF):
    cv = _f32(cv)
    H, W, D = cv.shape
    out = np.empty_like(cv)
    lib().orc_sgm_dirs(_p(cv), H, W, D, C.c_float(P1), C.c_float(P2), int(is_max), C.c_float(invalid_cost),
                       int(overcounting), int(dir_mask), _p(out))
    return out


def wta(cv, d0, subpix, is_max=False, invalid_disparity=-9999.0, validity=None):
    cv = _f32(cv)
    H, W, D = cv.shape
    disp = np.empty((H, W), np.float32)
    val = np.zeros((H, W), np.int64) if validity is None else np.ascontiguousarray(validity, np.int64).copy()
    lib().orc_wta(_p(cv), H, W, D, C.c_double(d0), subpix, int(is_max), C.c_float(invalid_disparity),
                  _p(disp), _p(val, C.c_int64))
    return disp, val


def refine(cv, disp, validity, d_min, d_max, subpix, is_max, method):
    """method: 'vfit' | 'quadratic'.  Returns (itp, disp, validity) - copies."""
    cv = _f32(cv)
    H, W, D = cv.shape
    disp = _f32(disp).copy()
    val = np.ascontiguousarray(validity, np.int64).copy()
    itp = np.empty((H, W), np.float32)
    lib().orc_refine(_p(cv), H, W, D, C.c_double(d_min), C.c_double(d_max), subpix, int(is_max),
                     {"vfit": 0, "quadratic": 1}[method], _p(disp), _p(val, C.c_int64), _p(itp))
    return itp, disp, val


def reverse_cost_volume(cv, min_disp):
    cv = _f32(cv)
    H, W, D = cv.shape
    out = np.empty_like(cv)
    lib().orc_reverse_cost_volume(_p(cv), H, W, D, int(min_disp), _p(out))
    return out


def reverse_disp_range(left_min, left_max):
    a, b = _f32(left_min), _f32(left_max)
    H, W = a.shape
    rmin, rmax = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
    lib().orc_reverse_disp_range(_p(a), _p(b), H, W, _p(rmin), _p(rmax))
    return rmin, rmax


def cross_checking(disp_left, validity_left, disp_right, dmin, dmax, threshold):
    """validation.py:226-371 -> (validity_left updated copy, confidence float32 [H][W])."""
    dl, dr = _f32(disp_left), _f32(disp_right)
    H, W = dl.shape
    val = np.ascontiguousarray(validity_left, np.int64).copy()
    conf = np.empty((H, W), np.float32)
    lib().orc_cross_checking(_p(dl), _p(val, C.c_int64), _p(dr), H, W, int(dmin), int(dmax), C.c_double(threshold), _p(conf))
    return val, conf


def median_filter(img, size):
    img = _f32(img)
    H, W = img.shape
    out = np.empty_like(img)
    lib().orc_median_filter(_p(img), H, W, int(size), _p(out))
    return out


def filter_median_disparity(disp, validity, size):
    """median.py:94-131 -> filtered copy of the disparity map."""
    d = _f32(disp).copy()
    v = np.ascontiguousarray(validity, np.int64)
    lib().orc_filter_median_disparity(_p(d), _p(v, C.c_int64), d.shape[0], d.shape[1], int(size))
    return d


def denoise_disparity(disp, validity, color, grad_row, grad_col, filter_size, sigma_euclidian, sigma_color, sigma_planar):
    """disparity_denoiser.py:223-313 -> filtered copy of the disparity map (grad_* = np.gradient of the blurred map)."""
    d = np.ascontiguousarray(disp, np.float32).copy()
    v = np.ascontiguousarray(validity, np.int64)
    maps = [np.ascontiguousarray(m, np.float32) for m in (color, grad_row, grad_col)]
    lib().orc_denoise_disparity(_p(d), _p(v, C.c_int64), _p(maps[0]), _p(maps[1]), _p(maps[2]), d.shape[0], d.shape[1], int(filter_size),
                                C.c_double(sigma_euclidian), C.c_double(sigma_color), C.c_double(sigma_planar))
    return d


def filter_bilateral_disparity(disp, validity, sigma_color, sigma_space):
    """bilateral.py:100-255 -> filtered copy of the disparity map."""
    d = _f32(disp).copy()
    v = np.ascontiguousarray(validity, np.int64)
    lib().orc_filter_bilateral_disparity(_p(d), _p(v, C.c_int64), d.shape[0], d.shape[1], C.c_double(sigma_color),
                                         C.c_double(sigma_space))
    return d


def disparity_range(disp, validity, win, marge, gmin, gmax):
    d = _f32(disp)
    v = np.ascontiguousarray(validity, np.int64)
    lo, hi = np.empty(d.shape, np.float32), np.empty(d.shape, np.float32)
    lib().orc_disparity_range(_p(d), _p(v, C.c_int64), d.shape[0], d.shape[1], int(win), int(marge), int(gmin), int(gmax),
                              _p(lo), _p(hi))
    return lo, hi


def risk(cv, etas, grid_min, grid_max, disp_range):
    """risk.cpp:28-197 as risk.py:144-166 calls it -> (risk_max, risk_min, disp_sup, disp_inf), float32 [H][W] each."""
    cv = _f32(cv)
    H, W, D = cv.shape
    e = np.ascontiguousarray(etas, np.float64)
    gmin = np.ascontiguousarray(grid_min, np.int64)
    gmax = np.ascontiguousarray(grid_max, np.int64)
    dr = _f32(disp_range)
    outs = [np.empty((H, W), np.float32) for _ in range(4)]
    lib().orc_risk(_p(cv), H, W, D, _p(e, C.c_double), len(e), _p(gmin, C.c_int64), _p(gmax, C.c_int64), _p(dr), *[_p(o) for o in outs])
    return tuple(outs)


def interval_bounds(cv, possibility_threshold, type_factor, grid_min, grid_max, disp_range):
    """interval_bounds.cpp:28-161 -> (interval_inf, interval_sup), float32 [H][W]."""
    cv = _f32(cv)
    H, W, D = cv.shape
    gmin = np.ascontiguousarray(grid_min, np.int64)
    gmax = np.ascontiguousarray(grid_max, np.int64)
    dr = _f32(disp_range)
    lo, hi = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
    lib().orc_interval_bounds(_p(cv), H, W, D, _p(dr), C.c_float(possibility_threshold), C.c_float(type_factor),
                              _p(gmin, C.c_int64), _p(gmax, C.c_int64), _p(dr), _p(lo), _p(hi))
    return lo, hi


def ambiguity(cv, etas, grid_min, grid_max, disp_range):
    """ambiguity.cpp:28-142 -> float32 [H][W] integral of the ambiguity (before normalisation)."""
    cv = _f32(cv)
    H, W, D = cv.shape
    e = _f32(etas)
    gmin = np.ascontiguousarray(grid_min, np.int64)
    gmax = np.ascontiguousarray(grid_max, np.int64)
    dr = _f32(disp_range)
    out = np.empty((H, W), np.float32)
    lib().orc_ambiguity(_p(cv), H, W, D, _p(e), len(e), _p(gmin, C.c_int64), _p(gmax, C.c_int64), _p(dr), _p(out))
    return out


def interpolate_nodata(img, msk, invalid_bits, filled_value):
    """img_tools.cpp:99-155 -> (filled image float32, mask int32)."""
    im = _f32(img)
    mk = np.ascontiguousarray(msk, np.int32)
    out_i, out_m = np.empty_like(im), np.empty_like(mk)
    lib().orc_interpolate_nodata(_p(im), _p(mk, C.c_int32), im.shape[0], im.shape[1], int(invalid_bits), int(filled_value),
                                 _p(out_i), _p(out_m, C.c_int32))
    return out_i, out_m


INTERP_PASSES = {"occlusion_mc_cnn": 0, "mismatch_mc_cnn": 1, "occlusion_sgm": 2, "mismatch_sgm": 3}


def interpolate_disparity(which, disp, valid):
    """interpolated_disparity.cpp, one pass -> (disparity float32, validity int32)."""
    d = _f32(disp)
    v = np.ascontiguousarray(valid, np.int32)
    out_d, out_v = np.empty_like(d), np.empty_like(v)
    lib().orc_interpolate_disparity(INTERP_PASSES[which], _p(d), _p(v, C.c_int32), d.shape[0], d.shape[1], _p(out_d),
                                    _p(out_v, C.c_int32))
    return out_d, out_v


def set_threads(n):
    """OpenMP threads of the census / SGM / WTA / refinement loops (results do not depend on it): 1 = the reference's serial
    execution, 0 = every core this process may use (usable_cores()).  Returns the count in effect."""
    return int(lib().orc_set_threads(int(n) if n > 0 else usable_cores()))
