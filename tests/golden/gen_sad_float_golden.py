"""Generates tests/golden/sad_float_order.npz: SAD/SSD cost volumes on NON-integer float32 images.

The reference's SAD/SSD is pure numpy (src/pandora/matching_cost/sad_ssd.py) but importing it needs xarray /
json_checker, which this image lacks.  The float32 rounding of its result is decided entirely by two numpy
expressions, which are executed here verbatim on plain arrays:

  * the pixel-wise cost written per disparity into a NaN-filled (disp, col+2o, row+2o) buffer
    (sad_ssd.py:180-188 with ad_cost :226-283 / sd_cost :285-338 and point_interval matching_cost.py:429-482);
  * pixel_wise_aggregation (sad_ssd.py:340-368): an as_strided 5-D view summed with np.sum(view, (0, 1)).

The order numpy adds the w*w terms in (window columns outer, window rows inner, sequential float32) is what the
fixture pins; integer-valued images (all of the reference's own tests) cannot see it.

Run:  python tests/golden/gen_sad_float_golden.py      (numpy only; deterministic)
"""
import os

import numpy as np


def reference_sad_ssd(left, right, d0, ndisp, win, squared):
    H, W = left.shape
    o = win // 2
    cv = np.full((ndisp, W + 2 * o, H + 2 * o), np.nan, dtype=np.float32)           # sad_ssd.py:209-224
    centre = cv[:, o:W + o, o:H + o] if o else cv                                       # crop_cost_volume
    for k in range(ndisp):
        d = d0 + k
        p0, p1 = max(0, -d), min(W, W - d)                                              # point_interval
        q0, q1 = max(0, d), min(W, W + d)
        if p1 <= p0:
            continue
        diff = left[:, p0:p1] - right[:, q0:q1]
        cost = diff ** 2 if squared else np.abs(diff)
        centre[k, p0:p1, :] = np.swapaxes(cost, 0, 1)                                   # sad_ssd.py:183-188
    nb_disp, nx_, ny_ = cv.shape
    s_disp, s_col, s_row = cv.strides
    view = np.lib.stride_tricks.as_strided(
        cv, (win, win, nb_disp, nx_ - (win - 1), ny_ - (win - 1)), (s_row, s_col, s_disp, s_col, s_row),
        writeable=False)
    agg = np.sum(view, (0, 1))                                                          # sad_ssd.py:367
    out = np.swapaxes(agg, 0, 2).copy()                                                 # (row, col, disp)
    if o:                                                                               # sad_ssd.py:199-204
        out[:o] = np.nan
        out[-o:] = np.nan
        out[:, :o] = np.nan
        out[:, -o:] = np.nan
    return out


def main():
    rng = np.random.default_rng(20260928)
    H, W = 14, 19
    left = (rng.random((H, W)) * 255).astype(np.float32)
    right = (rng.random((H, W)) * 255).astype(np.float32)
    out = {"left": left, "right": right}
    for win in (3, 5):
        for squared in (0, 1):
            out[f"w{win}_sq{squared}_d-3_n7"] = reference_sad_ssd(left, right, -3, 7, win, squared)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "sad_float_order.npz"), **out)


if __name__ == "__main__":
    main()
