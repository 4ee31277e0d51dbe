"""Known-answer vectors TRANSCRIBED from the reference's own tests (data only: inputs and expected
outputs; no reference code).  Citations are relative to /root/reference/tests.

Masks of NaN positions are written as strings, one character per cell: 'T' = NaN expected,
'F' = finite expected; blocks are (disp, row, col) exactly as the reference tests print them.
"""
import numpy as np

n = float("nan")


def nanmask(blocks):
    """(disp,row,col) list of row strings -> bool array (row, col, disp)"""
    a = np.array([[[ch == "T" for ch in row] for row in blk] for blk in blocks])
    return np.moveaxis(a, 0, -1)


# common.py:56-82 matching_cost_tests_setup
MC_LEFT = [[1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 2, 1], [1, 1, 1, 4, 3, 1], [1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1]]
MC_RIGHT = [[1, 1, 1, 2, 2, 2], [1, 1, 1, 4, 2, 4], [1, 1, 1, 4, 4, 1], [1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1]]

CENSUS = [
    # test_matching_cost/test_matching_cost_census.py:65-139 (window 3, d in [-1,1]); slices per disparity
    dict(
        cite="test_matching_cost_census.py:65-139",
        left=[[1, 1, 1, 3], [1, 2, 1, 0], [2, 1, 0, 1], [1, 1, 1, 1]],
        right=[[5, 1, 2, 3], [1, 2, 1, 0], [2, 2, 0, 1], [1, 1, 1, 1]],
        win=3, subpix=1, dmin=-1, dmax=1,
        expected_dhw=[
            [[n, n, n, n], [n, n, 3, n], [n, n, 7, n], [n, n, n, n]],
            [[n, n, n, n], [n, 1, 2, n], [n, 2, 0, n], [n, n, n, n]],
            [[n, n, n, n], [n, 4, n, n], [n, 5, n, n], [n, n, n, n]],
        ],
    ),
    # test_matching_cost_census.py:637-685 (window 3, subpix 2, full volume), blocks are (disp,row,col)
    dict(
        cite="test_matching_cost_census.py:637-685",
        left=[[4, 0, 4, 0, 4], [4, 1, 2, 3, 0], [0, 4, 0, 0, 0]],
        right=[[0, 0, 0, 0, 4], [4, 1, 2, 3, 0], [0, 4, 4, 0, 4]],
        win=3, subpix=2, dmin=-1, dmax=1,
        expected_dhw=[
            [[n, n, n, n, n], [n, n, 5, 5, n], [n, n, n, n, n]],
            [[n, n, n, n, n], [n, n, 4, 3, n], [n, n, n, n, n]],
            [[n, n, n, n, n], [n, 3, 2, 3, n], [n, n, n, n, n]],
            [[n, n, n, n, n], [n, 4, 2, n, n], [n, n, n, n, n]],
            [[n, n, n, n, n], [n, 4, 4, n, n], [n, n, n, n, n]],
        ],
    ),
]

# test_pandora_image.py:62-98: census bit strings (window 3) of the 3x4 interior of a 5x6 image are
# tested on img_tools.census_transform (python); the image is common.py matching_cost_tests_setup left.
# Expected codes for window 5 on the same image: single interior pixel row.
CENSUS_BITS_W5 = dict(cite="test_pandora_image.py:62-79", image=MC_LEFT, win=5,
                      # centre pixels (2,2) and (2,3)
                      expected=[0b0000000001000110000000000, 0b0])
CENSUS_BITS_W3 = dict(cite="test_pandora_image.py:62-75", image=MC_LEFT, win=3,
                      expected=[[0b000000000, 0b000000001, 0b000001011, 0b000000110],
                                [0b000000000, 0b000001000, 0b000000000, 0b000100000],
                                [0b000000000, 0b001000000, 0b011000000, 0b110000000]])

SAD_SSD = [
    # test_matching_cost_sad.py:59-122: pixel-wise AD (window 1) at disparity 0, then SAD window 5
    dict(cite="test_matching_cost_sad.py:64-85", left=MC_LEFT, right=MC_RIGHT, win=1, subpix=1, dmin=-1, dmax=1,
         squared=False, disp_index=1, masked=False,
         expected=[[0, 0, 0, 1, 1, 1], [0, 0, 0, 3, 0, 3], [0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0]]),
    dict(cite="test_matching_cost_sad.py:87-122", left=MC_LEFT, right=MC_RIGHT, win=5, subpix=1, dmin=-1, dmax=1,
         squared=False, disp_index=1, masked=True,
         expected=[[n] * 6, [n] * 6, [n, n, 6.0, 10.0, n, n], [n] * 6, [n] * 6]),
    # test_matching_cost_ssd.py:57-119
    dict(cite="test_matching_cost_ssd.py:62-83", left=MC_LEFT, right=MC_RIGHT, win=1, subpix=1, dmin=-1, dmax=1,
         squared=True, disp_index=1, masked=False,
         expected=[[0, 0, 0, 1, 1, 1], [0, 0, 0, 9, 0, 9], [0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0]]),
    dict(cite="test_matching_cost_ssd.py:85-119", left=MC_LEFT, right=MC_RIGHT, win=5, subpix=1, dmin=-1, dmax=1,
         squared=True, disp_index=1, masked=True,
         expected=[[n] * 6, [n] * 6, [n, n, 12.0, 22.0, n, n], [n] * 6, [n] * 6]),
]

# test_matching_cost_sad.py:207-277: full volume (row, col, disp), window 3, d in [-2,1]
SAD_FULL = dict(
    cite="test_matching_cost_sad.py:207-277",
    left=[[1, 2, 1, 4], [6, 2, 7, 4], [1, 1, 3, 6]],
    right=[[6, 7, 8, 10], [2, 4, 1, 6], [9, 10, 1, 2]],
    win=3, subpix=1, dmin=-2, dmax=1,
    expected=[
        [[n, n, n, n], [n, n, n, n], [n, n, n, n], [n, n, n, n]],
        [[n, n, n, n], [n, n, 48, 35], [n, 40, 43, n], [n, n, n, n]],
        [[n, n, n, n], [n, n, n, n], [n, n, n, n], [n, n, n, n]],
    ],
)

# test_matching_cost_zncc.py:125-199: SAD window 3 subpix 2, d in [-2,2], full volume (row,col,disp)
SAD_SUBPIX = dict(
    cite="test_matching_cost_zncc.py:125-199",
    left=[[7, 8, 1, 0, 2], [4, 5, 2, 1, 0], [8, 9, 10, 0, 0]],
    right=[[1, 5, 6, 3, 4], [2, 5, 10, 6, 9], [0, 7, 5, 3, 1]],
    win=3, subpix=2, dmin=-2, dmax=2,
    expected=[
        [[n] * 9] * 5,
        [[n] * 9, [n, n, n, n, 39, 32.5, 28, 34.5, 41], [n, n, 49, 41.5, 34, 35.5, 37, n, n],
         [45, 42.5, 40, 40.5, 41, n, n, n, n], [n] * 9],
        [[n] * 9] * 5,
    ],
)

# test_matching_cost_zncc.py:57-122: ZNCC window 5 on MC_LEFT/MC_RIGHT, d in [-1,1]; the reference
# computes the expectation with numpy mean/std of the two 5x5 patches (rtol 1e-5); row 2 of the volume.
ZNCC = dict(cite="test_matching_cost_zncc.py:57-122", left=MC_LEFT, right=MC_RIGHT, win=5, subpix=1, dmin=-1, dmax=1,
            # (disp index, left col slice, right col slice, column of the finite value)
            checks=[(0, (1, 6), (0, 5), 3), (2, (0, 5), (1, 6), 2)])

# test_matching_cost/test_matching_cost.py:699-1130 TestCvMasked (runs for census, sad, ssd, zncc)
_L45 = [[1, 1, 1, 3, 4], [1, 2, 1, 0, 2], [2, 1, 0, 1, 2], [1, 1, 1, 1, 4]]
_R45 = [[5, 1, 2, 3, 4], [1, 2, 1, 0, 2], [2, 2, 0, 1, 4], [1, 1, 1, 1, 2]]
_L67 = [[0, 0, 0, 0, 0, 0, 0], [0, 1, 1, 1, 3, 4, 0], [0, 1, 2, 1, 0, 2, 0], [0, 2, 1, 0, 1, 2, 0],
        [0, 1, 1, 1, 1, 4, 0], [0, 0, 0, 0, 0, 0, 0]]
_R67 = [[0, 0, 0, 0, 0, 0, 0], [0, 5, 1, 2, 3, 4, 0], [0, 1, 2, 1, 0, 2, 0], [0, 2, 2, 0, 1, 4, 0],
        [0, 1, 1, 1, 1, 2, 0], [0, 0, 0, 0, 0, 0, 0]]
CV_MASKED = [
    dict(cite="test_matching_cost.py:811-851 (invalids on left only)", left=_L45, right=_R45, win=3, subpix=1,
         dmin=-1, dmax=1, valid=0, nodata=1,
         left_mask=[[0, 0, 2, 0, 1], [0, 2, 0, 0, 0], [0, 0, 0, 0, 0], [1, 0, 0, 0, 2]],
         right_mask=[[0] * 5] * 4,
         nan=[["TTTTT", "TTFTT", "TTFFT", "TTTTT"], ["TTTTT", "TTFTT", "TTFFT", "TTTTT"],
              ["TTTTT", "TTFTT", "TTFTT", "TTTTT"]]),
    dict(cite="test_matching_cost.py:852-893 (invalids on right only)", left=_L45, right=_R45, win=3, subpix=1,
         dmin=-1, dmax=1, valid=0, nodata=1,
         left_mask=[[0] * 5] * 4,
         right_mask=[[0, 0, 0, 0, 2], [0, 1, 0, 0, 0], [0, 2, 0, 2, 0], [1, 0, 0, 0, 0]],
         nan=[["TTTTT", "TTTTT", "TTTTT", "TTTTT"], ["TTTTT", "TTTFT", "TTTTT", "TTTTT"],
              ["TTTTT", "TTFTT", "TTTTT", "TTTTT"]]),
    dict(cite="test_matching_cost.py:894-942 (invalids on both sides)", left=_L45, right=_R45, win=3, subpix=1,
         dmin=-1, dmax=1, valid=0, nodata=1,
         left_mask=[[1, 0, 0, 2, 0], [0, 0, 0, 0, 0], [0, 0, 2, 0, 0], [2, 0, 0, 0, 1]],
         right_mask=[[0, 2, 0, 0, 1], [0, 0, 0, 0, 0], [0, 0, 0, 2, 0], [1, 0, 2, 0, 0]],
         nan=[["TTTTT", "TTFFT", "TTTTT", "TTTTT"], ["TTTTT", "TTFTT", "TTTTT", "TTTTT"],
              ["TTTTT", "TTTTT", "TFTTT", "TTTTT"]]),
    dict(cite="test_matching_cost.py:943-1008 (both sides, window 5)", left=_L67, right=_R67, win=5, subpix=1,
         dmin=-1, dmax=1, valid=0, nodata=1,
         left_mask=[[2, 0, 0, 0, 0, 0, 1], [0] * 7, [0, 2, 0, 0, 0, 0, 0], [0, 0, 0, 2, 0, 0, 0],
                    [0, 0, 0, 0, 0, 2, 0], [1, 0, 0, 0, 0, 0, 2]],
         right_mask=[[1, 0, 0, 0, 0, 0, 2], [0] * 7, [2, 0, 2, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 2], [0] * 7,
                     [2, 0, 0, 0, 0, 0, 1]],
         nan=[["TTTTTTT", "TTTTTTT", "TTTTTTT", "TTTTFTT", "TTTTTTT", "TTTTTTT"],
              ["TTTTTTT", "TTTTTTT", "TTTFTTT", "TTTTTTT", "TTTTTTT", "TTTTTTT"],
              ["TTTTTTT", "TTTTTTT", "TTFFTTT", "TTTTTTT", "TTTTTTT", "TTTTTTT"]]),
    dict(cite="test_matching_cost.py:1040-1106 (subpix 2)", left=_L45, right=_R45, win=3, subpix=2,
         dmin=-1, dmax=1, valid=5, nodata=7,
         left_mask=[[5, 56, 5, 12, 5], [5, 5, 5, 5, 5], [5, 5, 5, 5, 5], [3, 5, 4, 5, 7]],
         right_mask=[[7, 5, 5, 5, 5], [5, 5, 5, 65, 5], [5, 5, 5, 5, 5], [5, 23, 5, 5, 2]],
         nan=[["TTTTT", "TTTFT", "TTFTT", "TTTTT"], ["TTTTT", "TTTTT", "TTFTT", "TTTTT"],
              ["TTTTT", "TTFTT", "TFFTT", "TTTTT"], ["TTTTT", "TTTTT", "TFFTT", "TTTTT"],
              ["TTTTT", "TFTTT", "TFFTT", "TTTTT"]]),
]

# test_disparity.py:54-197 to_disp: SAD window 1, WTA with invalid_disparity 0
WTA = dict(
    cite="test_disparity.py:54-197",
    left=[[1, 2, 4, 6], [2, 4, 1, 6], [6, 7, 8, 10]],
    right=[[6, 1, 2, 4], [6, 2, 4, 1], [10, 6, 7, 8]],
    cases=[((-3, 1), [[1, 1, 1, -3], [1, 1, 1, -3], [1, 1, 1, -3]]),
           ((-3, -1), [[0, -1, -2, -3], [0, -1, -1, -3], [0, -1, -2, -3]]),
           ((1, 3), [[1, 1, 1, 0], [1, 1, 1, 0], [1, 1, 1, 0]])],
)

# test_aggregation.py:50-96 setUp, :214-245 cross arms, :247-288 aggregated volume (rtol 1e-7)
CBCA = dict(
    cite="test_aggregation.py:50-96,214-288",
    left=[[5, 1, 15, 7, 3], [10, 9, 11, 9, 6], [1, 18, 4, 5, 9]],
    right=[[1, 5, 1, 15, 7], [2, 10, 9, 11, 9], [3, 1, 18, 4, 5]],
    distance=3, intensity=5.0,
    arms_top=[[0, 0, 0, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 2, 1]],
    arms_bottom=[[1, 1, 1, 2, 1], [1, 1, 1, 1, 1], [0, 0, 0, 0, 0]],
    arms_left=[[0, 1, 1, 1, 1], [0, 1, 2, 2, 1], [0, 1, 1, 1, 1]],
    arms_right=[[1, 1, 1, 1, 0], [2, 2, 1, 1, 0], [1, 1, 1, 1, 0]],
    aggregated=[
        [[n, (4 + 4 + 8 + 1) / 4, 0.0],
         [(0 + 7 + 10 + 1) / 4, (4 + 4 + 14 + 8 + 1 + 2) / 6, 0.0],
         [(0 + 10 + 6 + 7 + 1 + 0) / 6, (14 + 4 + 8 + 1 + 2 + 2 + 3) / 7, 0.0],
         [(10 + 6 + 12 + 1 + 0 + 5) / 6, (14 + 8 + 4 + 2 + 2 + 3) / 6, 0.0],
         [(6 + 12 + 0 + 5) / 4, (8 + 4 + 2 + 3 + 2) / 5, n]],
        [[n, (4 + 4 + 8 + 1 + 2 + 17) / 6, 0.0],
         [(0 + 10 + 7 + 1 + 15 + 3) / 6, (4 + 4 + 14 + 8 + 1 + 2 + 2 + 17 + 14) / 9, 0.0],
         [(0 + 10 + 6 + 7 + 1 + 0 + 15 + 3 + 13) / 9, (4 + 14 + 8 + 1 + 2 + 2 + 3 + 17 + 14 + 1) / 10, 0.0],
         [(10 + 6 + 12 + 1 + 0 + 5 + 3 + 13 + 5) / 9, (14 + 8 + 4 + 2 + 2 + 3 + 14 + 1 + 4) / 9, 0.0],
         [(6 + 12 + 0 + 5 + 13 + 5) / 6, (2 + 8 + 4 + 2 + 3 + 1 + 4) / 7, n]],
        [[n, (2 + 8 + 1 + 17) / 4, 0.0],
         [(7 + 1 + 15 + 3) / 4, (8 + 1 + 2 + 2 + 17 + 14) / 6, 0.0],
         [(7 + 1 + 0 + 15 + 3 + 13) / 6, (1 + 2 + 2 + 17 + 14 + 1 + 3) / 7, 0.0],
         [(1 + 0 + 5 + 3 + 13 + 5) / 6, (2 + 2 + 3 + 14 + 1 + 4) / 6, 0.0],
         [(0 + 5 + 13 + 5) / 4, (2 + 2 + 3 + 1 + 4) / 5, n]],
    ],
)

# test_cpp/test_matching_cost/test_matching_cost.cpp known answers are covered by the compiled
# reference (oracle/_ref) in tests/test_oracle_vs_reference.py.

# test_filter.py:52-118 datasets 1-2 + :198-212 expected (MedianFilter.filter_disparity, filter_size 3).
# validity values use constants.py bits; invalid = value & 0b01111000011 != 0.
MEDIAN = [
    dict(cite="test_filter.py:54-70,204",
         disp=[[5, 6, 7, 8, 9], [6, 85, 1, 36, 5], [5, 9, 23, 12, 2], [6, 1, 9, 2, 4]],
         valid=[[0, 0, 0, 0, 0], [0, 4, 0, 0, 0], [0, 16, 0, 0, 0], [0, 0, 0, 0, 8]],
         expected=[[5, 6, 7, 8, 9], [6, 6, 9, 8, 5], [5, 6, 9, 5, 2], [6, 1, 9, 2, 4]]),
    dict(cite="test_filter.py:72-118,210",
         disp=[[7, 8, 4, 5, 5], [5, 9, 4, 3, 8], [5, 2, 7, 2, 2], [6, 1, 9, 2, 4]],
         valid=[[4, 0, 4, 16 + 1, 0], [128, 1, 256, 0, 0], [64, 512, 2, 4 + 8, 0], [2, 256, 64, 0, 2]],
         expected=[[7, 8, 4, 5, 5], [5, 9, 4, 3.5, 8], [5, 2, 7, 2, 2], [6, 1, 9, 2, 4]]),
    dict(cite="test_filter.py:120-148,216 (dataset3)",
         disp=[[7, 8, 4, 5, 5], [5, 9, 4, 3, 8], [5, 2, 7, 2, 2], [6, 1, 9, 2, 4]],
         valid=[[4, 0, 4, 16 + 1, 0], [0, 0, 8, 0, 0], [0, 0, 0, 4 + 8, 0], [128, 0, 0, 0, 0]],
         expected=[[7, 8, 4, 5, 5], [5, 5, 4, 4, 8], [5, 5, 3, 4, 2], [6, 1, 9, 2, 4]]),
    dict(cite="test_filter.py:150-190,219-226 (dataset4, filter_size 5)", size=5,
         disp=[[7, 8, 4, 5, 5], [5, 9, 4, 3, 8], [5, 2, 7, 2, 2], [6, 1, 9, 2, 4], [1, 6, 2, 7, 8]],
         valid=[[4, 0, 4, 16 + 1, 0], [0, 0, 8, 0, 0], [0, 0, 0, 4 + 8, 0], [128, 0, 0, 0, 0], [64, 0, 4, 2 + 8, 0]],
         expected=[[7, 8, 4, 5, 5], [5, 9, 4, 3, 8], [5, 2, 5, 2, 2], [6, 1, 9, 2, 4], [1, 6, 2, 7, 8]]),
]


# ---- validation / cross-checking (tests/test_validation.py) -------------------------------------
_RND = 1 << 1   # PANDORA_MSK_PIXEL_RIGHT_NODATA_OR_DISPARITY_RANGE_MISSING
_LBORDER = 1 << 0  # PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER
_OCC, _MIS = 1 << 8, 1 << 9
_nan = float("nan")
CROSS_CHECKING = [
    {"cite": "test_validation.py:104-140 test_cross_checking",
     "left": [[0, -1, 1, -2], [2, 2, -1, 0]], "right": [[0, 2, -1, -1], [1, 1, -2, -1]],
     "validity": [[0, 0, 0, _RND], [0, 0, 0, 0]], "interval": (-2, 2), "threshold": 0.0,
     "conf": [[0.0, 1.0, 0.0, _nan], [0.0, 1.0, 0.0, 1.0]],
     "mask": [[0, _MIS, 0, _RND], [0, _MIS, 0, _OCC]]},
    {"cite": "test_validation.py:142-252 test_distance_lr_rl",
     "left": [[_nan] * 4, [_nan, 1, -1, _nan], [_nan] * 4], "right": [[_nan] * 4, [_nan, 0, -1, _nan], [_nan] * 4],
     "validity": [[_LBORDER] * 4, [_LBORDER, 0, 0, _LBORDER], [_LBORDER] * 4], "interval": (-1, 1), "threshold": 0.0,
     "conf": [[_nan] * 4, [_nan, 0.0, 1.0, _nan], [_nan] * 4], "mask": None},
    {"cite": "test_validation.py:255-308 test_cross_checking_float_disparity",
     "left": [[0, -1.2, 1, -2], [2, 1.8, -1, 0]], "right": [[0, 2, -1.2, -1], [0.8, 1, -2, -1]],
     "validity": [[0, 0, 0, _RND], [0, 0, 0, 0]], "interval": (-2, 2), "threshold": 0.0,
     "conf": None, "mask": [[0, _MIS, 0, _RND], [0, _MIS, 0, _OCC]]},
]


# ---- multiscale (tests/test_multiscale.py:52-137 test_disparity_range; window 3, marge 0, range [-30, 0]) ------
_INC, _STOP = 1 << 2, 1 << 3  # information bits (RIGHT_INCOMPLETE_DISPARITY_RANGE, STOPPED_INTERPOLATION): still valid
DISPARITY_RANGE = {
    "cite": "test_multiscale.py:52-137",
    "disp": [[-1, -2, -3, -4, -5, -6], [-7, -8, -9, _nan, -11, -12], [-13, -14, -15, -16, -17, -18],
             [-19, -20, -21, -22, -23, -24], [_nan, -26, -27, -28, -29, -30]],
    "validity": [[_INC] * 6, [0] * 6, [0] * 6, [_LBORDER] * 6, [_STOP] * 6],
    "window_size": 3, "marge": 0, "dmin": -30, "dmax": 0,
    "range_max": [[0, 0, 0, 0, 0, 0], [0, -1, -2, 0, -4, 0], [0, -7, -8, -9, -11, 0], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0]],
    "range_min": [[-30] * 6, [-30, -15, -16, -30, -18, -30], [-30, -15, -16, -17, -18, -30], [-30] * 6, [-30] * 6],
}


# ---- interpolation of rejected pixels (tests/test_validation.py:310-704; validity flags as constants.py) ----------
_FOCC, _FMIS, _INVL = 1 << 4, 1 << 5, 1 << 6   # FILLED_OCCLUSION, FILLED_MISMATCH, IN_VALIDITY_MASK_LEFT
_D45 = [[0, 1.2, -2, -1, -2], [1, 0, 1, 0, 0], [2, 1, -1, -2, -1], [1, -1, 1, -1, -1.3]]


def _m45(flag, corner=0, last=0):
    return [[_LBORDER, _INC, 0, _STOP, corner], [0, 0, flag, 0, 0], [0, _STOP, flag, _INVL, flag], [last, flag, 0, 0, 0]]


def _med(*v):
    import numpy as _np
    return float(_np.float32(_np.median(v)))   # the expected maps are float32 arrays of np.median values


INTERPOLATION = [
    {"cite": "test_validation.py:310-371 test_interpolate_occlusion_mc_cnn", "method": "mc-cnn",
     "disp": [[0, -1, 1, -2.1], [2, 2, -1.7, 0]], "validity": [[_RND, _OCC, _RND, 0], [_OCC, _INVL, 0, _OCC]],
     "out_disp": [[0, -2.1, 1, -2.1], [-1.7, 2, -1.7, -1.7]], "out_validity": [[_RND, _FOCC, _RND, 0], [_FOCC, _INVL, 0, _FOCC]]},
    {"cite": "test_validation.py:374-460 test_interpolate_mismatch_mc_cnn", "method": "mc-cnn",
     "disp": _D45, "validity": _m45(_MIS), "out_validity": _m45(_FMIS),
     "out_disp": [[0, 1.2, -2, -1, -2],
                  [1, 0, _med(1.2, 1, 0, 0, 0, 1, -2, -2, -2, -1, 0, 0, 0, -1, -1.3), 0, 0],
                  [2, 1, _med(1, 1, 1, 1, 1, 0, 1, -2, -1, 0, 0, -1, -1, 1), -2, _med(-1, -1, -1, 1, 1, 0, 0, 0, 0, 0)],
                  [1, _med(1, 1, 1, 2, 1, 1, 1, 0, 1, 1, 1), 1, -1, -1.3]]},
    {"cite": "test_validation.py:462-533 test_interpolate_occlusion_sgm", "method": "sgm",
     "disp": _D45, "validity": _m45(_OCC), "out_validity": _m45(_FOCC),
     "out_disp": [[0, 1.2, -2, -1, -2], [1, 0, 0, 0, 0], [2, 1, 0, -2, 0], [1, 1, 1, -1, -1.3]]},
    {"cite": "test_validation.py:536-613 test_interpolate_mismatch_sgm", "method": "sgm",
     "disp": _D45, "validity": _m45(_MIS), "out_validity": _m45(_FMIS),
     "out_disp": [[0, 1.2, -2, -1, -2], [1, 0, _med(1.2, -2, -1, 0, 0, 1, 1, -1.3), 0, 0],
                  [2, 1, _med(-2, 0, -1, -1, 1, 1, 0), -2, _med(0, -1.3, -1, 1, 0)], [1, _med(2, 1, 0, 1, 1), 1, -1, -1.3]]},
    {"cite": "test_validation.py:616-704 test_interpolate_mismatch_and_occlusion_sgm", "method": "sgm",
     "disp": [[0, 1, -2, -1, -2], [1, 0, 1, 0, 0], [2, 1, -1, -2, -1], [1, -1, 1, -1, -1]],
     "validity": _m45(_MIS, corner=_OCC, last=_OCC),
     "out_validity": [[_LBORDER, _INC, 0, _STOP, _FOCC], [0, 0, _FMIS, 0, 0], [0, _STOP, _FMIS, _INVL, _FMIS], [_FOCC, _FOCC, 0, 0, 0]],
     "out_disp": [[0, 1, -2, -1, 0], [1, 0, _med(1, 1, 0, 1, -2, -1, 0, -1), 0, 0],
                  [2, 1, _med(1, 1, 0, -2, 0, -1), -2, _med(-1, -1, 1, 0, 0)], [1, 1, 1, -1, -1]]},
]


# ---- more CBCA vectors (tests/test_aggregation.py) -------------------------------------------------------------
# arms are (left, right, top, bottom) per pixel, as aggregation_cpp.cross_support returns them
_AL3 = [[5, 1, 15, 7, 3], [10, 9, 11, 9, 6], [1, 18, 4, 5, 9]]
_AR3 = [[1, 5, 1, 15, 7], [2, 10, 9, 11, 9], [3, 1, 18, 4, 5]]
_AL4 = _AL3 + [[5, 1, 15, 7, 3]]
_AR4 = _AR3 + [[1, 5, 1, 15, 7]]


def _stack(left, right, top, bottom):
    return [[[left[r][c], right[r][c], top[r][c], bottom[r][c]] for c in range(len(left[0]))] for r in range(len(left))]


CROSS_SUPPORTS = [
    dict(cite="test_aggregation.py:485-562 test_computes_cross_support (no masks)", left=_AL3, right=_AR3, msk_left=None,
         msk_right=None, win=1, subpix=1, intensity=5.0, distance=3,
         arms_left=_stack([[0, 1, 1, 1, 1], [0, 1, 2, 2, 2], [0, 1, 1, 1, 1]], [[1, 1, 1, 1, 0], [2, 2, 2, 1, 0], [1, 1, 1, 1, 0]],
                          [[0, 0, 0, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 2, 1]], [[1, 1, 1, 2, 1], [1, 1, 1, 1, 1], [0, 0, 0, 0, 0]]),
         right_index=0,
         arms_right=_stack([[0, 1, 2, 1, 1], [0, 1, 1, 1, 2], [0, 1, 1, 1, 1]], [[2, 1, 1, 1, 0], [1, 1, 2, 1, 0], [1, 1, 1, 1, 0]],
                           [[0, 0, 0, 0, 0], [1, 1, 1, 1, 1], [2, 2, 1, 1, 2]], [[2, 2, 1, 1, 2], [1, 1, 1, 1, 1], [0, 0, 0, 0, 0]])),
    dict(cite="test_aggregation.py:564-666 test_computes_cross_support (invalid / no-data pixels)", left=_AL3, right=_AR3,
         msk_left=[[2, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 3, 0, 0, 0]], msk_right=[[0, 0, 0, 0, 0], [0, 1, 0, 3, 0], [0, 0, 0, 0, 0]],
         win=1, subpix=1, intensity=6.0, distance=3,
         arms_left=_stack([[0, 0, 1, 1, 1], [0, 1, 2, 0, 0], [0, 0, 0, 1, 2]], [[0, 1, 1, 1, 0], [2, 1, 0, 0, 0], [0, 0, 2, 1, 0]],
                          [[0, 0, 0, 0, 0], [0, 1, 1, 0, 1], [1, 0, 1, 0, 1]], [[0, 1, 1, 0, 1], [1, 0, 1, 0, 1], [0, 0, 0, 0, 0]]),
         right_index=0,
         arms_right=_stack([[0, 1, 2, 1, 1], [0, 0, 0, 0, 0], [0, 1, 1, 1, 1]], [[2, 1, 1, 1, 0], [0, 0, 0, 0, 0], [1, 1, 1, 1, 0]],
                           [[0, 0, 0, 0, 0], [1, 0, 1, 0, 1], [2, 0, 1, 0, 2]], [[2, 0, 1, 0, 2], [1, 0, 1, 0, 1], [0, 0, 0, 0, 0]])),
    dict(cite="test_aggregation.py:668-735 test_computes_cross_support_with_subpixel (half-pixel right image)", left=_AL3, right=_AR3,
         msk_left=None, msk_right=None, win=1, subpix=2, intensity=5.0, distance=3, arms_left=None, right_index=1,
         arms_right=_stack([[0, 1, 1, 1], [0, 1, 2, 2], [0, 1, 1, 1]], [[1, 1, 1, 0], [2, 2, 1, 0], [1, 1, 1, 0]],
                           [[0, 0, 0, 0], [1, 1, 1, 1], [2, 1, 2, 1]], [[2, 1, 2, 1], [1, 1, 1, 1], [0, 0, 0, 0]])),
    dict(cite="test_aggregation.py:737-808 test_computes_cross_support_with_subpixel (masks)", left=_AL3, right=_AR3,
         msk_left=[[0, 0, 0, 0, 0], [0, 1, 0, 3, 0], [0, 0, 0, 0, 0]], msk_right=[[2, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 3, 0, 0, 0]],
         win=1, subpix=2, intensity=6.0, distance=3, arms_left=None, right_index=1,
         arms_right=_stack([[0, 0, 1, 1], [0, 1, 0, 0], [0, 0, 0, 1]], [[0, 1, 1, 0], [1, 0, 0, 0], [0, 0, 1, 0]],
                           [[0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0]], [[0, 1, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]])),
    dict(cite="test_aggregation.py:810-897 test_computes_cross_support_with_offset (window 3)", left=_AL4, right=_AR4,
         msk_left=None, msk_right=None, win=3, subpix=1, intensity=5.0, distance=3,
         arms_left=_stack([[0, 1, 2], [0, 1, 2]], [[2, 1, 0], [2, 1, 0]], [[0, 0, 0], [1, 1, 1]], [[1, 1, 1], [0, 0, 0]]),
         right_index=0,
         arms_right=_stack([[0, 1, 1], [0, 1, 1]], [[1, 1, 0], [1, 1, 0]], [[0, 0, 0], [1, 1, 1]], [[1, 1, 1], [0, 0, 0]])),
]

# SAD (window `win`, `subpix`) -> cv_masked -> CBCA(intensity 5, distance 3), disparities [-1, 1]; rtol 1e-7 as in the reference
CBCA_PIPELINES = [
    dict(cite="test_aggregation.py:91-212 test_compute_cbca_subpixel", left=_AL3, right=_AR3, msk_left=None, msk_right=None,
         win=1, subpix=2, disp_index=None,
         expected=[
             [[n, n, (4 + 4 + 8 + 1) / 4, (2 + 2 + 4 + 0.5 + 1) / 5, 0.0],
              [(0 + 7 + 10 + 1) / 4, (2 + 12 + 3 + 1.5 + 1) / 5, (4 + 4 + 14 + 8 + 1 + 2) / 6, (2 + 2 + 7 + 4 + 0.5 + 1 + 1) / 7, 0.0],
              [(0 + 10 + 6 + 7 + 1 + 0) / 6, (2 + 12 + 1 + 3 + 1.5 + 1 + 4) / 7, (14 + 4 + 8 + 1 + 2 + 2 + 3) / 7,
               (2 + 7 + 4 + 4 + 0.5 + 1 + 1) / 7, 0.0],
              [(10 + 6 + 12 + 1 + 0 + 5) / 6, (12 + 1 + 8 + 3 + 1.5 + 1 + 4 + 6 + 5.5 + 4.5) / 10, (14 + 8 + 4 + 2 + 2 + 3) / 6,
               (7 + 4 + 0.5 + 1 + 1) / 5, 0.0],
              [(6 + 12 + 0 + 5) / 4, (1 + 8 + 1.5 + 1 + 4) / 5, (8 + 4 + 2 + 3 + 2) / 5, n, n]],
             [[n, n, (4 + 4 + 8 + 1 + 2 + 17) / 6, (2 + 2 + 4 + 0.5 + 1 + 1 + 8.5) / 7, 0.0],
              [(0 + 10 + 7 + 1 + 15 + 3) / 6, (2 + 12 + 3 + 1.5 + 1 + 16 + 5.5) / 7, (4 + 4 + 14 + 8 + 1 + 2 + 2 + 17 + 14) / 9,
               (2 + 2 + 7 + 4 + 0.5 + 1 + 1 + 1 + 8.5 + 7) / 10, 0.0],
              [(0 + 10 + 6 + 7 + 1 + 0 + 15 + 3 + 13) / 9, (2 + 12 + 1 + 3 + 1.5 + 1 + 4 + 16 + 5.5 + 6) / 10,
               (4 + 14 + 8 + 1 + 2 + 2 + 3 + 17 + 14 + 1) / 10, (2 + 7 + 4 + 4 + 0.5 + 1 + 1 + 8.5 + 7 + 0.5) / 10, 0.0],
              [(10 + 6 + 12 + 1 + 0 + 5 + 3 + 13 + 5) / 9, (12 + 1 + 8 + 3 + 1.5 + 1 + 4 + 5.5 + 6 + 4.5) / 10,
               (14 + 8 + 4 + 2 + 2 + 3 + 14 + 1 + 4) / 9, (7 + 4 + 0.5 + 1 + 1 + 7 + 0.5) / 7, 0.0],
              [(6 + 12 + 0 + 5 + 13 + 5) / 6, (1 + 8 + 1.5 + 1 + 4 + 6 + 4.5) / 7, (2 + 8 + 4 + 2 + 3 + 1 + 4) / 7, n, n]],
             [[n, n, (2 + 8 + 1 + 17) / 4, (4 + 0.5 + 1 + 1 + 8.5) / 5, 0.0],
              [(7 + 1 + 15 + 3) / 4, (3 + 1.5 + 1 + 16 + 5.5) / 5, (8 + 1 + 2 + 2 + 17 + 14) / 6, (4 + 0.5 + 1 + 1 + 1 + 8.5 + 7) / 7, 0.0],
              [(7 + 1 + 0 + 15 + 3 + 13) / 6, (3 + 1.5 + 1 + 4 + 16 + 5.5 + 6) / 7, (1 + 2 + 2 + 17 + 14 + 1 + 3) / 7,
               (4 + 0.5 + 1 + 1 + 8.5 + 7 + 0.5) / 7, 0.0],
              [(1 + 0 + 5 + 3 + 13 + 5) / 6, (1 + 8 + 3 + 1.5 + 1 + 4 + 5.5 + 6 + 4.5 + 12) / 10, (2 + 2 + 3 + 14 + 1 + 4) / 6,
               (0.5 + 1 + 1 + 7 + 0.5) / 5, 0.0],
              [(0 + 5 + 13 + 5) / 4, (1.5 + 1 + 4 + 6 + 4.5) / 5, (2 + 2 + 3 + 1 + 4) / 5, n, n]]]),
    dict(cite="test_aggregation.py:305-390 test_compute_cbca_with_invalid_cost (slice d=0)", left=_AL4, right=_AR4,
         msk_left=[[0, 1, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 1, 0], [3, 0, 0, 0, 0]],
         msk_right=[[0, 0, 0, 0, 0], [0, 0, 5, 1, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]], win=1, subpix=1, disp_index=1,
         expected=[[(4 + 8 + 1) / 3, n, (14 + 8) / 2, (8 + 14 + 4) / 3, (4 + 8 + 3) / 3],
                   [(8 + 4 + 1 + 2 + 17) / 5, (8 + 1 + 2 + 17 + 14) / 5, n, n, (8 + 4 + 3 + 4 + 4 + 8) / 6.0],
                   [(2 + 8 + 1 + 17) / 4, (8 + 1 + 2 + 17 + 14 + 4 + 14) / 7, (17 + 14 + 4 + 14 + 8) / 5, n, (4 + 3 + 4 + 8) / 4],
                   [n, (4 + 2 + 17 + 14 + 14) / 5, (14 + 17 + 14 + 4 + 8) / 5, (14 + 8 + 4) / 3, (4 + 4 + 8) / 3]]),
    dict(cite="test_aggregation.py:392-483 test_compute_cbca_with_offset (window 3)", left=_AL4, right=_AR4, msk_left=None,
         msk_right=None, win=3, subpix=1, disp_index=None,
         expected=[[[n, n, n]] * 5,
                   [[n, n, n], [n, (66.0 + 63 + 66 + 63) / 4, 0.0], [55.0, (66 + 63 + 52 + 66 + 63 + 52) / 6, 0.0],
                    [55.0, (63 + 63 + 52 + 52) / 4, n], [n, n, n]],
                   [[n, n, n], [n, (66.0 + 63 + 66 + 63) / 4, 0.0], [55.0, (66 + 63 + 52 + 66 + 63 + 52) / 6, 0.0],
                    [55.0, (63 + 63 + 52 + 52) / 4, n], [n, n, n]],
                   [[n, n, n]] * 5]),
]

# ---- refinement (tests/test_refinement.py) ---------------------------------------------------------------------
_RCV = [[[39, 32.5, 28, 34.5, 41], [49, 41.5, 37, 34, 35.5], [42.5, 40, 45, 40.5, 41], [22, 30, 45, 50, 31]]]
_RCV_NAN = [[[39, 32.5, 28, 34.5, 41], [49, 41.5, n, 34, 35.5], [42.5, 40, n, 40.5, 41], [22, 30, 45, 50, 31]]]
_STOP_R = 1 << 3
_x0, _x1, _x2 = -((34.5 - 32.5) / (2 * (32.5 + 34.5 - 2 * 28))), -((35.5 - 37) / (2 * (37 + 35.5 - 2 * 34))), -((45 - 42.5) / (2 * (42.5 + 45 - 2 * 40)))
_Q = [((32.5 + 34.5 - 2 * 28) / 2) * _x0 * _x0 + ((34.5 - 32.5) / 2) * _x0 + 28,
      ((37 + 35.5 - 2 * 34) / 2) * _x1 * _x1 + ((35.5 - 37) / 2) * _x1 + 34,
      ((42.5 + 45 - 2 * 40) / 2) * _x2 * _x2 + ((45 - 42.5) / 2) * _x2 + 40]
_v0, _v1, _v2 = (32.5 - 34.5) / (2 * (34.5 - 28)), (37 - 35.5) / (2 * (37 - 34)), (42.5 - 45) / (2 * (45 - 40))
_V = [34.5 + (_v0 - 1) * (34.5 - 28), 35.5 + (_v1 - 1) * (37 - 34), 45 + (_v2 - 1) * (45 - 40)]
REFINEMENT = [
    dict(cite="test_refinement.py:87-140 test_quadratic", method="quadratic", cv=_RCV, d_min=-2, d_max=2, subpix=1, disp=[[0, 1, -1, -2]],
         out_disp=[[0 + _x0, 1 + _x1, -1 + _x2, -2]], itp=[[_Q[0], _Q[1], _Q[2], 22]], mask=[[0, 0, 0, _STOP_R]]),
    dict(cite="test_refinement.py:142-225 test_quadratic_subpix", method="quadratic", cv=_RCV, d_min=-1, d_max=1, subpix=2,
         disp=[[0, 0.5, -0.5, -1]], out_disp=[[0 + _x0 / 2, 0.5 + _x1 / 2, -0.5 + _x2 / 2, -1]], itp=[[_Q[0], _Q[1], _Q[2], 22]],
         mask=[[0, 0, 0, _STOP_R]]),
    dict(cite="test_refinement.py:227-318 test_quadratic_with_nan_and_subpix", method="quadratic", cv=_RCV_NAN, d_min=-1, d_max=1, subpix=2,
         disp=[[0, 0.5, -0.5, -1]], out_disp=[[0 + _x0 / 2, 0.5, -0.5, -1]], itp=[[_Q[0], 34, 40, 22]],
         mask=[[0, _STOP_R, _STOP_R, _STOP_R]]),
    dict(cite="test_refinement.py:320-367 test_vfit", method="vfit", cv=_RCV, d_min=-2, d_max=2, subpix=1, disp=[[0, 1, -1, -2]],
         out_disp=[[0 + _v0, 1 + _v1, -1 + _v2, -2]], itp=[[_V[0], _V[1], _V[2], 22]], mask=[[0, 0, 0, _STOP_R]]),
    dict(cite="test_refinement.py:369-446 test_vfit_subpix", method="vfit", cv=_RCV, d_min=-1, d_max=1, subpix=2, disp=[[0, 0.5, -0.5, -1]],
         out_disp=[[0 + _v0 / 2, 0.5 + _v1 / 2, -0.5 + _v2 / 2, -1]], itp=[[_V[0], _V[1], _V[2], 22]], mask=[[0, 0, 0, _STOP_R]]),
    dict(cite="test_refinement.py:514-566 test_vfit_with_nan", method="vfit", cv=[[[n, n, n], [n, 2, 4], [3, 1, 4]]], d_min=-1, d_max=1,
         subpix=1, disp=[[0, 0, 0]], out_disp=[[0, 0, 0 + ((3 - 4) / (2 * (4 - 1)))]],
         itp=[[n, 2, 4 + (((3 - 4) / (2 * (4 - 1))) - 1) * (4 - 1)]], mask=[[0, _STOP_R, 0]]),
    dict(cite="test_refinement.py:568-655 test_vfit_with_nan_and_subpix", method="vfit", cv=_RCV_NAN, d_min=-1, d_max=1, subpix=2,
         disp=[[0, 0.5, -0.5, -1]], out_disp=[[0 + _v0 / 2, 0.5, -0.5, -1]], itp=[[_V[0], 34, 40, 22]],
         mask=[[0, _STOP_R, _STOP_R, _STOP_R]]),
]


# ---- more WTA vectors (tests/test_disparity.py; images of its setUp = WTA["left"] / WTA["right"]) --------------------
_B = -99
WTA_MORE = [
    dict(cite="test_disparity.py:255-292 test_to_disp_with_offset [-3, 1]", method="sad", win=3, subpix=1, dmin=-3, dmax=1, masked=True,
         is_max=False, invalid=-99, disp=[[_B, _B, _B, _B], [_B, 1, 0, _B], [_B, _B, _B, _B]]),
    dict(cite="test_disparity.py:294-323 test_to_disp_with_offset [-3, -1]", method="sad", win=3, subpix=1, dmin=-3, dmax=-1, masked=True,
         is_max=False, invalid=-99, disp=[[_B, _B, _B, _B], [_B, _B, -1, _B], [_B, _B, _B, _B]]),
    dict(cite="test_disparity.py:325-365 test_to_disp_with_offset [1, 3]", method="sad", win=3, subpix=1, dmin=1, dmax=3, masked=True,
         is_max=False, invalid=-99, disp=[[_B, _B, _B, _B], [_B, 1, _B, _B], [_B, _B, _B, _B]]),
    dict(cite="test_disparity.py:372-399 test_argmin_split (sub-pixel volume, NaN as +inf)", method="sad", win=1, subpix=2, dmin=-3, dmax=1,
         masked=False, is_max=False, invalid=0, disp=[[1.0, 1.0, 1.0, -3.0], [1.0, -0.5, 1.0, -3.0], [1.0, 1.0, -1.5, -3]]),
    dict(cite="test_disparity.py:401-430 test_argmax_split (zncc, NaN as -inf, first maximum)", method="zncc", win=1, subpix=2, dmin=-3,
         dmax=1, masked=False, is_max=True, invalid=0, disp=[[0.0, -1.0, -2.0, -3.0]] * 3),
]


# ---- cost-volume confidence: risk and interval bounds (tests/test_confidence/) ---------------------------------------
# variable-disparity volume of tests/test_confidence/conftest.py:92-113 ([disp][row][col] there, [row][col][disp] here)
_CVV = [[[n, 5, n], [1, n, 2], [3, n, 4], [2, n, 5]],
        [[4, 6.2, n], [1, n, 5], [1, n, 0], [1, n, 1]],
        [[n, 0, 0], [n, n, 0], [n, 0, 2], [n, 0, n]],
        [[n, 5, n], [1, n, 2], [3, n, 4], [2, n, 5]]]
_GRIDS_VAR = [[[-1, 0, -1, 0], [0, -1, 0, -1], [0, 0, 0, -1], [-1, -1, -1, -1]],
              [[1, 1, 1, 1], [1, 0, 1, 1], [1, 1, 1, 0], [0, 0, 0, 1]]]  # conftest.py:76-84
RISK = [
    # risk_min of these tests is computed from hand-written sampled ambiguities; the maps that do not depend on them:
    {"cite": "test_risk.py:32-160 test_compute_risk", "cv": [[[39, 28.03, 28, 34.5], [49, 34, 41.5, 34.1], [n, n, n, n]]],
     "disp_range": [-1, 0, 1, 2], "grid_min": [[-1, -1, -1]], "grid_max": [[1, 1, 1]], "etas": [0.0, 0.3],
     "risk_max": [[0.5, 1.0, n]], "disp_sup": [[1.0, 1.0, n]], "disp_inf": [[0.5, 0.0, n]]},
    {"cite": "test_risk.py:270-319 test_compute_risk_with_variable_disparity", "cv": _CVV, "disp_range": [-1, 0, 1],
     "grid_min": _GRIDS_VAR[0], "grid_max": _GRIDS_VAR[1], "etas": [0.0, 0.3],
     "risk_max": [[2.0, 1.5, 1.5, 1.0], [2.0, 1.0, 1.5, 2.0], [1.0, 1.0, 0.0, 1.0], [1.0, 1.5, 1.5, 1.0]],
     "disp_sup": [[1.0, 0.5, 0.5, 0.0], [1.0, 0.0, 1.0, 1.0], [1.0, 1.0, 0.0, 0.0], [0.0, 0.5, 0.5, 0.0]],
     "disp_inf": [[-1.0, -1.0, -1.0, -1.0], [-1.0, -1.0, -0.5, -1.0], [0.0, 0.0, 0.0, -1.0], [-1.0, -1.0, -1.0, -1.0]]},
]

# test_interval_bounds.py:30-116: SAD window 1 on the confidence pair (conftest.py:34-89, range [-1, 1], left mask on
# (1,1) and (3,3)), possibility threshold 0.7, "min" measure; the cost volume is the one the test's comment spells out
CONFIDENCE_LEFT = [[2, 5, 3, 1], [5, 3, 2, 1], [4, 2, 3, 2], [4, 5, 3, 2]]
CONFIDENCE_RIGHT = [[1, 2, 1, 2], [2, 3, 5, 3], [0, 2, 4, 2], [5, 3, 1, 4]]
CONFIDENCE_LEFT_MASK = [[0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1]]
INTERVAL_BOUNDS = {
    "cite": "test_interval_bounds.py:30-116", "threshold": 0.7, "type_factor": -1.0, "disp_range": [-1, 0, 1],
    "cv": [[[n, 1, 0], [4, 3, 4], [1, 2, 1], [0, 1, n]], [[n, 3, 2], [n, n, n], [1, 3, 1], [4, 2, n]],
           [[n, 4, 2], [2, 0, 2], [1, 1, 1], [2, 0, n]], [[n, 1, 1], [0, 2, 4], [0, 2, 1], [n, n, n]]],
    "inf": [[0, -1, -1, -1], [0, n, -1, -1], [0, -1, -1, -1], [-1, -1, -1, n]],
    "sup": [[1, 1, 1, 0], [1, n, 1, 1], [1, 1, 1, 1], [1, 0, 1, n]],
}


# ---- median_for_intervals filter (tests/test_filter.py:662-800) -----------------------------------------------------------
_IREG = 1 << 11  # PANDORA_MSK_PIXEL_INTERVAL_REGULARIZED
MEDIAN_FOR_INTERVALS = {
    "inf": [[4, 5, 7, 7, 8], [5, 84, 0, 35, 4], [2, 7, 21, 10, 1], [5, 0, 8, 1, 3]],
    "sup": [[6, 7, 9, 9, 10], [7, 86, 2, 37, 6], [4, 10, 23, 12, 3], [7, 2, 10, 3, 5]],
    "plain": {"cite": "test_filter.py:696-727", "cfg": {"filter_method": "median_for_intervals", "filter_size": 3},
              "inf": [[4, 5, 7, 7, 8], [5, 5, 7, 7, 4], [2, 5, 8, 4, 1], [5, 0, 8, 1, 3]],
              "sup": [[6, 7, 9, 9, 10], [7, 7, 10, 9, 6], [4, 7, 10, 6, 3], [7, 2, 10, 3, 5]]},
    "regularized": {"cite": "test_filter.py:729-800",
                    "cfg": {"filter_method": "median_for_intervals", "filter_size": 3, "regularization": True, "ambiguity_kernel_size": 3,
                            "ambiguity_threshold": 0.8, "vertical_depth": 2, "quantile_regularization": 0.8},
                    "ambiguity": [[1.0, 0.7, 1.0, 1.0, 1.0], [0.7, 1.0, 1.0, 1.0, 1.0], [1.0, 1.0, 1.0, 1.0, 0.7], [1.0, 1.0, 1.0, 0.7, 1.0]],
                    "inf": [[4.8, 4.8, 4.8, 7, 8], [4.8, 4.8, 7, 7, 4], [2, 5, 8, 2.2, 1], [5, 0, 2.2, 2.2, 3]],
                    "sup": [[7.4, 7.4, 7.4, 9, 10], [7.4, 7.4, 10, 9, 6], [4, 7, 10, 8.4, 3], [7, 2, 8.4, 8.4, 5]],
                    "validity": [[_IREG, _IREG, _IREG, 0, 0], [_IREG, _IREG, 0, 0, 0], [0, 0, 0, _IREG, 0], [0, 0, _IREG, _IREG, 0]]},
}
