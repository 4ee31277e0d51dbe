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
