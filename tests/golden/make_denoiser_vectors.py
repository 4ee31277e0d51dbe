#!/usr/bin/env python3
"""Golden vectors of the disparity_denoiser filter, taken from the LITERAL arrays of the reference's own tests
(/root/reference/tests/test_disparity_denoiser.py: test_get_grad, test_with_valid_pixel_multiband_and_monoband,
test_with_invalid_center).  The reference module itself cannot be imported here (xarray is not installed), so the expected map of
the end-to-end case is computed the way that test computes it: from its hand-written intermediate arrays (euclidian / colour /
planar distances) with the three gaussians and the normalised weighted sum.  Run in the build container (reads /root/reference);
writes tests/golden/disparity_denoiser.json, which is what the tests read."""
import ast
import json
import os

import numpy as np

SRC = "/root/reference/tests/test_disparity_denoiser.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "disparity_denoiser.json")


def assignments(func, names):
    """evaluate, in order, the top-level `name = <expr>` statements of a test method for the wanted names"""
    env = {"np": np}
    for node in func.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
            if name in names:
                env[name] = eval(compile(ast.Expression(node.value), SRC, "eval"), env)  # literal arrays / numpy arithmetic only
    return env


tree = ast.parse(open(SRC).read())
methods = {f.name: f for c in tree.body if isinstance(c, ast.ClassDef) for f in c.body if isinstance(f, ast.FunctionDef)}

g = assignments(methods["test_get_grad"], {"disp", "gt_y", "gt_x"})
e = assignments(methods["test_with_valid_pixel_multiband_and_monoband"],
                {"disp", "win_coords", "gt_euclidian_dist", "clr_dist", "planar_dist_centered", "planar_dist", "gt_weights"})
w = e["gt_weights"] / np.sum(e["gt_weights"], axis=(-2, -1), keepdims=True)  # DisparityDenoiser.bilateral_filter
expected = e["disp"] + np.sum(e["planar_dist"] * w, axis=(-2, -1)).squeeze()
c = assignments(methods["test_with_invalid_center"], {"disp", "data"})

vectors = {
    "source": "literal arrays of /root/reference/tests/test_disparity_denoiser.py (see make_denoiser_vectors.py)",
    "get_grad": {"sigma_grad": 0.0, "disp": g["disp"].tolist(), "grad_row": g["gt_y"].tolist(), "grad_col": g["gt_x"].tolist()},
    "end_to_end": {"cfg": {"filter_size": 3, "sigma_euclidian": 4.0, "sigma_color": 100.0, "sigma_planar": 12.0},
                   "disp": e["disp"].tolist(), "band": [[1, 1], [1, 3]], "expected": expected.tolist(),
                   "planar_dist": np.asarray(e["planar_dist"]).tolist(),
                   "planar_dist_centered": np.asarray(e["planar_dist_centered"]).tolist()},
    "invalid_center": {"disp": [[2, 4, 8, 5, 6], [7, 82, 3, 33, 4], [4, 8, 21, 13, 4], [3, 2, 8, 1, 3], [3, 6, 2, 3, 2]],
                       "band_green": [[2, 3, 4, 6, 8], [8, 7, 0, 4, 7], [4, 9, 1, 5, 1], [6, 5, 2, 1, 4], [1, 5, 4, 3, 2]],
                       "invalid_at": [2, 2]},
}
with open(OUT, "w") as f:
    json.dump(vectors, f, indent=1)
print(OUT, "expected =", expected.tolist())
