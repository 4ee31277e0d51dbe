#!/usr/bin/env python3
"""Generates tests/golden/census_cases.json from the reference's own parametrised test
tests/test_matching_cost/test_matching_cost_census.py::test_census (inputs + expected cost layers = data only).

The reference package cannot be imported here (xarray / rasterio / json_checker are absent), so the test MODULE is loaded
with those imports stubbed out and only the literal arrays of its @pytest.mark.parametrize decorator are read.
Run in the build container:  python tests/golden/gen_census_golden.py /root/reference
"""
import importlib.util
import json
import os
import sys
from unittest import mock

import numpy as np


def _nan_to_none(a):
    a = np.asarray(a, np.float64)
    return [[None if np.isnan(v) else float(v) for v in row] for row in a] if a.ndim == 2 else \
        [[[None if np.isnan(v) else float(v) for v in col] for col in row] for row in a]


def main(ref_root):
    for name in ["xarray", "rasterio", "rasterio.io", "rasterio.windows", "json_checker", "transitions", "skimage", "skimage.transform",
                 "pandora", "pandora.img_tools", "pandora.criteria", "pandora.constants", "pandora.matching_cost",
                 "pandora.margins", "pandora.margins.descriptors", "tests", "tests.common"]:
        sys.modules.setdefault(name, mock.MagicMock())
    path = os.path.join(ref_root, "tests", "test_matching_cost", "test_matching_cost_census.py")
    spec = importlib.util.spec_from_file_location("ref_test_census", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases = []
    for mark in getattr(mod.test_census, "pytestmark", []):
        if mark.name != "parametrize":
            continue
        names = [n.strip() for n in mark.args[0].split(",")]
        for p in mark.args[1]:
            v = dict(zip(names, p.values))
            cases.append({"id": p.id, "left": np.asarray(v["left_data"]).tolist(), "right": np.asarray(v["right_data"]).tolist(),
                          "expected": _nan_to_none(v["ref_out"]), "window_size": int(v["window_size"]), "subpix": int(v["subpix"]),
                          "disp_interval": [int(x) for x in v["disp_interval"]],
                          "tested_layer": v["tested_layer"] if isinstance(v["tested_layer"], str) else float(v["tested_layer"])})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "census_cases.json")
    with open(out, "w") as f:
        json.dump({"source": "tests/test_matching_cost/test_matching_cost_census.py::test_census (reference), parametrize literals",
                   "cases": cases}, f)
    print(f"wrote {len(cases)} cases to {out}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
