#!/usr/bin/env python3
"""Generates tests/golden/cv_masked_cases.json from the reference's own parametrised tests of cv_masked
(tests/test_matching_cost/test_matching_cost.py: TestCvMasked, TestCvMaskedWithGrid and their sub-pixel variants):
images, masks, mask conventions, disparity ranges or grids, matching-cost methods, expected NaN masks = data only.

The reference package cannot be imported here (xarray / rasterio / json_checker are absent), so the test MODULE is loaded
with those imports stubbed out; its image fixtures are evaluated with `make_image` replaced by a recorder, and only the
literal arrays of the @pytest.mark.parametrize decorators are read.
Run in the build container:  python tests/golden/gen_cv_masked_golden.py /root/reference
"""
import importlib.util
import inspect
import json
import os
import sys
from unittest import mock

import numpy as np


class Image:
    def __init__(self, data, disparity):
        self.data, self.disparity = np.asarray(data), disparity
        self.attrs = {"valid_pixels": 0, "no_data_mask": 1}  # tests/common.py img_attrs


def fixtures_of(owner):
    out = {}
    for name, attr in vars(owner).items():
        fn = getattr(attr, "_get_wrapped_function", None)
        if fn is not None:
            out[name] = (fn(), attr)
    return out


def resolve(name, fixtures, instance, cache):
    if name not in cache:
        fn, _ = fixtures[name]
        args = [a for a in inspect.signature(fn).parameters if a != "self"]
        vals = [resolve(a, fixtures, instance, cache) for a in args]
        cache[name] = fn(instance, *vals) if "self" in inspect.signature(fn).parameters else fn(*vals)
    return cache[name]


def main(ref_root):
    for name in ["xarray", "rasterio", "rasterio.io", "rasterio.windows", "json_checker", "json_checker.core", "json_checker.core.exceptions",
                 "transitions", "skimage", "skimage.transform", "pandora", "pandora.img_tools", "pandora.criteria", "pandora.constants",
                 "pandora.matching_cost", "pandora.margins", "pandora.margins.descriptors", "tests", "tests.common"]:
        sys.modules.setdefault(name, mock.MagicMock())
    path = os.path.join(ref_root, "tests", "test_matching_cost", "test_matching_cost.py")
    spec = importlib.util.spec_from_file_location("ref_test_matching_cost", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.make_image = Image
    mod.xr.DataArray = lambda arr, dims=None: np.asarray(arr)
    default_methods = list(vars(mod)["matching_cost_method"]._fixture_function_marker.params)
    cases = []
    for cname, cls in vars(mod).items():
        if not inspect.isclass(cls) or not cname.startswith("Test"):
            continue
        if "make_image" in vars(cls):
            cls.make_image = staticmethod(Image)
        fixtures = fixtures_of(cls)
        methods = default_methods
        if "matching_cost_method" in fixtures:
            methods = list(fixtures["matching_cost_method"][1]._fixture_function_marker.params)
        for tname, fn in vars(cls).items():
            for mark in getattr(fn, "pytestmark", []):
                if mark.name != "parametrize" or "make_cv_masked_parameters" not in mark.args[0]:
                    continue
                for p in mark.args[1]:
                    param, expected = p.values
                    inst, cache = cls(), {}
                    left = resolve(param["left_image"], fixtures, inst, cache)
                    right = resolve(param["right_image"], fixtures, inst, cache)
                    disp = left.disparity
                    case = {"id": f"{cname}::{tname}[{p.id}]", "methods": methods, "window_size": int(param["cfg"]["window_size"]),
                            "subpix": int(param["cfg"]["subpix"]), "left": left.data.tolist(), "right": right.data.tolist(),
                            "valid_pixels": left.attrs["valid_pixels"], "no_data_mask": left.attrs["no_data_mask"],
                            "left_mask": None if param.get("left_mask") is None else np.asarray(param["left_mask"]).tolist(),
                            "right_mask": None if param.get("right_mask") is None else np.asarray(param["right_mask"]).tolist(),
                            "expected_nan_mask": np.asarray(expected).astype(int).tolist()}
                    assert right.attrs == left.attrs
                    if isinstance(disp, (list, tuple)):
                        case["disparity"] = [int(disp[0]), int(disp[1])]
                    else:
                        case["disparity_grids"] = np.asarray(disp).astype(int).tolist()
                    extra = set(param["cfg"]) - {"window_size", "subpix"}
                    assert not extra, extra
                    cases.append(case)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cv_masked_cases.json")
    with open(out, "w") as f:
        json.dump({"source": "tests/test_matching_cost/test_matching_cost.py (reference): parametrize literals + image fixtures of the "
                             "cv_masked test classes", "cases": cases}, f)
    print(f"wrote {len(cases)} cases to {out}")
    for c in cases:
        print(" ", c["id"], c["methods"], c["window_size"], c["subpix"], np.shape(c["expected_nan_mask"]),
              "grid" if "disparity_grids" in c else c["disparity"])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
