#!/usr/bin/env python3
"""Generates tests/golden/criteria_cases.json from the reference's parametrised tests of the criteria sub-functions
(tests/test_criteria.py: test_binary_dilation_msk, test_mask_border, test_mask_invalid_variable_disparity_range,
test_allocate_right_mask, test_allocate_left_mask): inputs + expected masks = data only.  (test_validity_mask has its own
generator, gen_validity_mask_golden.py.)

The reference package cannot be imported here, so the test MODULE is loaded with its imports stubbed out and only the
literal arrays of the @pytest.mark.parametrize decorators are read.
Run in the build container:  python tests/golden/gen_criteria_golden.py /root/reference
"""
import importlib.util
import json
import os
import sys
from unittest import mock

import numpy as np

WANTED = ("test_binary_dilation_msk", "test_mask_border", "test_mask_invalid_variable_disparity_range", "test_allocate_right_mask",
          "test_allocate_left_mask")


def plain(v):
    if isinstance(v, dict):
        return {k: plain(x) for k, x in v.items() if k in ("valid_pixels", "no_data_mask")}
    if isinstance(v, tuple):  # np.where(...) results
        return [plain(x) for x in v]
    if isinstance(v, np.ndarray):
        if v.dtype.kind == "f":
            return [[None if np.isnan(x) else float(x) for x in row] for row in v] if v.ndim == 2 else \
                [None if np.isnan(x) else float(x) for x in v]
        return v.astype(int).tolist()
    if isinstance(v, (list, int, float, str)) or v is None:
        return v
    return int(v)


def main(ref_root):
    for name in ["xarray", "rasterio", "rasterio.io", "rasterio.windows", "json_checker", "transitions", "skimage", "skimage.transform",
                 "pandora", "pandora.img_tools", "pandora.criteria", "pandora.constants", "pandora.matching_cost",
                 "pandora.disparity", "pandora.margins", "tests", "tests.common"]:
        sys.modules.setdefault(name, mock.MagicMock())
    import pandora_amd.constants as real_cst

    sys.modules["pandora"].constants = real_cst
    sys.modules["pandora.constants"] = real_cst
    path = os.path.join(ref_root, "tests", "test_criteria.py")
    spec = importlib.util.spec_from_file_location("ref_test_criteria2", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for obj in vars(mod).values():
        for tname in WANTED:
            fn = getattr(obj, tname, None)
            if fn is None or tname in out:
                continue
            for mark in getattr(fn, "pytestmark", []):
                if mark.name != "parametrize":
                    continue
                names = [n.strip() for n in mark.args[0].split(",")] if isinstance(mark.args[0], str) else list(mark.args[0])
                out[tname] = [dict({"id": p.id}, **{k: plain(v) for k, v in zip(names, p.values)}) for p in mark.args[1]]
    path_out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "criteria_cases.json")
    with open(path_out, "w") as f:
        json.dump({"source": "tests/test_criteria.py (reference), parametrize literals", "tests": out}, f)
    for k, v in out.items():
        print(k, len(v), sorted(v[0]))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
