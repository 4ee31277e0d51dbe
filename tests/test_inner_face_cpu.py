"""CPU: pandora_amd.inner_cpp (pybind11 face over the C ABI) loads, exports the reference's native-function names, refuses to
compute without a GPU (no fallback), and its two scalar callbacks equal the reference's compiled ones."""
import numpy as np
import pytest


def test_face_exports_the_reference_names_and_has_no_cpu_fallback():
    from pandora_amd import inner_cpp as face

    for name in ("compute_matching_costs", "reverse_cost_volume", "reverse_disp_range", "cross_support", "cbca", "loop_refinement", "vfit_refinement_method",
                 "quadratic_refinement_method"):
        assert callable(getattr(face, name)), name
    from pandora_amd import _lib

    if _lib.lib().pmx_device_count() == 0:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            face.cross_support(np.zeros((4, 4), np.float32), 5, 30.0)


@pytest.mark.parametrize("measure", ["min", "max"])
def test_scalar_callbacks_equal_the_compiled_reference(measure):
    from oracle import ref
    from pandora_amd import inner_cpp as face

    rf = ref.load("refinement_cpp")
    if rf is None:
        pytest.skip("oracle/_ref is not built here")
    rng = np.random.default_rng(3)
    cases = [rng.integers(0, 6, 3).astype(np.float32) for _ in range(200)] + [np.array(c, np.float32) for c in
                                                                              ([3, 1, 2], [1, 1, 1], [np.nan, 1, 2], [2, 1, np.nan], [2, 5, 3], [0, 0, 1])]
    for c in cases:
        for name in ("vfit_refinement_method", "quadratic_refinement_method"):
            got, exp = getattr(face, name)(c, 0.0, measure, 8), getattr(rf, name)(c, 0.0, measure, 8)
            np.testing.assert_array_equal(np.array(got, np.float64), np.array(exp, np.float64))
