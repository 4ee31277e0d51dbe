"""The context's kernel-route options (include/pandora_amd.h pmx_set_option): set / get / clear, unknown names refused, the
environment read once - when the context is created - and never again, and a forced route really taken."""
import numpy as np
import pytest

from tests.test_gpu_parity import pair

pytestmark = pytest.mark.gpu


def test_options_set_get_clear_and_unknown_names(monkeypatch):
    from pandora_amd.engine import Engine, PmxError

    monkeypatch.setenv("PMX_SGM_SCHED", "seq")  # seeds the option of contexts created from now on
    e = Engine(0)
    try:
        assert "SGM8_FAM" in Engine.option_names() and "CBCA_FAST" in Engine.option_names()
        assert e.get_option("SGM_SCHED") == "seq" and e.get_option("SGM8_FAM") is None
        monkeypatch.setenv("PMX_SGM8_FAM", "1")  # after pmx_create: the library does not look
        assert e.get_option("SGM8_FAM") is None
        e.set_option("PMX_SGM8_FAM", "1")        # (with or without the prefix)
        assert e.get_option("SGM8_FAM") == "1"
        e.set_option("SGM8_FAM", None)
        assert e.get_option("SGM8_FAM") is None
        e.options_from_env()                     # what a script does that changes its environment between steps
        assert e.get_option("SGM8_FAM") == "1"
        with pytest.raises(PmxError):
            e.set_option("NO_SUCH_OPTION", "1")
    finally:
        e.close()


def test_a_forced_route_is_the_one_that_runs(oracle):
    """SGM8_FAM=1 on a small pair: three byte volumes instead of eight (pmx_debug_path_costs has nothing to show), same bits."""
    from pandora_amd.engine import Engine

    L, R = pair(40, 100, seed=11)
    e = Engine(0)
    try:
        e.set_images(L, R, 1)
        ref = oracle.sgm(oracle.census_cost(L, R, 17, -8, 1, 5), 8, 32, False, 26.0, False)
        for forced in (None, "1"):
            e.set_option("SGM8_FAM", forced)
            cv = e.alloc_cv(17, -8)
            e.census(cv, 5)
            e.sgm(cv, 8, 32, False, 26.0, False)
            if forced:
                with pytest.raises(Exception):
                    e.debug_path_costs(cv, raw=True)
            else:
                assert e.debug_path_costs(cv, raw=True)[0].shape[0] == 8
            np.testing.assert_array_equal(cv.to_host(), ref)
            cv.free()
    finally:
        e.close()
