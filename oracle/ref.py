"""Loader for oracle/_ref: the reference's own pybind11 C++ modules compiled by oracle/Makefile from
/root/reference (sources are never copied).  TEST INFRASTRUCTURE ONLY."""
import importlib.util
import os
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    """name in {'matching_cost_cpp','aggregation_cpp','refinement_cpp','cost_volume_confidence_cpp','img_tools_cpp','validation_cpp','interval_tools_cpp'}; None when not built."""
    path = os.path.join(_HERE, "_ref", name + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
