cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fam8.py::test_short_wide_images_take_the_families_by_default tests/test_gpu_parity.py::test_float_sgm_mid_size_takes_the_marching_schedule_by_default -q -x 2>&1 | tail -8
