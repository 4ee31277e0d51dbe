#!/bin/bash
# Full-size parity tests (tests/test_gpu_full_size.py: the integer path against the float32 kernels at 4096^2 x 257, the local steps against
# the oracle on the image's bottom rows) once per forced kernel route: the routes the size rules do not pick at this size - fallbacks
# that otherwise only run under memory pressure or on other shapes (round 6: CBCA's in-place passes were wrong at this size).
# Usage (GPU box): bash tools/route_sweep_full_size.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run () { echo "== $*"; env "$@" timeout 900 python -m pytest tests/test_gpu_full_size.py -q -p no:cacheprovider -k "(agree_at_full_size and (C4 or C3)) or bottom_strip" 2>&1 | grep -E "passed|failed|^FAILED" | tail -4; }
run PMX_SGM8_FAM=0
run PMX_SGM8_FAM=1
run PMX_SGM8_HPAIR=2
run PMX_SGM8_HPAIR=3
run PMX_SGM8_CODES=1
run PMX_SGM8_FAMCODES=1
run PMX_COST5=0
run PMX_WTA3=0
run PMX_SGM8=0
run PMX_SGM8_HF=0
run PMX_SGM8_OVERLAP=0
run PMX_SGM8_FAM_NW=4
run PMX_SGM_SCHED=seq
run PMX_SGM_SCHED=par
run PMX_SGM_HFUSED=0
run PMX_SGM_PENDING=0
run PMX_SGM_FAM_PAR=1
run PMX_SGM_FAM_SHAPE=32,9,4
run PMX_SGM_FAM_SHAPE=32,9,10
run PMX_CBCA_MARCH=0
run PMX_CBCA_FAST=0
run PMX_CBCA_FAST=2
run PMX_CBCA_FAST=4
run PMX_CBCA_VBUF=0
run PMX_CBCA_SIGN=0
run PMX_CBCA_FUSE=0
run PMX_CBCA_ARMS_FLAT=0
