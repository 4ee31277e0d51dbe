#!/bin/bash
# a long fuzz campaign over every fuzzer with fresh seeds; each leg bounded.  FUZZ_BASE=<n> moves the seed ranges (default 400000)
cd $GRAFT_REPO_ROOT
B=${FUZZ_BASE:-400000}
mkdir -p gpurun_out/camp
FUZZ_FROM=$B FUZZ_TO=$((B + 10000)) timeout 420 python tools/fuzz_machine.py 2>&1 | tail -6 > gpurun_out/camp/machine.log
FUZZ_FROM=$B FUZZ_TO=$((B + 5000)) timeout 420 python tools/fuzz_validation.py 2>&1 | tail -6 > gpurun_out/camp/validation.log
FUZZ_FROM=$B FUZZ_TO=$((B + 2000)) timeout 420 python tools/fuzz_mid.py 2>&1 | tail -6 > gpurun_out/camp/mid.log
FUZZ_FROM=$B FUZZ_TO=$((B + 3000)) timeout 300 python tools/fuzz_filters.py 2>&1 | tail -6 > gpurun_out/camp/filters.log
FUZZ_FROM=$B FUZZ_TO=$((B + 2000)) timeout 300 python tools/fuzz_confidence.py 2>&1 | tail -6 > gpurun_out/camp/confidence.log
FUZZ_FROM=$B FUZZ_TO=$((B + 6000)) timeout 420 python tools/fuzz_more.py 2>&1 | tail -6 > gpurun_out/camp/more.log
