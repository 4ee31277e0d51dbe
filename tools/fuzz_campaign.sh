#!/bin/bash
# a long fuzz campaign over every fuzzer with fresh seeds; each leg bounded
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/camp
FUZZ_FROM=400000 FUZZ_TO=410000 timeout 420 python tools/fuzz_machine.py 2>&1 | tail -6 > gpurun_out/camp/machine.log
FUZZ_FROM=400000 FUZZ_TO=405000 timeout 420 python tools/fuzz_validation.py 2>&1 | tail -6 > gpurun_out/camp/validation.log
FUZZ_FROM=400000 FUZZ_TO=402000 timeout 420 python tools/fuzz_mid.py 2>&1 | tail -6 > gpurun_out/camp/mid.log
FUZZ_FROM=400000 FUZZ_TO=403000 timeout 300 python tools/fuzz_filters.py 2>&1 | tail -6 > gpurun_out/camp/filters.log
FUZZ_FROM=400000 FUZZ_TO=402000 timeout 300 python tools/fuzz_confidence.py 2>&1 | tail -6 > gpurun_out/camp/confidence.log
FUZZ_FROM=400000 FUZZ_TO=406000 timeout 420 python tools/fuzz_more.py 2>&1 | tail -6 > gpurun_out/camp/more.log
