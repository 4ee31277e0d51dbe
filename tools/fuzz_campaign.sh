#!/bin/bash
# a long fuzz campaign over every fuzzer with fresh seeds; each leg bounded
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/camp
FUZZ_FROM=300000 FUZZ_TO=310000 timeout 420 python tools/fuzz_machine.py 2>&1 | tail -6 > gpurun_out/camp/machine.log
FUZZ_FROM=300000 FUZZ_TO=305000 timeout 420 python tools/fuzz_validation.py 2>&1 | tail -6 > gpurun_out/camp/validation.log
FUZZ_FROM=300000 FUZZ_TO=302000 timeout 420 python tools/fuzz_mid.py 2>&1 | tail -6 > gpurun_out/camp/mid.log
FUZZ_FROM=300000 FUZZ_TO=303000 timeout 300 python tools/fuzz_filters.py 2>&1 | tail -6 > gpurun_out/camp/filters.log
FUZZ_FROM=300000 FUZZ_TO=302000 timeout 300 python tools/fuzz_confidence.py 2>&1 | tail -6 > gpurun_out/camp/confidence.log
FUZZ_FROM=300000 FUZZ_TO=306000 timeout 420 python tools/fuzz_more.py 2>&1 | tail -6 > gpurun_out/camp/more.log
