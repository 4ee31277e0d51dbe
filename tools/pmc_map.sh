#!/bin/bash
# instruction mix of the fused kernel for a forced lane map: bash tools/pmc_map.sh 8x17
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcm
export PMX_FUSED_MAP=$1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY -d gpurun_out/pmcm -o m_$1 -- python bench.py --steps 2 --warmup 1 --cpu-rows 0 > gpurun_out/pmcm/log_$1.txt 2>&1
python tools/rocpd_pmc.py gpurun_out/pmcm/m_$1*.db | grep fused | rev | cut -d, -f2-4 | rev
rm -f gpurun_out/pmcm/*.db
