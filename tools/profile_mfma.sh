#!/bin/bash
# MFMA-pipe and LDS counters of the step's kernels (separate --pmc passes, kernel trace only):
#   gpurun --timeout 900 -- 'bash tools/profile_mfma.sh'    then    python tools/pmc_mfma_summary.py
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out
rm -rf $O/pmc_mfma $O/pmc_lds
export NEMO_GRAPHS=0
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-torch-gpu-baseline > $O/pmc_m.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_lds -o l -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-torch-gpu-baseline > $O/pmc_l.log 2>&1
find $O/pmc_mfma $O/pmc_lds -name "*counter_collection.csv" | head
tail -c 300 $O/pmc_m.log
