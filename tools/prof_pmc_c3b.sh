#!/bin/bash
# MFMA-pipe counters of the C3 bf16 step (eager launches: every kernel its own dispatch):  gpurun -- 'bash tools/prof_pmc_c3b.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r03p_c3b
rm -rf $O; mkdir -p $O
export NEMO_GRAPHS=0
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --instances 40 --dtype bf16 --steps 4 --warmup 1 --repeat 1 --minibatch-steps 0 --no-cpu-baseline --no-torch-gpu-baseline > $O/pmc_m.log 2>&1
python3 tools/pmc_mfma_summary.py $O > $O/pmc_mfma.md 2>&1
head -14 $O/pmc_mfma.md | cut -c1-200
