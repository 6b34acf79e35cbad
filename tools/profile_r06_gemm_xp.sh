#!/bin/bash
# L2 / LDS counters of the headline step's launches, eager (every kernel its own dispatch):  gpurun -- 'bash tools/profile_r06_gemm_xp.sh'
# -> gpurun_out/r06p/pmc_l2/ , pmc_ldsx/ (csv); summarised by tools/pmc_summary_kernel.py into profiles/r06_pmc_gemm_xp.md
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r06p
mkdir -p $O
B="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --repeat 1 --minibatch-steps 0 --steps 3 --warmup 1"
export NEMO_GRAPHS=0
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_l2 -o l -- python3 bench.py $B > $O/pmc_l2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_ldsx -o l -- python3 bench.py $B > $O/pmc_ldsx.log 2>&1
find $O -name "*.db" -delete
ls -la $O/pmc_l2 $O/pmc_ldsx | head -20; tail -3 $O/pmc_l2.log | cut -c1-300
