#!/bin/bash
# HBM-side traffic of the C4 leg (256 x 1024, fp32 defaults of round 6), separate FETCH_SIZE / WRITE_SIZE passes, eager launches:
#   gpurun --timeout 1500 -- 'bash tools/profile_r06_traffic_c4.sh'   then (here)   python3 tools/write_profiles_r06_c4.py
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r06p
mkdir -p $O
B="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --repeat 1 --minibatch-steps 0"
export NEMO_GRAPHS=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_c4 -o f -- python3 bench.py --instances 256 --frames 1024 --steps 2 --warmup 1 $B > $O/pmc_f_c4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_c4 -o w -- python3 bench.py --instances 256 --frames 1024 --steps 2 --warmup 1 $B > $O/pmc_w_c4.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch_c4/f_results.db $O/pmc_write_c4/w_results.db > $O/pmc_traffic_c4.md 2>&1
find $O -name "*.db" -delete
tail -n 3 $O/pmc_traffic_c4.md
