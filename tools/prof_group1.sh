#!/bin/bash
# kernel trace of the SHARDED step in an RCCL group of one (collectives inside the step's graph):  gpurun -- 'bash tools/prof_group1.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r03p
mkdir -p $O
export NEMO_BENCH_SHARD_OF_ONE=1
rocprofv3 --kernel-trace --stats -d $O/trace_g1 -o t -- python3 bench.py --instances 1 --shard-mode split --steps 20 --warmup 2 --repeat 1 --minibatch-steps 0 --no-cpu-baseline --no-torch-gpu-baseline > $O/trace_g1.log 2>&1
python3 tools/prof_summary.py $O/trace_g1/t_results.db 40 > $O/summary_g1.md 2>&1
python3 tools/step_timeline.py $O/trace_g1/t_results.db 12 > $O/timeline_g1.txt 2>&1
find $O -name "*.db" -size +30M -delete
tail -3 $O/trace_g1.log | cut -c1-300
