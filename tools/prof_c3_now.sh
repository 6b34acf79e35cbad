#!/bin/bash
# kernel trace + one-step timeline of the C3 bf16 step as built:  gpurun -- 'bash tools/prof_c3_now.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=${OUT:-gpurun_out/r05/c3n}
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-torch-gpu-baseline --repeat 1 --minibatch-steps 0"
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 bench.py --instances 40 --dtype bf16 --steps 10 --warmup 2 $B > $O/trace.log 2>&1
python3 tools/prof_summary.py $O/trace/t_results.db 25 > $O/summary.md 2>&1
python3 tools/step_timeline.py $O/trace/t_results.db 12 > $O/timeline.txt 2>&1
find $O -name "*.db" -size +30M -delete
