#!/bin/bash
# Kernel traces (graph replays) of the shard-size steps and the headline step:
#   gpurun --timeout 900 -- 'bash tools/profile_small.sh r02c'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
TAG=${1:-r02}
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
for v in 1 8; do
  rocprofv3 --kernel-trace --stats -d $O/trace_v$v -o t -- python3 bench.py --instances $v --steps 20 --warmup 2 --no-cpu-baseline --no-torch-gpu-baseline > $O/trace_v$v.log 2>&1
  python3 tools/prof_summary.py $O/trace_v$v/t_results.db 35 > $O/summary_v$v.md 2>&1
  python3 tools/step_timeline.py $O/trace_v$v/t_results.db 15 > $O/timeline_v$v.txt 2>&1
done
find $O -name "*.db" -size +40M -delete
ls -la $O $O/trace_v1 | head -30
