#!/bin/bash
# kernel trace of the headline step (graph replays) + summary + one step in time order:  gpurun -- 'bash tools/profile_r06_trace.sh [tag] [bench args]'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
TAG=${1:-c2}; shift || true
O=gpurun_out/r06p
mkdir -p $O; rm -rf $O/trace_$TAG
B="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --repeat 1 --minibatch-steps 0"
rocprofv3 --kernel-trace --stats -d $O/trace_$TAG -o t -- python3 bench.py --steps 20 --warmup 2 $B "$@" > $O/trace_$TAG.log 2>&1
python3 tools/prof_summary.py $O/trace_$TAG/t_results.db 35 > $O/summary_$TAG.md 2>&1
python3 tools/step_timeline.py $O/trace_$TAG/t_results.db 12 > $O/timeline_$TAG.txt 2>&1
find $O -name "*.db" -delete
cat $O/summary_$TAG.md | cut -c1-160 | head -40; cat $O/timeline_$TAG.txt | cut -c1-130
