#!/bin/bash
# kernel trace of the C2 bf16 step with the split-precision mesh kernel off / on:  gpurun -- 'bash tools/prof_ab_mesh.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r03s
mkdir -p $O
B="--no-cpu-baseline --no-torch-gpu-baseline --steps 10 --warmup 2 --repeat 1 --minibatch-steps 0 --dtype bf16"
for m in 0 1; do
  export NEMO_MESH_SPLIT=$m
  rocprofv3 --kernel-trace --stats -d $O/trace_m$m -o t -- python3 bench.py $B > $O/trace_m$m.log 2>&1
  python3 tools/prof_summary.py $O/trace_m$m/t_results.db 12 > $O/summary_m$m.md 2>&1
done
find $O -name "*.db" -size +30M -delete
head -20 $O/summary_m0.md $O/summary_m1.md
