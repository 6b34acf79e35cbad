#!/bin/bash
# kernel traces of C3 bf16 with operands in memory and on the fly
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r03c3b
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-torch-gpu-baseline --repeat 1 --minibatch-steps 0"
for m in 1 0; do
  export NEMO_BF16_MEM=$m
  rocprofv3 --kernel-trace --stats -d $O/trace_m$m -o t -- python3 bench.py --instances 40 --dtype bf16 --steps 10 --warmup 2 $B > $O/trace_m$m.log 2>&1
  python3 tools/prof_summary.py $O/trace_m$m/t_results.db 25 > $O/summary_m$m.md 2>&1
done
find $O -name "*.db" -size +30M -delete
