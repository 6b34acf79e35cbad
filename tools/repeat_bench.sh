#!/bin/bash
# ms per step of `n` separate processes of the same bench configuration (graph instantiation differs from process to process):
#   gpurun -- 'bash tools/repeat_bench.sh 10 [bench args]'
N=${1:-10}; shift || true
B="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --minibatch-steps 0 --repeat 3 --steps 200 --warmup 20"
for i in $(seq $N); do
  python3 bench.py $B "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.4f ms' % d['ms_per_step'])"
done
