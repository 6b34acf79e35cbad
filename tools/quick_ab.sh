#!/bin/bash
# Same-box A/B of environment switches on the headline, the minibatch-512 leg and the C3 bf16 leg (ms per step each):
#   gpurun -- 'bash tools/quick_ab.sh "NEMO_ORDERED_REDUCE=0" "NEMO_ORDERED_REDUCE=1"'
Q="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --steps 30 --warmup 5 --repeat 3"
for rep in 1 2; do
for kv in "$@"; do
    a=$(env $kv python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['minibatch512']['ms_per_step'])")
    b=$(env $kv python bench.py $Q --instances 40 --dtype bf16 --steps 20 --minibatch-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$kv : headline/minibatch512 $a   c3_bf16 $b"
done
done
