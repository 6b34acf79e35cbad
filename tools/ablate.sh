#!/bin/bash
# Un-profiled contribution of single kernels to the step time: bench.py with one entry point turned into a no-op
# (NEMO_ABLATE, nemo_cvpr2023_amd/_lib.py).   bash tools/ablate.sh <instances> name1 name2 ...
V=$1; shift
run() { env NEMO_ABLATE="$1" python3 bench.py --instances $V --steps 100 --repeat 3 --minibatch-steps 0 --no-cpu-baseline --no-torch-gpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
base=$(run "")
echo "full step: $base ms"
for n in "$@"; do
  t=$(run "$n")
  echo "$n: $t ms (delta $(python3 -c "print(round(1e3*($base-$t),1))") us)"
done
echo "full step again: $(run "") ms"
