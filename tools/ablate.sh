#!/bin/bash
# Un-profiled contribution of single kernels to the step time: bench.py with one entry point turned into a no-op
# (tools/ablate.py wraps the library as the engine sees it).   bash tools/ablate.sh <instances> name1 name2 ...
V=$1; shift
ARGS="--instances $V --steps 100 --repeat 3 --minibatch-steps 0 --no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs"
run() { if [ -z "$1" ]; then python3 bench.py $ARGS; else python3 tools/ablate.py "$1" $ARGS; fi 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
base=$(run "")
echo "full step: $base ms"
for n in "$@"; do
  t=$(run "$n")
  echo "$n: $t ms (delta $(python3 -c "print(round(1e3*($base-$t),1))") us)"
done
echo "full step again: $(run "") ms"
