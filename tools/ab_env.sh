#!/bin/bash
# A/B of environment switches on one box:  bash tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...   (each variant 3 runs)
set -u
ARGS=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then
      r=$(python3 bench.py $ARGS --no-cpu-baseline --no-torch-gpu-baseline 2>/dev/null | tail -1)
    else
      r=$(env $v python3 bench.py $ARGS --no-cpu-baseline --no-torch-gpu-baseline 2>/dev/null | tail -1)
    fi
    echo "$v rep$rep ms_per_step=$(echo "$r" | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  done
done
