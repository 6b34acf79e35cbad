#!/bin/bash
V=${1:-8}; R=${2:-2}; shift 2
for i in $(seq $R); do
  (cd _base && python bench.py --instances $V --steps 100 --no-cpu-baseline --no-torch-gpu-baseline 2>/dev/null | python -c "import sys,json; print('base', json.loads(sys.stdin.read())['ms_per_step'])")
  for e in "$@"; do env ${e//,/ } python bench.py --instances $V --steps 100 --no-cpu-baseline --no-torch-gpu-baseline 2>/dev/null | python -c "import sys,json; print('new $e', json.loads(sys.stdin.read())['ms_per_step'])"; done
done
