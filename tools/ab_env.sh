#!/bin/bash
# same-box A/B of one environment switch at 8 x 300, alternating:  gpurun -- 'bash tools/ab_env.sh NEMO_KP_FIN_LATE 1 [rounds] [bench args]'
# prints ms per step with the variable unset / set, `rounds` times each (default 3)
VAR=$1; VAL=$2; R=${3:-3}; shift 3 || shift $#
B="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --minibatch-steps 0 --repeat 3 --steps 200 --warmup 20"
run() { python3 bench.py $B "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.4f ms' % d['ms_per_step'])"; }
for i in $(seq $R); do
  echo -n "unset      "; run "$@"
  echo -n "$VAR=$VAL  "; env $VAR=$VAL python3 bench.py $B "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.4f ms' % d['ms_per_step'])"
done
