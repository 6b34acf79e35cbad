#!/bin/bash
# VPoser chain on nemo_gemm_xp (NEMO_VP_XP_MIN_ROWS, here from 8192 rows) against its fp32 launches (the default), ms per step:  gpurun -- 'bash tools/ab_vposer_xp.sh'
B="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --minibatch-steps 0 --repeat 1"
run() { python3 bench.py $B "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.4f ms' % d['ms_per_step'])"; }
for i in 1 2; do
  for shape in "--instances 40 --steps 30 --warmup 5" "--instances 256 --frames 1024 --steps 4 --warmup 2"; do
    echo -n "$shape  xp    "; NEMO_VP_XP_MIN_ROWS=8192 run $shape
    echo -n "$shape  fp32  "; run $shape
  done
done
