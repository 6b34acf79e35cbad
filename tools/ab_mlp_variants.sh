#!/bin/bash
# ms per step of the three MotionNet-chain arithmetics (NEMO_MLP_GEMM) at several batch shapes:  gpurun -- 'bash tools/ab_mlp_variants.sh'
B="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --minibatch-steps 0 --repeat 1"
run() { python3 bench.py $B "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.4f ms' % d['ms_per_step'])"; }
for shape in "--instances 1 --steps 100" "--instances 2 --steps 100" "--instances 4 --steps 60" "--instances 8 --steps 50" "--instances 40 --steps 20" "--instances 256 --frames 1024 --steps 3 --warmup 1"; do
  for v in f32 f32_split3 f32_split2; do
    echo -n "$shape  $v  "; NEMO_XP_MIN_ROWS=0 NEMO_MLP_GEMM=$v run $shape
  done
done
