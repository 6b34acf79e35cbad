#!/bin/bash
# HIP runtime switches against the defaults at 8 x 300 (same box, alternating):  gpurun -- 'bash tools/ab_runtime_env.sh'
B="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --minibatch-steps 0 --repeat 3 --steps 200 --warmup 20"
run() { python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.4f ms' % d['ms_per_step'])"; }
for i in 1 2; do
  echo -n "base                      "; run
  echo -n "HIP_FORCE_DEV_KERNARG=1   "; HIP_FORCE_DEV_KERNARG=1 run
  echo -n "HIP_FORCE_DEV_KERNARG=0   "; HIP_FORCE_DEV_KERNARG=0 run
  echo -n "GPU_MAX_HW_QUEUES=8       "; GPU_MAX_HW_QUEUES=8 run
  echo -n "GPU_MAX_HW_QUEUES=2       "; GPU_MAX_HW_QUEUES=2 run
  echo -n "GRAPH_PACKET_CAPTURE=1    "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 run
  echo -n "GRAPH_PACKET_CAPTURE=0    "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run
done
