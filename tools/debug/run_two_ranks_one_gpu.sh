export NEMO_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/r04_g2.json 2> gpurun_out/r04_g2.err
echo rc=$?
grep -v "amdgpu.ids\|hostname\|Gloo\|^$" gpurun_out/r04_g2.err | tail -15
