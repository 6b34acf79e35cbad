"""How much does the time of ONE captured step graph depend on its instantiation?  The 8 x 300 step is captured `trials` times in one
process (same kernels, same dependencies; the runtime assigns the graph's branches to hardware queues at instantiation) and each
instance is timed over `steps` replays:   gpurun -- 'python3 tools/graph_lottery.py [trials] [steps]'"""
import sys
import time
import types

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cx = types.SimpleNamespace(sharded=False, device='cuda:0', rank=0, world=1)
torch.cuda.set_device(0)
model, engine, args = bench.build_model(cx, 8, 300, 'f32')
step = lambda: model.step(None, None, update=True, full_batch=True)
for _ in range(6):
    step()
w = next(w for w in engine.ws.values() if any(isinstance(g, torch.cuda.CUDAGraph) for g in w['graphs'].values()))
key = next(k for k, g in w['graphs'].items() if isinstance(g, torch.cuda.CUDAGraph))
for t in range(trials):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    print('instance %d: %.4f ms per step' % (t, 1e3 * (time.perf_counter() - t0) / steps), flush=True)
    w['graphs'][key] = model.GRAPH_AFTER          # the next step captures the variant again
