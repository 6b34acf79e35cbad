"""Can an RCCL all-reduce be captured into a HIP graph on this stack (torch.distributed nccl backend, world of one)?
   python3 tools/debug/graph_capture_rccl.py"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dist.init_process_group('nccl', rank=0, world_size=1)
torch.cuda.set_device(0)
x = torch.ones(2_260_000, device='cuda:0')
y = torch.zeros(1024, device='cuda:0')
dist.all_reduce(x); torch.cuda.synchronize()          # communicator set up outside the capture
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        y.add_(1.0)
        dist.all_reduce(x)
        y.mul_(2.0)
    ok = True
except Exception as ex:
    ok = False
    print('capture failed:', type(ex).__name__, str(ex)[:300])
if ok:
    y.zero_(); torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print('capture ok; y[0] after 3 replays =', float(y[0]), '(expected 14)')
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        g.replay()
    b.record(); torch.cuda.synchronize()
    print('graph replay with the all-reduce inside: %.1f us' % (a.elapsed_time(b) * 1e3 / 200))
    a.record()
    for _ in range(200):
        y.add_(1.0); dist.all_reduce(x); y.mul_(2.0)
    b.record(); torch.cuda.synchronize()
    print('eager: %.1f us' % (a.elapsed_time(b) * 1e3 / 200))
dist.destroy_process_group()
