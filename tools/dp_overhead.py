"""Where does the data-parallel step spend its time on ONE GPU (1-rank RCCL communicator)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.zeros(14_600_000, device="cuda:0")
for n in (14_600_000, 7441):
    v = x[:n]
    for _ in range(3):
        dist.all_reduce(v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        dist.all_reduce(v)
    torch.cuda.synchronize()
    print(f"eager 1-rank all_reduce of {n} floats: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        w = dist.all_reduce(x, async_op=True)
        y = x[:1000] * 2
        w.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    print(f"captured all_reduce replay: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per replay")
except Exception as ex:
    print("capture failed:", repr(ex))
dist.destroy_process_group()
