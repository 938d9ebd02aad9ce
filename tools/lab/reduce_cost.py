"""Cost of GradAllReducer.reduce() on a one-rank RCCL communicator at the config-3 model size (host time and GPU time)."""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from singa_amd import dp
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA
model = SINGA(load_config(lmax=4), device="cuda")
for p in model.parameters():
    p.grad = torch.randn_like(p)
red = dp.GradAllReducer(model, always=True)
for _ in range(3):
    red.reduce()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); a.record()
for _ in range(10):
    red.reduce()
b.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"reduce(): host {1e3 * (t1 - t0) / 10:.2f} ms per call, GPU {a.elapsed_time(b) / 10:.2f} ms per call, "
      f"{red.payload_bytes / 1e6:.1f} MB in {len(red.buckets)} buckets")
dist.destroy_process_group()
