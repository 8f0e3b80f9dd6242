"""The stream-ordered all-reduce probe of bench.py on a one-rank RCCL group (debug aid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
from gpirt_amd.dist import torch_allreduce
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
world = 1
probe = torch_allreduce(device_buffers=True, on_stream=True)
side = torch.cuda.Stream()
probe.stream_ptr = side.cuda_stream
ones = torch.ones(1024, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
probe(ones.data_ptr(), ones.numel())
side.synchronize()
print("probe ok:", bool(torch.all(ones == float(world)).item()))
dist.destroy_process_group()
