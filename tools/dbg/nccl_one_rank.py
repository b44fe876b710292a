"""GPU box: exercise the RCCL calls of bands.py with a one-rank communicator (all a 1-GPU box allows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
send = torch.arange(34 * 8 * 3840, dtype=torch.int32, device="cuda").to(torch.int16).view(34 * 8, 3840)
recv = torch.zeros((1, 34 * 8, 3840), dtype=torch.int16, device="cuda")
w = dist.all_gather_into_tensor(recv.view(-1).view(torch.uint8), send.contiguous().view(-1).view(torch.uint8), async_op=True)
x = send.float().sum()          # independent work while the collective runs
w.wait()
assert torch.equal(recv[0], send)
recv.zero_()
dist.all_gather_into_tensor(recv.view(-1).view(torch.uint8), send.view(-1).view(torch.uint8))
assert torch.equal(recv[0], send)
# two collectives in flight, waited out of issue order around independent kernels (bench.py's frame pipeline)
from vulkanrenderer_amd import bands
a_send = torch.randint(-30000, 30000, (2, 270, 3840), dtype=torch.int16, device="cuda"); a_recv = torch.zeros((1, 2, 270, 3840), dtype=torch.int16, device="cuda")
b_send = torch.randint(-30000, 30000, (272, 3840), dtype=torch.int16, device="cuda"); b_recv = torch.zeros((1, 272, 3840), dtype=torch.int16, device="cuda")
for it in range(3):
    wa = bands.all_gather_frame(a_send, a_recv, async_op=True)
    y = (send.float() * 2).sum()
    wb = bands.all_gather_bytes(b_send, b_recv, async_op=True)
    wa.wait(); z = a_recv.float().sum()
    wb.wait()
    assert torch.equal(b_recv[0], b_send) and torch.equal(a_recv[0], a_send)
    a_send += 1; b_send -= 1
torch.cuda.synchronize()
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("rccl one-rank ok", float(x), float(t))
dist.destroy_process_group()
