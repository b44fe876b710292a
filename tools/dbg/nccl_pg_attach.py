"""GPU box: the path bench.py takes for N > 1 under RCCL, with the one rank a 1-GPU box allows -- the process group's own
communicator (ProcessGroupNCCL._comm_ptr) attached through vkr_comm_attach, the two C-ABI gathers issued on a communication stream
and ordered by events, result compared with vkr_rtao_execute on a second context."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
import numpy as np
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.cuda.set_device(0)
dist.barrier()
backend = dist.distributed_c10d._get_default_group()._get_backend(dev)
ptr = int(backend._comm_ptr())
print("comm ptr", hex(ptr))
from vulkanrenderer_amd import FrameState, camera_path, make_test_scene
from vulkanrenderer_amd.renderer import Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass, generate_gbuffer
scene = make_test_scene()
W, H = 322, 187
outs = []
for sharded in (False, True):
    ctx = Context(W, H)
    rt = RayTracingContext(ctx); rt.init(scene)
    gb = GBuffer(ctx, with_shading_inputs=True)
    rp = RtaoComputePass(ctx, gb, atrous_iterations=4)
    fs = FrameState(W, H, ray_count=4)
    if sharded:
        ctx.attachComm(ptr, 0, 1)
        comm = torch.cuda.Stream(device=dev)
        ev_k1, ev_hist, ev_dn, ev_out = (torch.cuda.Event() for _ in range(4))
    for eye, tgt in camera_path(scene, 3):
        ubo = fs.update(eye, tgt)
        generate_gbuffer(ctx, ubo, gb)
        if not sharded:
            rp.execute(PassExecuteContext(ubo=ubo))
            continue
        main = torch.cuda.current_stream()
        rp.traceStripes(ubo, 0, 1); ev_k1.record(main)
        comm.wait_event(ev_k1)
        with torch.cuda.stream(comm):
            rp.gatherHistory(); ev_hist.record(comm)
        main.wait_event(ev_hist)
        rp.denoiseBand(ubo, 0, H); ev_dn.record(main)
        comm.wait_event(ev_dn)
        with torch.cuda.stream(comm):
            ctx.gatherBands([rp.rtao_full]); ev_out.record(comm)
        main.wait_event(ev_out)
    ctx.sync(); torch.cuda.synchronize()
    outs.append((rp.rtao_full.cpu().numpy().copy(), rp.image(0).cpu().numpy().copy()))
    if sharded:
        ctx.attachComm(0, 0, 0)
    ctx.close()
assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
dist.barrier()
print("process-group communicator through vkr_comm_attach: ok, sharded == unsharded")
dist.destroy_process_group()
