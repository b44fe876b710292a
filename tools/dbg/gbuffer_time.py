"""Development aid: G-buffer producer time (k_gbuffer, packet vs lane form via VKR_GBUFFER_MODE) on the bistro-class scene."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vulkanrenderer_amd import Context, FrameState, GBuffer, RayTracingContext, make_bistro_class
from vulkanrenderer_amd.renderer import generate_gbuffer

scene = make_bistro_class()
for W, H in ((1920, 1080), (3840, 2160)):
    ctx = Context(W, H, 0)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    gb = GBuffer(ctx, with_shading_inputs=True)
    ubo = FrameState(W, H).update(*scene.cameras[0])
    for samples in (1, 4):
        generate_gbuffer(ctx, ubo, gb, samples)
        ctx.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            generate_gbuffer(ctx, ubo, gb, samples)
        e1.record()
        torch.cuda.synchronize()
        print(f"{os.environ.get('VKR_GBUFFER_MODE', 'packet')} {W}x{H} samples {samples}: {e0.elapsed_time(e1) / 5:.3f} ms  checksum {int(gb.linear_depth.to(torch.int64).sum().item())}")
    ctx.close()
