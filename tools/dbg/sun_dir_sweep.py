"""Development aid: k_shadow_sun_pkt time at 4K (16 samples, radius 0.01) for the eight sign combinations of the sun direction
(|d| = (0.3, 0.8, 0.5) as in FrameManager.cpp:470).  Run once per build (VKR_LIB_PATH) to compare child-slot visiting orders."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from vulkanrenderer_amd import Context, FrameState, GBuffer, RayTracingContext, make_bistro_class
from vulkanrenderer_amd.renderer import ShadowRays, generate_gbuffer
from vulkanrenderer_amd.scenes import UBO_DIR_LIGHT_DIR

W, H = 3840, 2160
scene = make_bistro_class()
ctx = Context(W, H, 0)
rt = RayTracingContext(ctx)
rt.init(scene)
gb = GBuffer(ctx, with_shading_inputs=True)
ubo = FrameState(W, H, ray_count=8, sun_samples=16).update(*scene.cameras[0])
generate_gbuffer(ctx, ubo, gb)
ctx.sync()
sh = ShadowRays(ctx, gb)
out = []
for sx, sy, sz in itertools.product((1, -1), repeat=3):
    u = ubo.copy()
    u[UBO_DIR_LIGHT_DIR:UBO_DIR_LIGHT_DIR + 3] = (-0.3 * sx, -0.8 * sy, -0.5 * sz)   # L = -dir
    sh.sun(u); ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        sh.sun(u)
    e1.record(); torch.cuda.synchronize()
    lit = float(sh.sun_vis.mean().item())
    out.append(f"L signs ({sx:+d},{sy:+d},{sz:+d}): {e0.elapsed_time(e1) / 3:.3f} ms  mean visibility {lit:.3f}")
print("\n".join(out))
