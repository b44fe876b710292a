"""Development aid: counters of k_shadow_sun_pkt (library built with -DVKR_SUN_STATS) on the C4 frame."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401

from vulkanrenderer_amd import Context, FrameState, GBuffer, RayTracingContext, make_bistro_class
from vulkanrenderer_amd.renderer import ShadowRays, generate_gbuffer, trace_stats

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
sh.sun(ubo)
ctx.sync()
trace_stats(ctx, True)
sh.sun(ubo)
ctx.sync()
st = trace_stats(ctx, True)
pix, node, pre, wnode, wtri, tri = st.rays, st.nodes_visited, st.tris_tested, st.wave_node_iters, st.wave_tri_iters, st.overflow_tiles
print(f"pixels {pix}  node items/px {node / pix:.2f}  leaf triangles/px {pre / pix:.2f}  after cone filter/px {tri / pix:.2f}")
print(f"wave node steps {wnode} (lane util {node / (64 * max(wnode, 1)):.2f})  wave tri steps {wtri} (lane util {tri / (64 * max(wtri, 1)):.2f})")
