"""Development aid: counters of traverse_packet (library built with -DVKR_PKT_STATS): node visits and triangle tests per 8x8 tile."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401

from vulkanrenderer_amd import Context, FrameState, GBuffer, RayTracingContext, make_bistro_class
from vulkanrenderer_amd.renderer import generate_gbuffer, trace_stats

W, H = 1920, 1080
scene = make_bistro_class()
ctx = Context(W, H, 0)
rt = RayTracingContext(ctx)
rt.init(scene)
gb = GBuffer(ctx, with_shading_inputs=True)
ubo = FrameState(W, H).update(*scene.cameras[0])
trace_stats(ctx, True)
generate_gbuffer(ctx, ubo, gb)
ctx.sync()
st = trace_stats(ctx, True)
print(f"packets {st.rays}  node visits/packet {st.nodes_visited / max(1, st.rays):.1f}  triangle tests/packet {st.tris_tested / max(1, st.rays):.1f}")
