"""Worker of tests/test_gpu_smallq.py: parity of the library named by VKR_LIB_PATH against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O                                                    # noqa: E402
from vulkanrenderer_amd import FrameState, _capi, camera_path, make_street_scene  # noqa: E402
from vulkanrenderer_amd.renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass,  # noqa: E402
                                         ShadowRays)


def main():
    assert os.path.basename(_capi.LIB_PATH) in ("libvkr_rt_smallq.so", "libvkr_rt_w7dbg.so"), _capi.LIB_PATH
    scene = make_street_scene(7, 0.02, "smallq")
    W, H = 96, 54
    osc = O.OracleScene(scene)
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    gb = GBuffer(ctx)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=2, want_hit_mask=True)
    opass = O.OracleRtaoPass(osc, W, H, 2)
    sh = ShadowRays(ctx, gb)
    fs = FrameState(W, H, ray_count=8, sun_samples=16, point_lights=1)
    for eye, tgt in camera_path(scene, 2):
        ubo = fs.update(eye, tgt)
        depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
        gb.upload(depth, normal, wp, ng, valid)
        rpass.execute(PassExecuteContext(ubo=ubo))
        ctx.sync()
        ref_full, ref_mask = opass.execute(ubo, depth, normal, want_mask=True)
        assert np.array_equal(rpass.rtao_full.cpu().numpy().view(np.uint16), ref_full)
        assert np.array_equal(rpass.hit_mask.cpu().numpy().view(np.uint32), ref_mask)
        sun = sh.sun(ubo).cpu().numpy()
        assert np.array_equal(sun.view(np.uint32),
                              osc.shadow_sun(wp, ng, valid, FrameState.sun_dir_to_light(ubo), 0.01, 16).view(np.uint32))
        pt = sh.point(ubo, 0).cpu().numpy()
        assert np.array_equal(pt.view(np.uint32), osc.shadow_point(wp, ng, valid, ubo[60:63]).view(np.uint32))
    # the variant libraries are built with -DVKR_DEBUG_BOUNDS: under VKR_PHASE_DUMP the counters call prints the violation bits
    from vulkanrenderer_amd.renderer import trace_stats
    trace_stats(ctx, reset=True)
    print("smallq ok")


if __name__ == "__main__":
    main()
