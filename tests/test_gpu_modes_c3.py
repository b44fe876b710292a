"""GPU parity, part 2: (a) the kernels the library selects automatically as fallbacks or through switches -- one ray per
lane (VKR_TRACE_MODE=lane), one queue item per sun ray (VKR_SUN_MODE=ray), narrow queue items (VKR_Q_RAY_BITS=6/7), the
gather form of A-Trous (VKR_ATROUS=gather) -- against the oracle on a small scene, fresh context per case; (b) BASELINE
configs[3] (bistro-class 1920x1080, 8 rays/px, temporal, 4 A-Trous iterations, two frames) against the oracle on a row
band plus whole-frame properties; (c) shadow launches on a second stream overlapping K1.  Bit-exact like test_gpu_parity."""
import numpy as np
import pytest

from helpers import compare_r16f, to_np_u16

pytestmark = pytest.mark.gpu
AO_TOL_FRAC = 0.005        # north_star: <= 0.5 % of pixels may differ by > 1/255


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


MODES = [{"VKR_TRACE_MODE": "lane"}, {"VKR_SUN_MODE": "ray"}, {"VKR_TRACE_MODE": "lane", "VKR_SUN_MODE": "ray"},
         {"VKR_Q_RAY_BITS": "6"}, {"VKR_Q_RAY_BITS": "7"}, {"VKR_ATROUS": "gather"}]


@pytest.mark.parametrize("env", MODES, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
@pytest.mark.parametrize("rays", [4, 16])
def test_switchable_and_fallback_kernels_match_the_oracle(torch_mod, oracle_mod, monkeypatch, env, rays):
    """The environment is read by vkr_create, so each case gets a fresh context."""
    from vulkanrenderer_amd import FrameState, camera_path, make_street_scene
    from vulkanrenderer_amd.renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass, ShadowRays)
    O = oracle_mod
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    scene = make_street_scene(9, 0.02, "modes")
    W, H = 176, 98
    osc = O.OracleScene(scene)
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    gb = GBuffer(ctx, with_shading_inputs=True)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=3, want_hit_mask=True)
    opass = O.OracleRtaoPass(osc, W, H, 3)
    sh = ShadowRays(ctx, gb)
    fs = FrameState(W, H, ray_count=rays, sun_samples=8, point_lights=1)
    for frame, (eye, tgt) in enumerate(camera_path(scene, 2)):
        ubo = fs.update(eye, tgt)
        depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
        gb.upload(depth, normal, wp, ng, valid)
        rpass.execute(PassExecuteContext(ubo=ubo))
        sun = sh.sun(ubo)
        pt = sh.point(ubo, 0)
        ctx.sync()
        ref_full, ref_mask = opass.execute(ubo, depth, normal, want_mask=True)
        full = compare_r16f(to_np_u16(rpass.rtao_full), ref_full)
        hist = compare_r16f(to_np_u16(rpass.image(0)), opass.hist[opass.last_curr])
        got_mask = rpass.hit_mask.cpu().numpy().view(np.uint32)
        assert full["frac_gt_1_255"] <= AO_TOL_FRAC and hist["frac_gt_1_255"] <= AO_TOL_FRAC
        assert int(np.unpackbits((got_mask ^ ref_mask).view(np.uint8)).sum()) == 0
        assert full["bit_mismatch"] == 0 and hist["bit_mismatch"] == 0, (env, frame, full, hist)
        ref_sun = osc.shadow_sun(wp, ng, valid, FrameState.sun_dir_to_light(ubo), 0.01, 8)
        assert np.array_equal(sun.cpu().numpy(), ref_sun), (env, frame, "sun")
        from vulkanrenderer_amd.scenes import UBO_LIGHT_POSITIONS
        ref_pt = osc.shadow_point(wp, ng, valid, ubo[UBO_LIGHT_POSITIONS:UBO_LIGHT_POSITIONS + 3])
        assert np.array_equal(pt.cpu().numpy(), ref_pt), (env, frame, "point")
    ctx.close()


def _oracle_chain_band(O, osc, ubo, depth, normal, hist_prev, iters, ya, yb):
    H, W = depth.shape
    halo = 2 * ((1 << iters) - 1)
    t0, t1 = max(0, ya - halo), min(H, yb + halo)
    cur, mask = osc.trace(ubo, depth, hist_prev, t0, t1, want_mask=True)
    src = np.asarray(cur).view(np.uint16).reshape(H, W)
    for i in range(iters):
        rem = 2 * ((1 << iters) - (1 << (i + 1)))
        src = O.atrous(ubo, 1 << i, src, depth, normal, max(0, ya - rem), min(H, yb + rem))
    return src[ya:yb], np.asarray(cur).view(np.uint16).reshape(H, W)[ya:yb], np.asarray(mask).reshape(H, W)[ya:yb]


def test_config3_bistro_class_1080p_two_frames(torch_mod, oracle_mod):
    """BASELINE.json configs[3]: bistro-class scene (2.78 M triangles), 1920x1080, 8 rays/px + temporal + 4 A-Trous iterations,
    two frames with a moving camera.  Oracle: the whole chain on two row bands per frame (frame 1 starts from the GPU's own
    frame-0 history, which frame 0's check has compared on the bands); whole frame: value range, sky, ray counters."""
    from vulkanrenderer_amd import FrameState, camera_path, make_bistro_class
    from vulkanrenderer_amd.renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass,
                                             generate_gbuffer, trace_stats, trace_stats_enable)
    O = oracle_mod
    scene = make_bistro_class()
    W, H, rays, iters = 1920, 1080, 8, 4
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    assert rt.info().triangle_count == scene.triangle_count() > 2_000_000
    osc = O.OracleScene(scene)
    gb = GBuffer(ctx)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=iters, want_hit_mask=True)
    fs = FrameState(W, H, ray_count=rays)
    ONE = np.float16(1.0).view(np.uint16)
    hist_prev = np.full((H, W), ONE, dtype=np.uint16)
    bands = [(200, 330), (620, 760)]                                      # 270 rows of the frame per frame, through the whole chain
    trace_stats_enable(ctx, True)
    for frame, (eye, tgt) in enumerate(camera_path(scene, 2, step=0.05)):
        ubo = fs.update(eye, tgt)
        generate_gbuffer(ctx, ubo, gb)
        trace_stats(ctx, reset=True)
        rpass.execute(PassExecuteContext(ubo=ubo))
        ctx.sync()
        st = trace_stats(ctx, reset=True)
        depth, normal = to_np_u16(gb.linear_depth).reshape(H, W), to_np_u16(gb.normal).reshape(H, W, 4)
        full, hist = to_np_u16(rpass.rtao_full).reshape(H, W), to_np_u16(rpass.image(0)).reshape(H, W).copy()
        mask = rpass.hit_mask.cpu().numpy().view(np.uint32).reshape(H, W)
        for ya, yb in bands:
            ref_full, ref_hist, ref_mask = _oracle_chain_band(O, osc, ubo, depth, normal, hist_prev, iters, ya, yb)
            cf, ch = compare_r16f(full[ya:yb], ref_full), compare_r16f(hist[ya:yb], ref_hist)
            print(f"c3 frame {frame} rows [{ya},{yb}): rtao_full {cf} history {ch}")
            assert cf["frac_gt_1_255"] <= AO_TOL_FRAC and ch["frac_gt_1_255"] <= AO_TOL_FRAC
            assert int(np.unpackbits((mask[ya:yb] ^ ref_mask).view(np.uint8)).sum()) == 0
            assert cf["bit_mismatch"] == 0 and ch["bit_mismatch"] == 0
        # whole frame
        f32 = full.view(np.float16).astype(np.float32)
        assert np.isfinite(f32).all() and f32.min() >= 0.0 and f32.max() <= 1.0
        sky = depth.view(np.float16).astype(np.float32) <= 1e-6
        assert np.all(hist[sky] == ONE)                                   # :240-243 stores exactly 1.0 for sky pixels
        nvalid = int((~sky).sum())
        assert rays * nvalid <= st.rays <= 2 * rays * nvalid              # one query per ray + at most one retry each
        assert st.nodes_visited >= st.rays and st.tris_tested > 0
        hist_prev = hist
    trace_stats_enable(ctx, False)
    ctx.close()


def test_shadow_rays_on_the_bistro_class_tree_match_the_oracle(torch_mod, oracle_mod):
    """K4 / K5 where the tree is deep and MASK-heavy: sun-disc shadows (16 samples, radius 0.01 rad, pbr.frag:217-260) and the
    three point lights (pbr.frag:169-196) over the bistro-class scene at 1920x1080, two row bands each against the oracle's
    per-ray any-hit queries (every triangle occludes: TerminateOnFirstHit | Opaque, pbr.frag:199-215).  Bit-exact."""
    from vulkanrenderer_amd import FrameState, make_bistro_class
    from vulkanrenderer_amd.renderer import Context, GBuffer, RayTracingContext, ShadowRays, generate_gbuffer
    from vulkanrenderer_amd.scenes import UBO_LIGHT_POSITIONS
    O = oracle_mod
    scene = make_bistro_class()
    W, H = 1920, 1080
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    osc = O.OracleScene(scene)
    gb = GBuffer(ctx, with_shading_inputs=True)
    fs = FrameState(W, H, ray_count=8, sun_samples=16, point_lights=3)
    ubo = fs.update(*scene.cameras[0])
    generate_gbuffer(ctx, ubo, gb)
    sh = ShadowRays(ctx, gb)
    sun = sh.sun(ubo).cpu().numpy()
    pts = [sh.point(ubo, k).cpu().numpy().copy() for k in range(3)]
    ctx.sync()
    wp, ng, valid = gb.world_pos.cpu().numpy(), gb.geom_normal.cpu().numpy(), gb.valid.cpu().numpy()
    assert valid.mean() > 0.5
    L = FrameState.sun_dir_to_light(ubo)
    for ya, yb in [(260, 308), (700, 748)]:
        sl = slice(ya, yb)
        ref_sun = osc.shadow_sun(np.ascontiguousarray(wp[sl]), np.ascontiguousarray(ng[sl]), np.ascontiguousarray(valid[sl]), L, 0.01, 16)
        assert np.array_equal(sun[sl].view(np.uint32), np.asarray(ref_sun).reshape(yb - ya, W).view(np.uint32)), ("sun", ya)
        for k in range(3):
            P = ubo[UBO_LIGHT_POSITIONS + 4 * k:UBO_LIGHT_POSITIONS + 4 * k + 3]
            ref_pt = osc.shadow_point(np.ascontiguousarray(wp[sl]), np.ascontiguousarray(ng[sl]), np.ascontiguousarray(valid[sl]), P)
            assert np.array_equal(pts[k][sl].view(np.uint32), np.asarray(ref_pt).reshape(yb - ya, W).view(np.uint32)), ("point", k, ya)
    # the bands are not trivially lit or dark
    band = np.concatenate([sun[260:308], sun[700:748]])
    assert 0.0 < float(band.mean()) < 1.0 or float(band.std()) > 0.0
    ctx.close()


def test_config4_4k_band_through_k1_and_atrous_matches_the_oracle(torch_mod, oracle_mod):
    """BASELINE.json configs[4] at its own size: bistro-class 3840x2160, 8 AO rays/px + temporal + 4 A-Trous iterations, two frames;
    one 40-row band per frame through the whole chain against the oracle (frame 1 from the GPU's own frame-0 history), and a band
    of the 16-sample sun shadows.  The full-size sharded == unsharded property lives in test_gpu_bvh_shadows_bands.py."""
    from vulkanrenderer_amd import FrameState, camera_path, make_bistro_class
    from vulkanrenderer_amd.renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass, ShadowRays,
                                             generate_gbuffer)
    O = oracle_mod
    scene = make_bistro_class()
    W, H, rays, iters = 3840, 2160, 8, 4
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    osc = O.OracleScene(scene)
    gb = GBuffer(ctx, with_shading_inputs=True)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=iters, want_hit_mask=True)
    sh = ShadowRays(ctx, gb)
    fs = FrameState(W, H, ray_count=rays, sun_samples=16)
    ONE = np.float16(1.0).view(np.uint16)
    hist_prev = np.full((H, W), ONE, dtype=np.uint16)
    for frame, (eye, tgt) in enumerate(camera_path(scene, 2, step=0.05)):
        ubo = fs.update(eye, tgt)
        generate_gbuffer(ctx, ubo, gb)
        rpass.execute(PassExecuteContext(ubo=ubo))
        sun = sh.sun(ubo).cpu().numpy()
        ctx.sync()
        depth, normal = to_np_u16(gb.linear_depth).reshape(H, W), to_np_u16(gb.normal).reshape(H, W, 4)
        full, hist = to_np_u16(rpass.rtao_full).reshape(H, W), to_np_u16(rpass.image(0)).reshape(H, W).copy()
        mask = rpass.hit_mask.cpu().numpy().view(np.uint32).reshape(H, W)
        ya, yb = (1300, 1340) if frame == 0 else (640, 680)
        ref_full, ref_hist, ref_mask = _oracle_chain_band(O, osc, ubo, depth, normal, hist_prev, iters, ya, yb)
        cf, ch = compare_r16f(full[ya:yb], ref_full), compare_r16f(hist[ya:yb], ref_hist)
        print(f"c4 frame {frame} rows [{ya},{yb}): rtao_full {cf} history {ch}")
        assert cf["frac_gt_1_255"] <= AO_TOL_FRAC and ch["frac_gt_1_255"] <= AO_TOL_FRAC
        assert int(np.unpackbits((mask[ya:yb] ^ ref_mask).view(np.uint8)).sum()) == 0
        assert cf["bit_mismatch"] == 0 and ch["bit_mismatch"] == 0
        if frame == 0:
            sl = slice(1500, 1524)
            wp, ng, valid = gb.world_pos[sl].cpu().numpy(), gb.geom_normal[sl].cpu().numpy(), gb.valid[sl].cpu().numpy()
            ref_sun = osc.shadow_sun(wp, ng, valid, FrameState.sun_dir_to_light(ubo), 0.01, 16)
            assert np.array_equal(sun[sl].view(np.uint32), np.asarray(ref_sun).reshape(24, W).view(np.uint32))
        hist_prev = hist
    ctx.close()


@pytest.mark.parametrize("sun_mode", ["packet", "ray"])
def test_shadow_launches_on_a_second_stream_overlap_k1(torch_mod, monkeypatch, sun_mode):
    """k_shadow_q (point lights always; the sun under VKR_SUN_MODE=ray) has its own chunk counter and ray slices, so both
    shadow calls may run on a second stream while K1 of the same context is in flight: the bits are those of the
    one-stream order, several frames in a row, at a size where all kernels run many persistent waves for a while."""
    import torch
    from vulkanrenderer_amd import FrameState, make_sponza_class
    from vulkanrenderer_amd.renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass, ShadowRays,
                                             generate_gbuffer)
    if sun_mode == "ray":
        monkeypatch.setenv("VKR_SUN_MODE", "ray")
    scene = make_sponza_class()
    W, H = 1280, 720
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    gb = GBuffer(ctx, with_shading_inputs=True)
    fs = FrameState(W, H, ray_count=8, sun_samples=8, point_lights=1)
    eye, tgt = scene.cameras[0]
    ubos = [fs.update((eye[0] + 0.05 * f, eye[1], eye[2]), tgt) for f in range(3)]
    generate_gbuffer(ctx, ubos[0], gb)
    ctx.sync()
    sh = ShadowRays(ctx, gb)
    rp = RtaoComputePass(ctx, gb, atrous_iterations=2)
    want = []
    for ubo in ubos:                                   # one-stream reference
        rp.execute(PassExecuteContext(ubo=ubo))
        sun = sh.sun(ubo).clone()
        pt = sh.point(ubo, 0).clone()
        ctx.sync()
        want.append((to_np_u16(rp.rtao_full).copy(), sun.cpu().numpy(), pt.cpu().numpy()))
    rp.reset()
    side, ev = torch.cuda.Stream(), torch.cuda.Event()
    for k, ubo in enumerate(ubos):                     # shadows on the second stream, issued BEFORE the RTAO frame so they start first
        with torch.cuda.stream(side):
            sh.point(ubo, 0)
            sh.sun(ubo)
            ev.record(side)
        rp.execute(PassExecuteContext(ubo=ubo))
        torch.cuda.current_stream().wait_event(ev)
        ctx.sync()
        assert to_np_u16(rp.rtao_full).tobytes() == want[k][0].tobytes(), k
        assert sh.sun_vis.cpu().numpy().tobytes() == want[k][1].tobytes(), k
        assert sh.point_vis[0].cpu().numpy().tobytes() == want[k][2].tobytes(), k
    ctx.close()
