"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): per-ray hit/miss identical except grazing rays; final AO <= 0.5 % of pixels
differing by > 1/255, max-abs stated.  Because both sides implement the same numeric contract the
tests additionally require bit-exact R16F outputs and zero ray mismatches; the north-star tolerances are asserted
first so a future contract relaxation fails on the right line.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import compare_r16f, to_np_u16

pytestmark = pytest.mark.gpu

AO_TOL_FRAC = 0.005        # north_star: <= 0.5 % of pixels may differ by > 1/255
ONE16 = np.float16(1.0).view(np.uint16)


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _setup(scene, W, H, rays, iters, torch, brute=False, want_mask=True, **fs_kw):
    from oracle import oracle as O
    from vulkanrenderer_amd import FrameState
    from vulkanrenderer_amd.renderer import Context, RayTracingContext, GBuffer, RtaoComputePass
    osc = O.OracleScene(scene, brute_force=brute)
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    gb = GBuffer(ctx)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=iters, want_hit_mask=want_mask)
    opass = O.OracleRtaoPass(osc, W, H, iters)
    fs = FrameState(W, H, ray_count=rays, **fs_kw)
    return O, osc, ctx, rt, gb, rpass, opass, fs


def _check_frame(tag, ctx, rpass, opass, ref_full, ref_mask):
    got_full = to_np_u16(rpass.rtao_full)
    hist = compare_r16f(to_np_u16(rpass.image(0)), opass.hist[opass.last_curr])
    full = compare_r16f(got_full, ref_full)
    ray_mismatch = 0
    if ref_mask is not None:
        got_mask = rpass.hit_mask.cpu().numpy().view(np.uint32)
        ray_mismatch = int(np.unpackbits((got_mask ^ ref_mask).view(np.uint8)).sum())
    print(f"{tag}: history {hist} rtao_full {full} ray hit/miss mismatches {ray_mismatch}")
    assert hist["frac_gt_1_255"] <= AO_TOL_FRAC and full["frac_gt_1_255"] <= AO_TOL_FRAC
    assert ray_mismatch == 0
    assert hist["bit_mismatch"] == 0 and full["bit_mismatch"] == 0


def _run_chain(scene, W, H, rays, iters, frames, torch, tag, **fs_kw):
    from vulkanrenderer_amd import camera_path
    from vulkanrenderer_amd.renderer import PassExecuteContext
    O, osc, ctx, rt, gb, rpass, opass, fs = _setup(scene, W, H, rays, iters, torch, **fs_kw)
    assert rt.info().triangle_count == osc.triangle_count() == scene.triangle_count()
    for frame, (eye, tgt) in enumerate(camera_path(scene, frames)):
        ubo = fs.update(eye, tgt)
        depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
        gb.upload(depth, normal, wp, ng, valid)
        rpass.execute(PassExecuteContext(ubo=ubo))
        ctx.sync()
        ref_full, ref_mask = opass.execute(ubo, depth, normal, want_mask=True)
        _check_frame(f"{tag} frame {frame}", ctx, rpass, opass, ref_full, ref_mask)
    ctx.close()


def test_rtao_chain_small_scene_three_frames(torch_mod, oracle_mod):
    from vulkanrenderer_amd import make_street_scene
    _run_chain(make_street_scene(7, 0.02, "t"), 160, 90, 4, 2, 3, torch_mod, "street160x90")


def test_config0_sponza_class_640x360(torch_mod, oracle_mod):
    """BASELINE.json configs[0]: Sponza 640x360, 4 rays/px + temporal + 2 A-Trous iterations (synthetic
    sponza-class scene of the same size; the glTF itself is not in the image)."""
    from vulkanrenderer_amd import make_sponza_class
    _run_chain(make_sponza_class(), 640, 360, 4, 2, 2, torch_mod, "c0")


@pytest.mark.parametrize("W,H,rays,iters", [(67, 45, 1, 1), (33, 17, 16, 5), (8, 8, 8, 4), (130, 7, 3, 3)])
def test_ragged_sizes_ray_counts_and_iterations(torch_mod, oracle_mod, W, H, rays, iters):
    from vulkanrenderer_amd import make_test_scene
    _run_chain(make_test_scene(), W, H, rays, iters, 2, torch_mod, f"{W}x{H} r{rays} i{iters}")


def test_reference_default_five_iterations_at_an_odd_hd_size(torch_mod, oracle_mod):
    """The reference ships 5 A-Trous iterations (steps 1..16, AppConfig.h:88-96).  At 1283x725 the step-16 iteration has 256
    comb sub-images of 81x46 texels, ragged in both directions, and the work items of k_atrous_tile do not divide by the 8 XCD
    runs they are dealt out in: every pixel of rtao_full and of the history against the oracle."""
    from vulkanrenderer_amd import make_sponza_class
    _run_chain(make_sponza_class(), 1283, 725, 2, 5, 1, torch_mod, "1283x725 r2 i5")


def test_ray_count_is_clamped_like_the_shader(torch_mod, oracle_mod):
    # rtao_trace_half.comp:252 clamps rayCount to [1, 16]; 40 and 0 must behave as 16 and 1 on both sides
    from vulkanrenderer_amd import make_test_scene
    _run_chain(make_test_scene(), 40, 24, 40, 1, 1, torch_mod, "rays40")
    _run_chain(make_test_scene(), 40, 24, 0, 1, 1, torch_mod, "rays0")


def test_material_matrix_scene(torch_mod, oracle_mod):
    # opaque / MASK above+below cutoff / BLEND in front of a wall (SURVEY 8c pin 6)
    from vulkanrenderer_amd import make_material_matrix_scene
    from vulkanrenderer_amd.renderer import PassExecuteContext
    scene = make_material_matrix_scene()
    W, H = 128, 48
    O, osc, ctx, rt, gb, rpass, opass, fs = _setup(scene, W, H, 8, 1, torch_mod, radius=3.0)
    scene.cameras = [((3.5, 0.5, -4.0), (3.5, 0.5, 2.0))]
    ubo = fs.update(*scene.cameras[0])
    depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
    assert valid.any()
    gb.upload(depth, normal, wp, ng, valid)
    rpass.execute(PassExecuteContext(ubo=ubo))
    ctx.sync()
    ref_full, ref_mask = opass.execute(ubo, depth, normal, want_mask=True)
    _check_frame("material matrix", ctx, rpass, opass, ref_full, ref_mask)
    ctx.close()


def test_debug_taps_9_to_12(torch_mod, oracle_mod):
    from vulkanrenderer_amd import make_test_scene, camera_path
    from vulkanrenderer_amd.renderer import PassExecuteContext
    scene = make_test_scene()
    W, H = 96, 54
    for mode in (9, 10, 11, 12):
        O, osc, ctx, rt, gb, rpass, opass, fs = _setup(scene, W, H, 4, 2, torch_mod, debug_view=mode)
        for frame, (eye, tgt) in enumerate(camera_path(scene, 2)):
            ubo = fs.update(eye, tgt)
            assert ubo[95] == float(mode)
            depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
            gb.upload(depth, normal, wp, ng, valid)
            rpass.execute(PassExecuteContext(ubo=ubo))
            ctx.sync()
            ref_full, ref_mask = opass.execute(ubo, depth, normal, want_mask=True)
            _check_frame(f"debug {mode} frame {frame}", ctx, rpass, opass, ref_full, None)
        ctx.close()


def test_separate_dispatches_equal_execute(torch_mod, oracle_mod):
    """RtaoComputePass::render = dispatchTrace + dispatchAtrous + dispatchUpsample (RtaoComputePass.cpp:503-571)."""
    from vulkanrenderer_amd import make_test_scene, camera_path
    from vulkanrenderer_amd.renderer import PassExecuteContext
    scene = make_test_scene()
    W, H = 96, 54
    for iters in (1, 2, 3):
        O, osc, ctx, rt, gb, rpass, opass, fs = _setup(scene, W, H, 4, iters, torch_mod)
        ctx2 = type(ctx)(W, H)
        rt2 = type(rt)(ctx2)
        rt2.init(scene)
        gb2 = type(gb)(ctx2)
        rpass2 = type(rpass)(ctx2, gb2, atrous_iterations=iters)
        for frame, (eye, tgt) in enumerate(camera_path(scene, 3)):
            ubo = fs.update(eye, tgt)
            depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
            gb.upload(depth, normal)
            gb2.upload(depth, normal)
            rpass.execute(PassExecuteContext(ubo=ubo))
            rpass2.dispatchTrace(ubo)
            rpass2.dispatchAtrous(ubo)
            rpass2.dispatchUpsample(ubo)
            rpass2.endFrame()
            ctx.sync(); ctx2.sync()
            assert np.array_equal(to_np_u16(rpass.rtao_full), to_np_u16(rpass2.rtao_full))
            assert np.array_equal(to_np_u16(rpass.image(0)), to_np_u16(rpass2.image(0)))
            # the final ping-pong image is the one index (iterations-1)%2 names (RtaoComputePass.cpp:560-561)
            assert np.array_equal(to_np_u16(rpass.image(2 + (iters - 1) % 2)), to_np_u16(rpass2.image(2 + (iters - 1) % 2)))
        ctx.close(); ctx2.close()


def test_golden_vectors(torch_mod):
    """HIP path against the committed fixture alone (tests/golden/rtao_64x36.npz): no oracle in the loop."""
    import os
    from vulkanrenderer_amd import make_street_scene
    from vulkanrenderer_amd.renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass,
                                             ShadowRays)
    from vulkanrenderer_amd.scenes import FrameState
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rtao_64x36.npz"))
    seed, W, H, rays, iters, frames = (int(x) for x in g["meta"])
    scene = make_street_scene(seed, float(g["detail"][0]), "golden")
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    gb = GBuffer(ctx)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=iters, want_hit_mask=True)
    sh = ShadowRays(ctx, gb)
    tap_ctx = Context(W, H)
    tap_rt = RayTracingContext(tap_ctx)
    tap_rt.init(scene)
    tap_gb = GBuffer(tap_ctx)
    tap_pass = RtaoComputePass(tap_ctx, tap_gb, atrous_iterations=iters)
    for f in range(frames):
        ubo = g[f"ubo{f}"]
        gb.upload(g[f"depth{f}"], g[f"normal{f}"], g[f"world_pos{f}"], g[f"geom_normal{f}"], g[f"valid{f}"])
        rpass.execute(PassExecuteContext(ubo=ubo))
        ctx.sync()
        assert np.array_equal(to_np_u16(rpass.rtao_full), g[f"rtao_full{f}"])
        assert np.array_equal(to_np_u16(rpass.image(0)), g[f"history{f}"])
        assert np.array_equal(rpass.hit_mask.cpu().numpy().view(np.uint32), g[f"hit_mask{f}"])
        fs_ubo = ubo.copy()
        fs_ubo[56:58] = (0.01, 8.0)
        sun = sh.sun(fs_ubo).cpu().numpy()
        assert np.array_equal(sun.view(np.uint32), g[f"sun{f}"].view(np.uint32))
        pt = sh.point(ubo, 0).cpu().numpy()
        assert np.array_equal(pt.view(np.uint32), g[f"point{f}"].view(np.uint32))
        tap_gb.upload(g[f"depth{f}"], g[f"normal{f}"])
        for mode in (10, 11, 12):
            u = ubo.copy()
            u[95] = float(mode)
            tap_pass.reset()
            tap_pass.dispatchTrace(u)
            tap_pass.endFrame()
            tap_ctx.sync()
            assert np.array_equal(to_np_u16(tap_pass.image(0)), g[f"tap{mode}_{f}"]), f"tap {mode} frame {f}"
    ctx.close(); tap_ctx.close()


def test_determinism_and_reset(torch_mod):
    from vulkanrenderer_amd import make_test_scene, camera_path, FrameState
    from vulkanrenderer_amd.renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass,
                                             generate_gbuffer)
    scene = make_test_scene()
    W, H = 200, 120
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    gb = GBuffer(ctx)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=4)
    runs = []
    for rep in range(2):
        fs = FrameState(W, H, ray_count=8)
        rpass.reset()
        outs = []
        for eye, tgt in camera_path(scene, 3):
            ubo = fs.update(eye, tgt)
            generate_gbuffer(ctx, ubo, gb)
            rpass.execute(PassExecuteContext(ubo=ubo))
            ctx.sync()
            outs.append(to_np_u16(rpass.rtao_full).copy())
        runs.append(outs)
        if rep == 0:
            rt.rebuild()          # a rebuilt BVH must not change a single bit either
    for a, b in zip(*runs):
        assert np.array_equal(a, b)
    ao = runs[0][-1].view(np.float16).astype(np.float32)
    assert ao.min() >= 0.0 and ao.max() <= 1.0 and ao.min() < 0.9
    ctx.close()


def test_all_sky_and_empty_scene(torch_mod, oracle_mod):
    from vulkanrenderer_amd import FrameState, make_test_scene
    from vulkanrenderer_amd.scenes import SceneData
    from vulkanrenderer_amd.renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass,
                                             ShadowRays, generate_gbuffer)
    W, H = 64, 40
    # (a) camera looking at the sky: every pixel invalid -> AO 1.0 everywhere, hit mask 0
    scene = make_test_scene()
    O, osc, ctx, rt, gb, rpass, opass, fs = _setup(scene, W, H, 4, 2, torch_mod)
    ubo = fs.update((0.0, 50.0, 0.0), (0.0, 150.0, 0.1))
    generate_gbuffer(ctx, ubo, gb)
    rpass.execute(PassExecuteContext(ubo=ubo))
    ctx.sync()
    assert int(gb.valid.sum().item()) == 0
    assert np.all(to_np_u16(rpass.rtao_full) == ONE16) and int(rpass.hit_mask.abs().sum().item()) == 0
    ctx.close()
    # (b) empty scene: builds, G-buffer is all sky, rays all miss, shadows fully lit
    empty = SceneData([], [], np.zeros(0, dtype=np.uint32), np.zeros((0, 12), dtype=np.float32), "empty")
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(empty)
    info = rt.info()
    assert info.triangle_count == 0 and info.node_count == 0
    gb = GBuffer(ctx)
    fs = FrameState(W, H, ray_count=4)
    ubo = fs.update((0, 1, 0), (0, 1, 5))
    generate_gbuffer(ctx, ubo, gb)
    assert int(gb.valid.sum().item()) == 0
    # a synthetic floor G-buffer over the empty BVH: all AO rays miss -> visibility 1
    oscf = O.OracleScene(make_test_scene())
    depth, normal, wp, ng, valid = oscf.gbuffer(ubo, W, H)
    gb.upload(depth, normal, wp, ng, valid)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=2, want_hit_mask=True)
    rpass.execute(PassExecuteContext(ubo=ubo))
    ctx.sync()
    assert np.all(to_np_u16(rpass.rtao_full) == ONE16) and int(rpass.hit_mask.abs().sum().item()) == 0
    sh = ShadowRays(ctx, gb)
    assert float(sh.sun(ubo).min().item()) == 1.0 and float(sh.point(ubo, 0).min().item()) == 1.0
    oe = O.OracleScene(empty)
    ref = O.OracleRtaoPass(oe, W, H, 2).execute(ubo, depth, normal)
    assert np.array_equal(ref, to_np_u16(rpass.rtao_full))
    rt.updateTopLevelAS(empty.instance_mesh, empty.instance_transform)     # refit of nothing is a no-op
    # rays straight through the root's origin along (1, 1, 1): the three axis crossings coincide there, the one case in which
    # inverted planes alone would not make the empty root's slots miss (the kernels never walk a tree of zero triangles)
    import torch
    wp = torch.zeros((H, W, 3), dtype=torch.float32, device=ctx.device)
    wp[:] = torch.tensor([-2.0, -2.0, -2.0])
    gb.world_pos.copy_(wp)
    gb.geom_normal.zero_()
    gb.valid.fill_(1)
    u2 = ubo.copy()
    u2[48:51] = (-1.0, -1.0, -1.0)                                         # directionalLightDir: L = normalize(1, 1, 1)
    u2[60:63] = (2.0, 2.0, 2.0)                                            # light 0 on the same diagonal
    assert float(sh.sun(u2).min().item()) == 1.0 and float(sh.point(u2, 0).min().item()) == 1.0
    ctx.close()


def test_single_triangle_scene(torch_mod, oracle_mod):
    from vulkanrenderer_amd.scenes import Material, Mesh, SceneData, _make_vertices, trs
    from vulkanrenderer_amd.renderer import PassExecuteContext
    pos = np.array([[-5, 0, -5], [5, 0, -5], [0, 0, 8]], dtype=np.float32)
    floor = Mesh(_make_vertices(pos, np.zeros((3, 2), dtype=np.float32)), np.array([0, 2, 1], dtype=np.uint32), 0)
    scene = SceneData([floor], [Material()], np.array([0], dtype=np.uint32), np.stack([trs()]).astype(np.float32), "tri")
    W, H = 48, 32
    O, osc, ctx, rt, gb, rpass, opass, fs = _setup(scene, W, H, 4, 2, torch_mod)
    assert rt.info().triangle_count == 1
    ubo = fs.update((0.0, 2.0, -3.0), (0.0, 0.0, 2.0))
    depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
    assert valid.any()
    gb.upload(depth, normal, wp, ng, valid)
    rpass.execute(PassExecuteContext(ubo=ubo))
    ctx.sync()
    ref_full, ref_mask = opass.execute(ubo, depth, normal, want_mask=True)
    _check_frame("single triangle", ctx, rpass, opass, ref_full, ref_mask)
    ctx.close()


def test_gbuffer_primary_matches_oracle(torch_mod, oracle_mod):
    from vulkanrenderer_amd import make_street_scene, camera_path, FrameState
    from vulkanrenderer_amd.renderer import Context, GBuffer, RayTracingContext, generate_gbuffer
    from oracle import oracle as O
    scene = make_street_scene(11, 0.03, "gb")
    W, H = 192, 108
    osc = O.OracleScene(scene)
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    gb = GBuffer(ctx)
    fs = FrameState(W, H)
    for eye, tgt in camera_path(scene, 2, step=3.0):
        ubo = fs.update(eye, tgt)
        generate_gbuffer(ctx, ubo, gb)
        ctx.sync()
        depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
        assert np.array_equal(gb.valid.cpu().numpy(), valid)
        # closest-hit ties between coplanar/abutting triangles may resolve to a different primitive: report them
        dm = (to_np_u16(gb.linear_depth) != depth).mean()
        nm = (to_np_u16(gb.normal) != normal).any(axis=-1).mean()
        print(f"gbuffer mismatches: depth {dm:.5f} normal {nm:.5f}")
        assert dm == 0.0 and nm <= 0.001
        assert np.abs(gb.world_pos.cpu().numpy() - wp).max() <= 1e-4
    ctx.close()


def test_gbuffer_vertex_normals_scaled_and_mirrored_instances(torch_mod, oracle_mod):
    """Interpolated vertex normals through transpose(inverse(mat3(model))) incl. non-uniform scale and a mirrored
    instance (depth_prepass.vert:47-48), Ng flipped by facing (pbr.frag:339-346): bit-exact against the oracle."""
    from vulkanrenderer_amd import FrameState
    from vulkanrenderer_amd.renderer import Context, GBuffer, RayTracingContext, generate_gbuffer
    from vulkanrenderer_amd.scenes import Material, SceneData, box_mesh, cylinder_mesh, sphere_mesh, trs
    from oracle import oracle as O
    meshes = [sphere_mesh(20, 12, 0), cylinder_mesh(16, 3, 0), box_mesh(2, 0)]
    xf = [trs((-2.5, 0.0, 6.0), (2.0, 1.0, 1.5), 0.4), trs((0.0, 0.0, 7.0), (-1.5, 2.0, 1.0), 1.1),
          trs((2.5, 0.2, 6.0), (1.0, 1.0, 1.0), 0.7), trs((-6.0, -0.2, 0.0), (12.0, 0.2, 12.0)),
          trs((0.5, 0.5, 4.5), (0.6, -0.8, 0.6), 2.0)]
    scene = SceneData(meshes, [Material()], np.array([0, 1, 0, 2, 1], dtype=np.uint32), np.stack(xf).astype(np.float32), "n")
    W, H = 160, 96
    osc = O.OracleScene(scene)
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    gb = GBuffer(ctx)
    ubo = FrameState(W, H).update((0.0, 1.6, 0.0), (0.0, 0.6, 6.0))
    generate_gbuffer(ctx, ubo, gb)
    ctx.sync()
    depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
    assert valid.mean() > 0.3
    assert np.array_equal(gb.valid.cpu().numpy(), valid)
    assert np.array_equal(to_np_u16(gb.linear_depth), depth)
    assert np.array_equal(to_np_u16(gb.normal), normal)
    assert np.array_equal(gb.geom_normal.cpu().numpy().view(np.uint32), ng.view(np.uint32))
    # the encoded normals are smooth on the curved meshes: far more distinct values than triangles' face normals would give
    n = normal.view(np.float16)[..., :3][valid > 0]
    assert len(np.unique(n.view(np.uint16), axis=0)) > 1000
    # refit keeps the producer in step with moved instances
    xf2 = np.stack(xf).astype(np.float32)
    xf2[:, 3] += 0.3
    rt.updateTopLevelAS(scene.instance_mesh, xf2)
    osc.set_instances(scene.instance_mesh, xf2)
    generate_gbuffer(ctx, ubo, gb)
    ctx.sync()
    depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
    assert np.array_equal(to_np_u16(gb.normal), normal)
    ctx.close()


def test_self_hit_retry_stress_coincident_layers(torch_mod, oracle_mod):
    """G-buffer of the plain scene, rays traced against 13 copies of it stacked 0.4 mm apart (the origin rebuilt from
    R16F depth is off the surface by more than that): a large share of first hits land within 1.5 x tMin of the ray
    origin, so the traceRayVisibility self-hit retry
    (rtao_trace_half.comp:186-193) runs for many rays.  K1 does that retry in place while other rays of the item are
    still in flight; the result must stay bit-identical to the oracle's sequential two-query form."""
    import dataclasses
    from vulkanrenderer_amd import make_test_scene, camera_path
    from vulkanrenderer_amd.renderer import PassExecuteContext, trace_stats, trace_stats_enable
    from oracle import oracle as O_
    base = make_test_scene()
    xf = base.instance_transform
    layers = []
    for k in range(-6, 7):
        c = xf.copy()
        c[:, 3] += k * 4e-4; c[:, 7] += k * 4e-4; c[:, 11] += k * 4e-4
        layers.append(c)
    scene = dataclasses.replace(base, instance_mesh=np.concatenate([base.instance_mesh] * len(layers)),
                                instance_transform=np.concatenate(layers), name="coincident")
    gb_scene = O_.OracleScene(base)
    W, H, rays = 192, 108, 8
    O, osc, ctx, rt, gb, rpass, opass, fs = _setup(scene, W, H, rays, 2, torch_mod)
    for frame, (eye, tgt) in enumerate(camera_path(base, 3)):
        ubo = fs.update(eye, tgt)
        depth, normal, wp, ng, valid = gb_scene.gbuffer(ubo, W, H)
        gb.upload(depth, normal, wp, ng, valid)
        trace_stats_enable(ctx, frame == 0)       # frame 0 through the counting build of K1, the rest through the product one
        trace_stats(ctx, reset=True)
        rpass.execute(PassExecuteContext(ubo=ubo))
        ctx.sync()
        if frame == 0:
            st = trace_stats(ctx, reset=True)
            nvalid = int(np.asarray(valid).astype(bool).sum())
            retries = st.rays - rays * nvalid
            print(f"coincident: {nvalid} px, {st.rays} queries, {retries} retries ({retries / max(1, rays * nvalid):.1%})")
            assert retries >= rays * nvalid // 10, "the scene is meant to make the retry common"
        ref_full, ref_mask = opass.execute(ubo, depth, normal, want_mask=True)
        _check_frame(f"coincident frame {frame}", ctx, rpass, opass, ref_full, ref_mask)
    ctx.close()
