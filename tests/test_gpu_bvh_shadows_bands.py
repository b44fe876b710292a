"""GPU tests of the LBVH (structure, refit), shadow rays, row bands and error behaviour, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from helpers import compare_r16f, to_np_u16

pytestmark = pytest.mark.gpu

CHILD_NONE = 0x7fffffff


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _ctx(scene, W, H):
    from vulkanrenderer_amd.renderer import Context, GBuffer, RayTracingContext
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    return ctx, rt, GBuffer(ctx)


def _world_tris(scene):
    """Reference flatten: instance order, mesh triangle order (RayTracingContext.cpp:46-71 custom index = mesh)."""
    out = []
    for mi, xf in zip(scene.instance_mesh, scene.instance_transform):
        m = scene.meshes[int(mi)]
        M = np.asarray(xf, dtype=np.float64).reshape(3, 4)
        p = m.vertices[:, :3].astype(np.float64) @ M[:, :3].T + M[:, 3]
        out.append(p[m.indices.reshape(-1, 3)])
    return np.concatenate(out) if out else np.zeros((0, 3, 3))


def _check_bvh_structure(nodes_f, tris_f, scene, info):
    """Every triangle in exactly one leaf, every node reachable exactly once, every slot box encloses its subtree."""
    T = info.triangle_count
    nodes_u = nodes_f.view(np.uint32)
    tris_u = tris_f.view(np.uint32)
    flat = tris_u[:, 11]
    assert np.array_equal(np.sort(flat), np.arange(T, dtype=np.uint32)), "flat ids are a permutation"
    v0 = tris_f[:, 0:3].astype(np.float64)
    v1 = v0 + tris_f[:, 4:7].astype(np.float64)
    v2 = v0 + tris_f[:, 8:11].astype(np.float64)
    # records hold the world-space triangle the reference flatten gives (float rounding of the transform aside)
    wt = _world_tris(scene)[flat]
    scale = max(1.0, np.abs(wt).max())
    assert np.abs(v0 - wt[:, 0]).max() <= 1e-5 * scale and np.abs(v1 - wt[:, 1]).max() <= 2e-5 * scale
    tmin = np.minimum(np.minimum(v0, v1), v2)
    tmax = np.maximum(np.maximum(v0, v1), v2)
    seen_tri = np.zeros(T, dtype=np.int32)
    seen_node = np.zeros(info.node_count, dtype=np.int32)
    depth_max = 0

    def subtree(ref):
        """-> (min, max) true bounds of the triangles under ref"""
        nonlocal depth_max
        if ref < 0:
            x = ref & 0x7fffffff                      # leaf: bit 31 set, bits 30..1 first triangle, bit 0 count - 1
            first, cnt = x >> 1, (x & 1) + 1
            assert first + cnt <= T
            seen_tri[first:first + cnt] += 1
            return tmin[first:first + cnt].min(axis=0), tmax[first:first + cnt].max(axis=0)
        seen_node[ref] += 1
        n = nodes_u[ref]
        org = nodes_f[ref, 0:3].astype(np.float64)
        sc = np.array([nodes_f[ref, 10], nodes_f[ref, 11], nodes_f[ref, 3]], dtype=np.float64)   # scale x, y | z: powers of two
        assert np.all(sc > 0) and np.all(np.log2(sc) == np.round(np.log2(sc)))
        qmin = np.array([[(int(n[4 + a]) >> (8 * c)) & 0xff for a in range(3)] for c in range(4)], dtype=np.float64)
        qmax = np.array([[(int(n[(7, 8, 9)[a]]) >> (8 * c)) & 0xff for a in range(3)] for c in range(4)], dtype=np.float64)
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        used = 0
        for c in range(4):
            r = int(np.int32(n[12 + c]))
            if r == CHILD_NONE:
                assert np.all(qmin[c] == 255) and np.all(qmax[c] == 0), "unused slots are stored inverted"
                continue
            used += 1
            lo, hi = subtree(r)
            bmin, bmax = org + qmin[c] * sc, org + qmax[c] * sc
            assert np.all(bmin <= lo) and np.all(bmax >= hi), f"node {ref} slot {c} does not enclose its subtree"
            lo_all, hi_all = np.minimum(lo_all, lo), np.maximum(hi_all, hi)
        assert used >= 1
        return lo_all, hi_all

    import sys
    sys.setrecursionlimit(10000)
    lo, hi = subtree(0)
    assert np.all(seen_tri == 1) and np.all(seen_node == 1)
    assert np.all(lo >= np.array(info.bounds_min[:]) - 1e-4) and np.all(hi <= np.array(info.bounds_max[:]) + 1e-4)


def test_bvh_structure_after_build_and_refit(torch_mod):
    from vulkanrenderer_amd import make_street_scene
    scene = make_street_scene(5, 0.02, "bvh")
    ctx, rt, gb = _ctx(scene, 32, 32)
    info = rt.info()
    assert info.triangle_count == scene.triangle_count() and info.morton_bits in (10, 21)
    assert 0 < info.node_count <= info.triangle_count and info.height >= 2 and info.binary_height >= info.height
    nodes, tris = rt.download()
    assert nodes.shape == (info.node_count, 16) and tris.shape == (info.triangle_count, 12)
    _check_bvh_structure(nodes, tris, scene, info)
    # refit: same topology (child refs unchanged), boxes re-fitted to the moved instances
    xf = scene.instance_transform.copy()
    rng = np.random.default_rng(3)
    xf[:, 3] += rng.uniform(-1.5, 1.5, len(xf)).astype(np.float32)
    xf[:, 7] += rng.uniform(0.0, 0.8, len(xf)).astype(np.float32)
    rt.updateTopLevelAS(scene.instance_mesh, xf)
    ctx.sync()
    nodes2, tris2 = rt.download()
    assert np.array_equal(nodes.view(np.uint32)[:, 12:16], nodes2.view(np.uint32)[:, 12:16])
    assert np.array_equal(tris.view(np.uint32)[:, 11], tris2.view(np.uint32)[:, 11])
    moved = type(scene)(scene.meshes, scene.materials, scene.instance_mesh, xf, "moved")
    _check_bvh_structure(nodes2, tris2, moved, rt.info())
    ctx.close()


@pytest.mark.parametrize("scene_kind", ["street", "tiny", "single"])
@pytest.mark.parametrize("mode", ["wide", "binary"])
def test_refit_reproduces_the_build_boxes(torch_mod, scene_kind, mode, monkeypatch):
    """vkr_bvh_refit works level by level on the wide tree (k_wide_refit_level; VKR_REFIT=binary: binary fit +
    regather).  min/max are exact, so a refit with the build's own transforms must reproduce the build's nodes bit for
    bit (the build fitted the binary tree), a refit to moved instances must pass the structure check, and moving back
    must give the build's nodes again."""
    from vulkanrenderer_amd import make_street_scene
    from vulkanrenderer_amd.scenes import Material, Mesh, SceneData, grid_mesh, trs
    monkeypatch.setenv("VKR_REFIT", mode)
    if scene_kind == "street":
        scene = make_street_scene(5, 0.02, "bvh")
    elif scene_kind == "tiny":
        scene = SceneData([grid_mesh(1, 2, 0)], [Material()], np.zeros(2, dtype=np.uint32),
                          np.stack([trs((0, 0, 3), (4, 1, 4)), trs((0.5, 0.7, 3.5), (1, 1, 1))]).astype(np.float32), "tiny")
    else:
        g = grid_mesh(1, 1, 0)
        one = Mesh(g.vertices, g.indices[:3].copy(), 0)
        scene = SceneData([one], [Material()], np.zeros(1, dtype=np.uint32), np.stack([trs((0, 0, 3), (1, 1, 1))]).astype(np.float32), "single")
    ctx, rt, gb = _ctx(scene, 32, 32)
    nodes0, tris0 = rt.download()
    rt.updateTopLevelAS(scene.instance_mesh, scene.instance_transform)
    ctx.sync()
    nodes1, tris1 = rt.download()
    assert nodes1.tobytes() == nodes0.tobytes() and tris1.tobytes() == tris0.tobytes()
    rng = np.random.default_rng(11)
    xf = scene.instance_transform.copy()
    xf[:, 3] += rng.uniform(-1.0, 1.0, len(xf)).astype(np.float32)
    xf[:, 7] += rng.uniform(0.0, 0.5, len(xf)).astype(np.float32)
    rt.updateTopLevelAS(scene.instance_mesh, xf)
    ctx.sync()
    nodes2, tris2 = rt.download()
    assert np.array_equal(nodes0.view(np.uint32)[:, 12:16], nodes2.view(np.uint32)[:, 12:16])
    if scene_kind != "single":
        assert nodes2.tobytes() != nodes0.tobytes()
    moved = type(scene)(scene.meshes, scene.materials, scene.instance_mesh, xf, "moved")
    _check_bvh_structure(nodes2, tris2, moved, rt.info())
    rt.updateTopLevelAS(scene.instance_mesh, scene.instance_transform)
    ctx.sync()
    nodes3, _ = rt.download()
    assert nodes3.tobytes() == nodes0.tobytes()
    ctx.close()


@pytest.mark.parametrize("kind", ["tiny", "duplicates", "coplanar_grid", "street", "treelets_off", "many_duplicates", "line",
                                  "sah_off"])
def test_treelet_rebuild_keeps_the_tree_well_formed(torch_mod, oracle_mod, kind, monkeypatch):
    """k_treelet_ploc under VKR_VALIDATE_TREE=1 (host-side check of the binary tree before the fit walks it) on the
    inputs that stress it: fewer triangles than a treelet, hundreds of identical triangles (every union area ties),
    a regular coplanar grid (ties between different pairs), a normal scene; then the wide tree's structure and one
    traced frame against the oracle.  VKR_TREELETS=0 (plain Karras tree) must give the same pixels.  The binned-SAH top
    levels (k_sah_*, ranges above 1024 triangles) see: the grid and the street, thousands of coincident centroids (no
    split exists: the range is left to the Morton bits), centroids on one line (two degenerate axes), and
    VKR_SAH_TOP=0."""
    from vulkanrenderer_amd import FrameState, make_street_scene
    from vulkanrenderer_amd.renderer import PassExecuteContext, RtaoComputePass
    from vulkanrenderer_amd.scenes import Material, Mesh, SceneData, grid_mesh, trs, VERTEX_FLOATS
    monkeypatch.setenv("VKR_VALIDATE_TREE", "1")
    if kind == "treelets_off":
        monkeypatch.setenv("VKR_TREELETS", "0")
    if kind == "sah_off":
        monkeypatch.setenv("VKR_SAH_TOP", "0")
    if kind == "tiny":
        scene = SceneData([grid_mesh(1, 2, 0)], [Material()], np.zeros(2, dtype=np.uint32),
                          np.stack([trs((0, 0, 3), (4, 1, 4)), trs((0.5, 0.7, 3.5), (1, 1, 1))]).astype(np.float32), "tiny")   # 8 triangles
    elif kind == "duplicates":
        g = grid_mesh(1, 1, 0)
        xf = np.stack([trs((0, 0, 3), (4, 1, 4))] * 150 + [trs((0.5, 0.6, 3.5), (1, 1, 1))] * 3).astype(np.float32)
        scene = SceneData([g], [Material()], np.zeros(len(xf), dtype=np.uint32), xf, "dups")                                  # 306 triangles
    elif kind == "coplanar_grid":
        scene = SceneData([grid_mesh(40, 40, 0)], [Material()], np.zeros(2, dtype=np.uint32),
                          np.stack([trs((-4, 0, 1), (8, 1, 8)), trs((-1, 0.8, 3), (2, 1, 2))]).astype(np.float32), "grid")     # 6400 triangles
    elif kind == "many_duplicates":
        g = grid_mesh(1, 1, 0)
        xf = np.stack([trs((0, 0, 3), (4, 1, 4))] * 1300 + [trs((0.5, 0.6, 3.5), (1, 1, 1))] * 40 +
                      [trs((-2 + 0.01 * i, 0.3, 2.0), (0.5, 1, 0.5)) for i in range(300)]).astype(np.float32)
        scene = SceneData([g], [Material()], np.zeros(len(xf), dtype=np.uint32), xf, "manydups")                              # 3280 triangles
    elif kind == "line":
        g = grid_mesh(1, 1, 0)
        xf = np.stack([trs((-4 + 8.0 * i / 1499, 0.2, 3.0), (0.05, 1, 0.05)) for i in range(1500)] +
                      [trs((0, 0, 3), (4, 1, 4))]).astype(np.float32)
        scene = SceneData([g], [Material()], np.zeros(len(xf), dtype=np.uint32), xf, "line")                                  # 3002 triangles
    else:
        scene = make_street_scene(5, 0.02, "bvh")
    scene.cameras = scene.cameras or [((0.3, 2.5, -1.0), (0.2, 0.0, 4.0))]
    W, H = 64, 40
    ctx, rt, gb = _ctx(scene, W, H)
    info = rt.info()
    assert info.triangle_count == scene.triangle_count()
    nodes, tris = rt.download()
    _check_bvh_structure(nodes, tris, scene, info)
    osc = oracle_mod.OracleScene(scene)
    rp = RtaoComputePass(ctx, gb, atrous_iterations=2)
    op = oracle_mod.OracleRtaoPass(osc, W, H, 2)
    ubo = FrameState(W, H, ray_count=4).update(*scene.cameras[0])
    depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
    assert (valid > 0).mean() > 0.05
    gb.upload(depth, normal, wp, ng, valid)
    rp.execute(PassExecuteContext(ubo=ubo))
    ctx.sync()
    assert to_np_u16(rp.rtao_full).tobytes() == op.execute(ubo, depth, normal).tobytes()
    ctx.close()


def test_refit_matches_oracle_and_rebuild(torch_mod, oracle_mod):
    """updateTopLevelAS (RayTracingContext.cpp:163-197): traced results after a refit equal the oracle's on the moved
    instances and equal a from-scratch build's (results do not depend on the tree)."""
    from vulkanrenderer_amd import FrameState, make_street_scene
    from vulkanrenderer_amd.renderer import PassExecuteContext, RtaoComputePass
    O = oracle_mod
    scene = make_street_scene(9, 0.02, "refit")
    W, H = 128, 72
    ctx, rt, gb = _ctx(scene, W, H)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=2, want_hit_mask=True)
    osc = O.OracleScene(scene)
    fs = FrameState(W, H, ray_count=4)
    rng = np.random.default_rng(17)
    xf = scene.instance_transform.copy()
    eye, tgt = scene.cameras[0]
    for step in range(3):
        xf = xf.copy()
        xf[:, 3] += rng.uniform(-0.4, 0.4, len(xf)).astype(np.float32)
        xf[:, 11] += rng.uniform(-0.4, 0.4, len(xf)).astype(np.float32)
        rt.updateTopLevelAS(scene.instance_mesh, xf)
        osc.set_instances(scene.instance_mesh, xf)
        ubo = fs.update(eye, tgt)
        depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
        gb.upload(depth, normal, wp, ng, valid)
        rpass.reset()
        rpass.execute(PassExecuteContext(ubo=ubo))
        ctx.sync()
        ref, ref_mask = O.OracleRtaoPass(osc, W, H, 2).execute(ubo, depth, normal, want_mask=True)
        got = to_np_u16(rpass.rtao_full).copy()
        got_mask = rpass.hit_mask.cpu().numpy().view(np.uint32).copy()
        assert int(np.unpackbits((got_mask ^ ref_mask).view(np.uint8)).sum()) == 0
        assert np.array_equal(got, ref)
    # from-scratch build on the final transforms
    moved = type(scene)(scene.meshes, scene.materials, scene.instance_mesh, xf, "moved")
    ctx2, rt2, gb2 = _ctx(moved, W, H)
    gb2.upload(depth, normal, wp, ng, valid)
    rpass2 = RtaoComputePass(ctx2, gb2, atrous_iterations=2)
    rpass2.execute(PassExecuteContext(ubo=ubo))
    ctx2.sync()
    assert np.array_equal(to_np_u16(rpass2.rtao_full), got)
    ctx.close(); ctx2.close()


def test_sun_and_point_shadows_match_oracle(torch_mod, oracle_mod):
    from vulkanrenderer_amd import FrameState, make_street_scene, camera_path
    from vulkanrenderer_amd.renderer import ShadowRays
    O = oracle_mod
    scene = make_street_scene(13, 0.02, "shadow")
    W, H = 160, 90
    ctx, rt, gb = _ctx(scene, W, H)
    osc = O.OracleScene(scene)
    sh = ShadowRays(ctx, gb)
    for radius, samples in [(0.01, 8), (0.05, 16), (0.0, 8), (0.01, 1), (0.02, 40)]:
        fs = FrameState(W, H, sun_radius=radius, sun_samples=samples, point_lights=3)
        eye, tgt = camera_path(scene, 1)[0]
        ubo = fs.update(eye, tgt)
        depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
        gb.upload(depth, normal, wp, ng, valid)
        got = sh.sun(ubo).cpu().numpy()
        ref = osc.shadow_sun(wp, ng, valid, FrameState.sun_dir_to_light(ubo), radius, samples)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (radius, samples)
        v = valid > 0
        if radius <= 0.0 or samples <= 1:     # hard shadow keeps the 0.2 floor (pbr.frag:232-235)
            assert set(np.unique(got[v])) <= {np.float32(0.2), np.float32(1.0)} and (got[v] == np.float32(0.2)).any()
        else:
            n = min(16, samples)
            assert np.allclose(got[v] * n, np.round(got[v] * n), atol=1e-5) and got[v].min() == 0.0
        assert np.all(got[~v] == 1.0)
    for light in range(3):
        got = sh.point(ubo, light).cpu().numpy()
        ref = osc.shadow_point(wp, ng, valid, ubo[60 + 4 * light:63 + 4 * light])
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), light
        assert set(np.unique(got)) <= {np.float32(0.0), np.float32(1.0)}
    # a band of rows through the same entry point equals those rows of the full frame
    full = sh.sun(ubo).cpu().numpy().copy()
    sh.sun_vis.zero_()
    part = sh.sun(ubo, rows=(30, 61)).cpu().numpy()
    assert np.array_equal(part[30:61], full[30:61]) and not part[:30].any() and not part[61:].any()
    ctx.close()


@pytest.mark.parametrize("W,H", [(67, 45), (8, 8), (130, 7), (9, 70)])
def test_shadow_chunks_as_tiles_ragged_sizes_and_the_linear_fallback(torch_mod, oracle_mod, W, H):
    """The shadow kernels cut arrays that are rows of the context's image into 8x8 tiles (widths and heights that are no multiple
    of 8, a band that starts between tile rows), and anything else -- here: the frame without its last 5 pixels -- into runs of
    pixels; VKR_SUN_TILES=0 is the all-runs form.  Every pixel against the oracle, sun (packets) and point lights."""
    import ctypes as C
    from vulkanrenderer_amd import FrameState, make_street_scene, camera_path
    from vulkanrenderer_amd.renderer import ShadowRays
    from vulkanrenderer_amd.scenes import UBO_DIR_LIGHT_PARAMS
    O = oracle_mod
    scene = make_street_scene(13, 0.02, "shadow")
    ctx, rt, gb = _ctx(scene, W, H)
    osc = O.OracleScene(scene)
    sh = ShadowRays(ctx, gb)
    fs = FrameState(W, H, sun_radius=0.02, sun_samples=16, point_lights=1)
    ubo = fs.update(*camera_path(scene, 1)[0])
    depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
    gb.upload(depth, normal, wp, ng, valid)
    ref = osc.shadow_sun(wp, ng, valid, FrameState.sun_dir_to_light(ubo), 0.02, 16).view(np.uint32)
    assert np.array_equal(sh.sun(ubo).cpu().numpy().view(np.uint32), ref)
    refp = osc.shadow_point(wp, ng, valid, ubo[60:63]).view(np.uint32)
    assert np.array_equal(sh.point(ubo, 0).cpu().numpy().view(np.uint32), refp)
    if H > 12:
        sh.sun_vis.zero_()
        part = sh.sun(ubo, rows=(3, H - 4)).cpu().numpy().view(np.uint32)      # a band that starts and ends between tile rows
        assert np.array_equal(part[3:H - 4], ref[3:H - 4]) and not part[:3].any() and not part[H - 4:].any()
    if W * H > 5:
        # count is no multiple of the width: runs of pixels
        sh.sun_vis.fill_(7.0)
        L = (C.c_float * 3)(*[float(x) for x in FrameState.sun_dir_to_light(ubo)])
        rc = ctx.lib.vkr_shadow_sun(ctx.handle, gb.world_pos.data_ptr(), gb.geom_normal.data_ptr(), gb.valid.data_ptr(), W * H - 5, L,
                                    float(ubo[UBO_DIR_LIGHT_PARAMS]), float(ubo[UBO_DIR_LIGHT_PARAMS + 1]), sh.sun_vis.data_ptr(), ctx.stream())
        assert rc == 0
        ctx.sync()
        got = sh.sun_vis.cpu().numpy().reshape(-1)
        assert np.array_equal(got[:-5].view(np.uint32), ref.reshape(-1)[:-5]) and np.all(got[-5:] == 7.0)
    ctx.close()


def _band_vs_full(scene, W, H, rays, iters, n, frames, step=0.05):
    """N contexts on one GPU, each executing its band (+halo) and exchanging [history | rtao_full] exactly as
    bench.py does over RCCL (pack / gather / unpack), against one context running the full frame."""
    import torch
    from vulkanrenderer_amd import FrameState, bands, camera_path
    from vulkanrenderer_amd.renderer import PassExecuteContext, RtaoComputePass, generate_gbuffer
    full_ctx, full_rt, full_gb = _ctx(scene, W, H)
    full_pass = RtaoComputePass(full_ctx, full_gb, atrous_iterations=iters)
    ranks = []
    for r in range(n):
        ctx, rt, gb = _ctx(scene, W, H)
        p = RtaoComputePass(ctx, gb, atrous_iterations=iters)
        p.band = bands.band_rows(H, n, r)
        ranks.append((ctx, rt, gb, p))
    hb = bands.band_height(H, n)
    gathered = torch.zeros((n, bands.PLANES, hb, W), dtype=torch.int16, device=full_ctx.device)
    dummy = torch.zeros((H, W), dtype=torch.int16, device=full_ctx.device)
    fs = FrameState(W, H, ray_count=rays)
    for f, (eye, tgt) in enumerate(camera_path(scene, frames, step=step)):
        ubo = fs.update(eye, tgt)
        generate_gbuffer(full_ctx, ubo, full_gb)
        full_pass.execute(PassExecuteContext(ubo=ubo))
        full_ctx.sync()
        for r, (ctx, rt, gb, p) in enumerate(ranks):
            if p.band[0] >= p.band[1]:
                continue                                   # more ranks than rows: nothing to do
            gb.linear_depth.copy_(full_gb.linear_depth)
            gb.normal.copy_(full_gb.normal)
            p.execute(PassExecuteContext(ubo=ubo))
            ctx.sync()
            bands.pack_band([p.image(0), p.rtao_full, dummy], H, n, r, gathered[r])
        for r, (ctx, rt, gb, p) in enumerate(ranks):
            if p.band[0] >= p.band[1]:
                # idle ranks still need the parity flip + the gathered history to stay in step
                p.endFrame()
            bands.unpack_frame(gathered, [p.image(0), p.rtao_full, dummy.clone()], H, n)
            torch.cuda.synchronize()
            assert np.array_equal(to_np_u16(p.rtao_full), to_np_u16(full_pass.rtao_full)), f"frame {f} rank {r} rtao_full"
            assert np.array_equal(to_np_u16(p.image(0)), to_np_u16(full_pass.image(0))), f"frame {f} rank {r} history"
    for ctx, *_ in ranks:
        ctx.close()
    full_ctx.close()


@pytest.mark.parametrize("n,H,iters", [(2, 90, 2), (3, 91, 3), (4, 90, 4), (8, 90, 1), (8, 37, 5)])
def test_band_schedule_equals_full_frame(torch_mod, n, H, iters):
    from vulkanrenderer_amd import make_street_scene
    _band_vs_full(make_street_scene(21, 0.02, "band"), 160, H, 4, iters, n, 3, step=0.3)


def test_band_schedule_equals_full_frame_1080p_8_bands(torch_mod):
    """BASELINE.json full size (1920x1080, 8 rays/px, 4 A-Trous iterations) over 8 bands, sponza-class geometry:
    the size-independent property 'sharded == unsharded, bit for bit'."""
    from vulkanrenderer_amd import make_sponza_class
    _band_vs_full(make_sponza_class(), 1920, 1080, 8, 4, 8, 2)


def _stripes_vs_full(scene, W, H, rays, iters, n, frames, step=0.05, sun_samples=0):
    """The two-phase schedule bench.py runs for N > 1 (K1 on interleaved 8-row stripes -> gather history -> A-Trous on
    contiguous bands -> gather rtao_full), emulated with N contexts on one GPU, against one full-frame context."""
    import torch
    from vulkanrenderer_amd import FrameState, bands, camera_path
    from vulkanrenderer_amd.renderer import PassExecuteContext, RtaoComputePass, generate_gbuffer
    full_ctx, full_rt, full_gb = _ctx(scene, W, H)
    full_pass = RtaoComputePass(full_ctx, full_gb, atrous_iterations=iters)
    ranks = []
    for r in range(n):
        ctx, rt, gb = _ctx(scene, W, H)
        ranks.append((ctx, rt, gb, RtaoComputePass(ctx, gb, atrous_iterations=iters)))
    dev = full_ctx.device
    plan = bands.StripePlan(H, n, dev)
    hist_g = torch.zeros((n, plan.slot_rows, W), dtype=torch.int16, device=dev)
    hb = bands.band_height(H, n)
    out_g = torch.zeros((n, 1, hb, W), dtype=torch.int16, device=dev)
    fs = FrameState(W, H, ray_count=rays)
    for f, (eye, tgt) in enumerate(camera_path(scene, frames, step=step)):
        ubo = fs.update(eye, tgt)
        generate_gbuffer(full_ctx, ubo, full_gb)
        full_pass.execute(PassExecuteContext(ubo=ubo))
        full_ctx.sync()
        for r, (ctx, rt, gb, p) in enumerate(ranks):
            gb.linear_depth.copy_(full_gb.linear_depth)
            gb.normal.copy_(full_gb.normal)
            p.historyCurr().fill_(0x1234)                   # rows this rank does not trace must come from the gather
            p.traceStripes(ubo, r, n)
            plan.pack(p.historyCurr(), r, hist_g[r])
        for r, (ctx, rt, gb, p) in enumerate(ranks):
            plan.unpack(hist_g, p.historyCurr())
            y0, y1 = bands.band_rows(H, n, r)
            p.denoiseBand(ubo, y0, y1)
            bands.pack_band([p.rtao_full], H, n, r, out_g[r])
        for r, (ctx, rt, gb, p) in enumerate(ranks):
            bands.unpack_frame(out_g, [p.rtao_full], H, n)
            torch.cuda.synchronize()
            assert np.array_equal(to_np_u16(p.image(0)), to_np_u16(full_pass.image(0))), f"frame {f} rank {r} history"
            assert np.array_equal(to_np_u16(p.rtao_full), to_np_u16(full_pass.rtao_full)), f"frame {f} rank {r} rtao_full"
    for ctx, *_ in ranks:
        ctx.close()
    full_ctx.close()


@pytest.mark.parametrize("n,H,iters", [(2, 90, 2), (3, 91, 3), (4, 90, 4), (8, 90, 1), (8, 37, 5), (5, 8, 2)])
def test_stripe_schedule_equals_full_frame(torch_mod, n, H, iters):
    from vulkanrenderer_amd import make_street_scene
    _stripes_vs_full(make_street_scene(21, 0.02, "band"), 160, H, 4, iters, n, 3, step=0.3)


def test_stripe_schedule_equals_full_frame_1080p_8_ranks(torch_mod):
    from vulkanrenderer_amd import make_sponza_class
    _stripes_vs_full(make_sponza_class(), 1920, 1080, 8, 4, 8, 2)


def test_c4_full_size_sharded_equals_unsharded(torch_mod):
    """BASELINE configs[4] at full size (bistro-class, 3840x2160, 8 AO rays + 16 sun rays per pixel, 4 A-Trous
    iterations) over 8 ranks: the size-independent property 'sharded == unsharded, bit for bit' for history, rtao_full
    and the sun visibility, two frames (the second one reprojects the gathered history)."""
    import torch
    from vulkanrenderer_amd import FrameState, bands, camera_path, make_bistro_class
    from vulkanrenderer_amd.renderer import GBuffer, PassExecuteContext, RtaoComputePass, ShadowRays, generate_gbuffer
    scene = make_bistro_class()
    W, H, n = 3840, 2160, 8
    full_ctx, _, _ = _ctx(scene, W, H)
    full_gb = GBuffer(full_ctx, with_shading_inputs=True)
    full_pass = RtaoComputePass(full_ctx, full_gb, atrous_iterations=4)
    full_sh = ShadowRays(full_ctx, full_gb)
    # one sharded context reused for every rank (8 more 4K contexts would only cost memory): its history image is
    # reset to the previous gathered history before each rank's turn
    ctx, _, _ = _ctx(scene, W, H)
    p = RtaoComputePass(ctx, full_gb, atrous_iterations=4)
    sh = ShadowRays(ctx, full_gb)
    plan = bands.StripePlan(H, n, full_ctx.device)
    hist_g = torch.zeros((n, plan.slot_rows, W), dtype=torch.int16, device=full_ctx.device)
    fs = FrameState(W, H, ray_count=8, sun_samples=16)
    for f, (eye, tgt) in enumerate(camera_path(scene, 2)):
        ubo = fs.update(eye, tgt)
        generate_gbuffer(full_ctx, ubo, full_gb)
        full_pass.execute(PassExecuteContext(ubo=ubo))
        want_sun = full_sh.sun(ubo).clone()
        full_ctx.sync()
        want_hist, want_full = full_pass.image(0).clone(), full_pass.rtao_full.clone()
        for r in range(n):                                   # phase 1 of every rank
            p.historyCurr().fill_(0x1234)
            p.traceStripes(ubo, r, n)
            plan.pack(p.historyCurr(), r, hist_g[r])
        plan.unpack(hist_g, p.historyCurr())                 # all-gather #1
        assert torch.equal(p.historyCurr(), want_hist), f"frame {f}: gathered history"
        got_full = torch.zeros_like(want_full)
        got_sun = torch.full_like(want_sun, -1.0)
        for r in range(n):                                   # phase 2 of every rank (denoise_band ends the frame:
            y0, y1 = bands.band_rows(H, n, r)                #  undo the flip for all but the last rank)
            p.denoiseBand(ubo, y0, y1)
            got_full[y0:y1] = p.rtao_full[y0:y1]
            got_sun[y0:y1] = sh.sun(ubo, rows=(y0, y1))[y0:y1]
            if r != n - 1:
                p.endFrame()                                 # flips back: frame_parity ^= 1 twice
        torch.cuda.synchronize()
        assert torch.equal(got_full, want_full), f"frame {f}: rtao_full"
        assert torch.equal(got_sun, want_sun), f"frame {f}: sun"
        assert torch.equal(p.image(0), want_hist)
        ao = want_full.view(torch.float16).float()
        assert 0.3 < float(ao.mean()) < 1.0 and float(ao.min()) < 0.5
    ctx.close()
    full_ctx.close()


def test_sun_shadows_small_launch_chunks(torch_mod, oracle_mod):
    """Row-band sized sun-shadow launches pick 64/128-pixel chunks (k_shadow_sun_pkt): same bits as the oracle."""
    import torch
    from vulkanrenderer_amd import FrameState, make_street_scene
    from vulkanrenderer_amd.renderer import GBuffer, ShadowRays, generate_gbuffer
    scene = make_street_scene(21, 0.02, "band")
    W, H = 256, 150
    ctx, rt, gb0 = _ctx(scene, W, H)
    gb = GBuffer(ctx, with_shading_inputs=True)
    osc = oracle_mod.OracleScene(scene)
    fs = FrameState(W, H, ray_count=4, sun_samples=8)
    ubo = fs.update(*scene.cameras[0])
    generate_gbuffer(ctx, ubo, gb)
    ctx.sync()
    wp, ng, valid = gb.world_pos.cpu().numpy(), gb.geom_normal.cpu().numpy(), gb.valid.cpu().numpy()
    want = osc.shadow_sun(wp, ng, valid, FrameState.sun_dir_to_light(ubo), float(ubo[56]), int(ubo[57]))
    sh = ShadowRays(ctx, gb)
    for rows in [(0, H), (10, 11), (40, 97), (H - 3, H)]:      # 38400, 256, 14592, 768 pixels
        sh.sun_vis.fill_(-1.0)
        got = sh.sun(ubo, rows=rows).cpu().numpy()
        assert got[rows[0]:rows[1]].tobytes() == want[rows[0]:rows[1]].tobytes(), rows
        assert (got[:rows[0]] == -1.0).all() and (got[rows[1]:] == -1.0).all()
    ctx.close()


def test_sun_shadows_overlap_an_rtao_frame_on_a_second_stream(torch_mod):
    """vkr_shadow_sun shares no scratch with the vkr_rtao_* calls (own chunk counter): issued on a second stream while an
    RTAO frame of the same context runs, both produce the bits of the sequential calls -- at a size where both kernels
    run many persistent waves for a while (sponza-class, 1280x720), several frames in a row."""
    import torch
    from vulkanrenderer_amd import FrameState, make_sponza_class
    from vulkanrenderer_amd.renderer import GBuffer, PassExecuteContext, RtaoComputePass, ShadowRays, generate_gbuffer
    scene = make_sponza_class()
    W, H = 1280, 720
    ctx, rt, gb0 = _ctx(scene, W, H)
    gb = GBuffer(ctx, with_shading_inputs=True)
    fs = FrameState(W, H, ray_count=8, sun_samples=16)
    eye, tgt = scene.cameras[0]
    ubos = [fs.update((eye[0] + 0.05 * f, eye[1], eye[2]), tgt) for f in range(3)]
    generate_gbuffer(ctx, ubos[0], gb)
    ctx.sync()
    sh = ShadowRays(ctx, gb)
    rp = RtaoComputePass(ctx, gb, atrous_iterations=2)
    want = []
    for ubo in ubos:                                   # sequential reference: one stream
        rp.execute(PassExecuteContext(ubo=ubo))
        sun = sh.sun(ubo).clone()
        ctx.sync()
        want.append((to_np_u16(rp.rtao_full).copy(), sun.cpu().numpy()))
    rp.reset()
    side, ev = torch.cuda.Stream(), torch.cuda.Event()
    for k, ubo in enumerate(ubos):                     # overlapped: sun on the second stream, queued right behind K1
        rp.execute(PassExecuteContext(ubo=ubo))
        with torch.cuda.stream(side):
            sh.sun(ubo)
            ev.record(side)
        torch.cuda.current_stream().wait_event(ev)
        ctx.sync()
        assert to_np_u16(rp.rtao_full).tobytes() == want[k][0].tobytes(), k
        assert sh.sun_vis.cpu().numpy().tobytes() == want[k][1].tobytes(), k
    ctx.close()


def test_full_size_properties_1080p(torch_mod):
    """At BASELINE size without the oracle: AO in [0,1], sky pixels exactly 1, a static camera converges
    (|frame k+1 - frame k| shrinks), ray counters consistent with the valid-pixel count."""
    from vulkanrenderer_amd import FrameState, make_sponza_class
    from vulkanrenderer_amd.renderer import (PassExecuteContext, RtaoComputePass, generate_gbuffer, trace_stats,
                                             trace_stats_enable)
    scene = make_sponza_class()
    W, H = 1920, 1080
    ctx, rt, gb = _ctx(scene, W, H)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=4, want_hit_mask=True)
    fs = FrameState(W, H, ray_count=8)
    eye, tgt = scene.cameras[0]
    trace_stats_enable(ctx, True)
    trace_stats(ctx, reset=True)
    prev, deltas = None, []
    for f in range(4):
        ubo = fs.update(eye, tgt)
        generate_gbuffer(ctx, ubo, gb)
        rpass.execute(PassExecuteContext(ubo=ubo))
        ctx.sync()
        ao = to_np_u16(rpass.rtao_full).view(np.float16).astype(np.float32)
        valid = gb.valid.cpu().numpy() > 0
        assert ao.min() >= 0.0 and ao.max() <= 1.0 and np.all(ao[~valid] == 1.0)
        if f == 0:
            st = trace_stats(ctx, reset=True)
            nvalid = int(valid.sum())
            assert 8 * nvalid <= st.rays <= 16 * nvalid            # one ray per sample + at most one retry each
            assert st.nodes_visited >= st.rays and st.overflow_tiles <= st.nodes_visited // 20    # queue spills to the global stack stay rare (1 % with the 288 + 128-entry LDS queues)
            trace_stats_enable(ctx, False)
        if prev is not None:
            deltas.append(float(np.abs(ao - prev).mean()))
        prev = ao
    print("static-camera frame deltas", deltas)
    assert deltas[-1] < deltas[0]
    ctx.close()


def test_error_behaviour(torch_mod):
    """Errors mirror the reference's throws: state errors before init, instance count mismatch on updateTopLevelAS
    (RayTracingContext.cpp:51-53), invalid mesh index (:58-60); bad arguments are rejected before any launch."""
    from vulkanrenderer_amd import FrameState, _capi, make_test_scene
    from vulkanrenderer_amd.renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass,
                                             scene_to_desc)
    scene = make_test_scene()
    W, H = 32, 24
    ctx = Context(W, H)
    gb = GBuffer(ctx)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=2)
    ubo = FrameState(W, H).update((0, 1, 0), (0, 1, 5))
    with pytest.raises(RuntimeError, match="before vkr_bvh_build"):
        rpass.execute(PassExecuteContext(ubo=ubo))
    rt = RayTracingContext(ctx)
    with pytest.raises(RuntimeError, match="before vkr_scene_upload"):
        rt.rebuild()
    with pytest.raises(RuntimeError, match="before vkr_bvh_build"):
        rt.updateTopLevelAS(scene.instance_mesh, scene.instance_transform)
    rt.init(scene)
    with pytest.raises(RuntimeError, match="instance count mismatch"):
        rt.updateTopLevelAS(scene.instance_mesh[:-1], scene.instance_transform[:-1])
    swapped = scene.instance_mesh.copy()
    swapped[0] = (swapped[0] + 1) % len(scene.meshes)
    with pytest.raises(RuntimeError, match="mesh index changed"):
        rt.updateTopLevelAS(swapped, scene.instance_transform)
    # the failed refits left the BVH usable
    rpass.execute(PassExecuteContext(ubo=ubo))
    ctx.sync()
    rpass.band = (10, 10)
    with pytest.raises(RuntimeError, match="outside image height"):
        rpass.execute(PassExecuteContext(ubo=ubo))
    rpass.band = (0, H + 1)
    with pytest.raises(RuntimeError, match="outside image height"):
        rpass.execute(PassExecuteContext(ubo=ubo))
    rpass.band = None
    rpass.iterations = 17
    with pytest.raises(RuntimeError, match="atrous_iterations"):
        rpass.execute(PassExecuteContext(ubo=ubo))
    rpass.iterations = 2
    io = rpass._io(ubo)
    io.linear_depth_r16f = None
    lib = ctx.lib
    assert lib.vkr_rtao_execute(ctx.handle, C.byref(io), None) == _capi.VKR_ERR_INVALID_ARG
    assert b"linear_depth_r16f" in lib.vkr_last_error(ctx.handle)
    assert lib.vkr_rtao_execute(ctx.handle, None, None) == _capi.VKR_ERR_INVALID_ARG
    p = C.c_void_p()
    assert lib.vkr_rtao_image(ctx.handle, 9, C.byref(p)) == _capi.VKR_ERR_INVALID_ARG
    # invalid mesh index in an instance
    desc, keep = scene_to_desc(scene)
    desc.instances[0].mesh_index = 10 ** 6
    assert lib.vkr_scene_upload(ctx.handle, C.byref(desc)) == _capi.VKR_ERR_INVALID_ARG
    assert b"invalid meshIndex" in lib.vkr_last_error(ctx.handle)
    # an index past the vertex count
    desc, keep = scene_to_desc(scene)
    bad = np.array(scene.meshes[0].indices, dtype=np.uint32).copy()
    bad[0] = 10 ** 7
    desc.meshes[0].indices = bad.ctypes.data
    assert lib.vkr_scene_upload(ctx.handle, C.byref(desc)) == _capi.VKR_ERR_INVALID_ARG
    # the failed upload invalidated the scene: executing now is a state error, not a crash
    with pytest.raises(RuntimeError, match="before vkr_bvh_build"):
        rpass.execute(PassExecuteContext(ubo=ubo))
    bad_desc = _capi.CreateDesc(0, 0, 10)
    h = C.c_void_p()
    assert lib.vkr_create(C.byref(bad_desc), C.byref(h)) == _capi.VKR_ERR_INVALID_ARG
    bad_desc = _capi.CreateDesc(99, 16, 16)
    assert lib.vkr_create(C.byref(bad_desc), C.byref(h)) == _capi.VKR_ERR_INVALID_ARG
    ctx.close()


@pytest.mark.parametrize("L,radius,samples", [((0.0, 1.0, 0.0), 0.05, 16), ((1.0, 0.0, 0.0), 0.2, 16), ((0.0, -1.0, 0.0), 0.02, 7),
                                              ((-0.6, 0.01, 0.8), 0.3, 16), ((0.0, 0.0, 0.0), 0.01, 8), ((0.3, 0.9995, 0.01), 0.01, 12)])
def test_sun_packet_kernel_direction_edge_cases(torch_mod, oracle_mod, L, radius, samples):
    """The packet traversal's per-axis interval test: sample directions that straddle zero on one or two axes
    (sun at the zenith / along an axis, wide discs), the alternative tangent frame (|dir.y| >= 0.999, pbr.frag:239),
    a downward sun and the zero-length light direction (pbr.frag:219-222) -- every ray equal to the oracle's."""
    import ctypes as C
    from vulkanrenderer_amd import FrameState, make_street_scene, camera_path
    from vulkanrenderer_amd.renderer import _check
    O = oracle_mod
    scene = make_street_scene(13, 0.02, "shadow")
    W, H = 128, 72
    ctx, rt, gb = _ctx(scene, W, H)
    osc = O.OracleScene(scene)
    fs = FrameState(W, H)
    eye, tgt = camera_path(scene, 1)[0]
    ubo = fs.update(eye, tgt)
    depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
    gb.upload(depth, normal, wp, ng, valid)
    import torch
    out = torch.zeros((H, W), dtype=torch.float32, device=ctx.device)
    Lc = (C.c_float * 3)(*L)
    _check(ctx.handle, ctx.lib.vkr_shadow_sun(ctx.handle, gb.world_pos.data_ptr(), gb.geom_normal.data_ptr(), gb.valid.data_ptr(),
                                              W * H, Lc, float(radius), float(samples), out.data_ptr(), ctx.stream()),
           "vkr_shadow_sun")
    ctx.sync()
    got = out.cpu().numpy()
    ref = osc.shadow_sun(wp, ng, valid, np.asarray(L, dtype=np.float32), radius, samples)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    if any(L):
        assert 0.0 < got[valid > 0].mean() < 1.0 or L[1] < 0
    ctx.close()
