"""The one model the reference ships (assets/models/viking_room.glb, decoded into tests/golden/viking_room_mesh.npz by
tests/golden/make_viking_fixture.py) through the whole path: the file's own vertex normals, UVs, uint16 indices and
node rotation.  CPU: fixture sanity + oracle structure checks.  GPU: G-buffer, RTAO chain and shadow rays bit-exact
against the oracle on this real mesh (thin double-sided walls, long sliver triangles, T-junctions)."""
import os

import numpy as np
import pytest

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "viking_room_mesh.npz")
W, H = 192, 108


def viking_scene():
    from vulkanrenderer_amd.scenes import scene_from_gltf_arrays
    d = np.load(FIX)
    return scene_from_gltf_arrays(d["positions"], d["normals"], d["uvs"], d["indices"], d["node_local_col_major"],
                                  name="viking_room", scale=4.0, translate=(0.0, 0.0, 0.0)), d


def test_fixture_is_the_reference_model():
    sc, d = viking_scene()
    assert d["positions"].shape == (4677, 3) and d["indices"].shape == (11484,) and int(d["indices"].max()) == 4676
    assert sc.triangle_count() == 3828 and int(d["double_sided"]) == 1
    assert np.allclose(np.linalg.norm(d["normals"], axis=1), 1.0, atol=1e-3)
    assert 0.0 <= d["uvs"].min() and d["uvs"].max() <= 1.0
    # the node rotates +90 degrees about X (glTF quaternion 0.7071, 0, 0, 0.7071): column-major local matrix
    m = d["node_local_col_major"].reshape(4, 4).T
    assert np.allclose(m[:3, :3], [[1, 0, 0], [0, 0, -1], [0, 1, 0]], atol=1e-6)
    assert sc.meshes[0].vertices[:, 3:6].tobytes() == d["normals"].tobytes()       # the file's normals, not recomputed


def test_oracle_on_viking_room(oracle_mod):
    from vulkanrenderer_amd import FrameState
    sc, _ = viking_scene()
    osc = oracle_mod.OracleScene(sc)
    brute = oracle_mod.OracleScene(sc, brute_force=True)
    fs = FrameState(W // 2, H // 2, ray_count=4, radius=0.8)
    ubo = fs.update(*sc.cameras[0])
    depth, normal, wp, ng, valid = osc.gbuffer(ubo, W // 2, H // 2)
    d2, n2, _, _, v2 = brute.gbuffer(ubo, W // 2, H // 2)
    assert np.array_equal(depth, d2) and np.array_equal(normal, n2) and np.array_equal(valid, v2)   # BVH == brute force
    assert 0.3 < (valid > 0).mean() < 1.0
    p = oracle_mod.OracleRtaoPass(osc, W // 2, H // 2, 2)
    full, mask = p.execute(ubo, depth, normal, want_mask=True)
    pb = oracle_mod.OracleRtaoPass(brute, W // 2, H // 2, 2)
    fullb, maskb = pb.execute(ubo, depth, normal, want_mask=True)
    assert np.array_equal(full, fullb) and np.array_equal(mask, maskb)
    ao = full.view(np.float16).astype(np.float32)
    assert ao[valid > 0].min() < 0.7 and ao[valid > 0].mean() > 0.5         # a room interior: corners occluded, floor open


@pytest.mark.gpu
def test_gpu_viking_room_chain_and_shadows(oracle_mod):
    import torch
    from helpers import compare_r16f, to_np_u16
    from vulkanrenderer_amd import FrameState
    from vulkanrenderer_amd.renderer import (Context, RayTracingContext, GBuffer, RtaoComputePass, PassExecuteContext,
                                             ShadowRays, generate_gbuffer)
    sc, _ = viking_scene()
    osc = oracle_mod.OracleScene(sc)
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(sc)
    assert rt.info().triangle_count == 3828
    gb = GBuffer(ctx, with_shading_inputs=True)
    rpass = RtaoComputePass(ctx, gb, atrous_iterations=3, want_hit_mask=True)
    opass = oracle_mod.OracleRtaoPass(osc, W, H, 3)
    fs = FrameState(W, H, ray_count=8, radius=0.8, sun_samples=8)
    sh = ShadowRays(ctx, gb)
    for frame, (eye, tgt) in enumerate(sc.cameras):
        ubo = fs.update(eye, tgt)
        depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
        generate_gbuffer(ctx, ubo, gb)                                     # the device G-buffer producer, this mesh's normals
        ctx.sync()
        assert to_np_u16(gb.linear_depth).tobytes() == depth.tobytes()
        assert to_np_u16(gb.normal).tobytes() == normal.tobytes()
        assert gb.valid.cpu().numpy().tobytes() == valid.tobytes()
        assert gb.world_pos.cpu().numpy().tobytes() == wp.tobytes() and gb.geom_normal.cpu().numpy().tobytes() == ng.tobytes()
        rpass.execute(PassExecuteContext(ubo=ubo))
        ctx.sync()
        ref_full, ref_mask = opass.execute(ubo, depth, normal, want_mask=True)
        full = compare_r16f(to_np_u16(rpass.rtao_full), ref_full)
        hist = compare_r16f(to_np_u16(rpass.image(0)), opass.hist[opass.last_curr])
        got_mask = rpass.hit_mask.cpu().numpy().view(np.uint32)
        assert full["frac_gt_1_255"] <= 0.005 and hist["frac_gt_1_255"] <= 0.005          # north-star tolerance
        assert int(np.unpackbits((got_mask ^ ref_mask).view(np.uint8)).sum()) == 0
        assert full["bit_mismatch"] == 0 and hist["bit_mismatch"] == 0
        sun = sh.sun(ubo).cpu().numpy()
        want_sun = osc.shadow_sun(wp, ng, valid, FrameState.sun_dir_to_light(ubo), float(ubo[56]), int(ubo[57]))
        assert sun.tobytes() == want_sun.tobytes()
        pt = sh.point(ubo, 0).cpu().numpy()
        assert pt.tobytes() == osc.shadow_point(wp, ng, valid, ubo[60:63]).tobytes()
    ctx.close()
