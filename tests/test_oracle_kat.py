"""Known-answer tests that pin the CPU oracle (SURVEY.md 8c pins 1-9).

The reference has no tests or golden vectors of its own, so these hand-derived cases -- closed forms, exact bit
patterns, and invariants read off the GLSL -- are what anchors the restatement.  Citations are to
assets/shaders/CompShaders/rtao_trace_half.comp (trace), rtao_atrous.comp (atrous), FragShaders/pbr.frag.
"""
import ctypes as C
import math

import numpy as np
import pytest

from vulkanrenderer_amd import FrameState, make_material_matrix_scene
from vulkanrenderer_amd.scenes import (ALPHA_BLEND, ALPHA_MASK, ALPHA_OPAQUE, Material, Mesh, SceneData, _make_vertices, trs,
                                       UBO_IBL_PARAMS, UBO_RTAO_PARAMS1)


def f32(x):
    return np.float32(x)


def half(a):
    return np.asarray(a, dtype=np.float16).view(np.uint16)


def unhalf(a):
    return a.view(np.float16).astype(np.float32)


def quad_scene(z=1.0, size=1.0, material=0, materials=None, extra=None):
    """One quad in the plane z = const spanning [-size, size]^2 (two triangles), facing -z."""
    pos = np.array([[-size, -size, 0], [size, -size, 0], [-size, size, 0], [size, size, 0]], dtype=np.float32)
    uv = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], dtype=np.float32)
    mesh = Mesh(_make_vertices(pos, uv), np.array([0, 2, 1, 1, 2, 3], dtype=np.uint32), material)
    mats = materials or [Material(ALPHA_OPAQUE)]
    meshes = [mesh] + (extra or [])
    return SceneData(meshes, mats, np.array([0], dtype=np.uint32), np.stack([trs((0, 0, z))]), "quad")


# ---- pin 1: radicalInverseVdC / hammersley (:69-82) -- exact
def test_radical_inverse_exact(oracle_mod):
    L = oracle_mod.lib()
    expect = [0.0, 0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875, 0.0625, 0.5625, 0.3125, 0.8125, 0.1875, 0.6875,
              0.4375, 0.9375]
    for i, e in enumerate(expect):
        assert L.orc_radical_inverse(i) == f32(e)
    out = (C.c_float * 2)()
    for n in (4, 8, 16):
        for i in range(n):
            L.orc_hammersley(i, n, out)
            assert out[0] == f32(f32(i) / f32(n)) and out[1] == f32(expect[i])


# ---- pin 2: offsetRay (:112-123) -- exact bit patterns
def test_offset_ray_bits(oracle_mod):
    L = oracle_mod.lib()

    def ref(p, n):
        out = []
        for pc, nc in zip(p, n):
            pc, nc = f32(pc), f32(nc)
            if abs(pc) < f32(1.0 / 32.0):
                out.append(f32(pc + f32(1.0 / 65536.0) * nc))
            else:
                ofi = int(f32(nc * f32(256.0)))          # ivec3() truncates toward zero
                bits = int(np.array(pc, dtype=np.float32).view(np.int32)) + (-ofi if pc < 0 else ofi)
                out.append(np.array(bits, dtype=np.int32).view(np.float32))
        return np.array(out, dtype=np.float32)

    cases = [((1.0, 2.0, 3.0), (0.0, 1.0, 0.0)), ((-5.5, 0.01, 100.0), (1.0, 0.0, -1.0)),
             ((0.02, -0.03, 0.0), (0.6, -0.8, 0.0)), ((-40.25, 7.0, -0.5), (-0.577, 0.577, 0.577)),
             ((1e-3, 50.0, -50.0), (0.0, 0.0, 1.0))]
    for p, n in cases:
        pa, na, out = np.array(p, dtype=np.float32), np.array(n, dtype=np.float32), np.zeros(3, dtype=np.float32)
        L.orc_offset_ray(pa.ctypes.data, na.ctypes.data, out.ctypes.data)
        assert np.array_equal(out.view(np.uint32), ref(p, n).view(np.uint32)), (p, n)
        # the offset always moves along +n for components with |p| >= 1/32 (the point of the int-bit trick)
        for k in range(3):
            if abs(pa[k]) >= 1 / 32 and abs(na[k]) * 256 >= 1:
                assert (out[k] - pa[k]) * na[k] > 0


# ---- pin 3: buildTbn (:84-90)
def test_build_tbn(oracle_mod):
    L = oracle_mod.lib()

    def tbn(n):
        n = np.array(n, dtype=np.float32)
        t, b = np.zeros(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
        L.orc_build_tbn(n.ctypes.data, t.ctypes.data, b.ctypes.data)
        return t, b

    t, b = tbn((0, 1, 0))        # |n.y| >= 0.999 -> up = (1,0,0): t = normalize(cross(up, n)) = (0,0,1), b = cross(n,t)
    assert np.allclose(t, (0, 0, 1)) and np.allclose(b, (1, 0, 0))
    t, b = tbn((0, -1, 0))
    assert np.allclose(t, (0, 0, -1)) and np.allclose(b, (1, 0, 0))
    t, b = tbn((1, 0, 0))        # up = (0,1,0): t = cross(up,n) = (0,0,-1), b = cross(n,t) = (0,1,0)
    assert np.allclose(t, (0, 0, -1)) and np.allclose(b, (0, 1, 0))
    n = np.array((0.3, 0.5, -0.81), dtype=np.float32)
    n /= np.linalg.norm(n)
    t, b = tbn(n)
    assert abs(np.dot(t, n)) < 1e-6 and abs(np.dot(b, n)) < 1e-6 and abs(np.dot(t, b)) < 1e-6
    assert np.allclose(np.cross(t, b), n, atol=1e-6)      # right-handed: mat3(t,b,n)


# ---- the fixed transcendental definitions stay within GLSL-grade accuracy of libm
def test_det_math_accuracy(oracle_mod):
    L = oracle_mod.lib()
    xs = np.concatenate([np.linspace(-10, 10, 2001), np.linspace(0, 7e4, 1500), [1.3e6 + 0.37, 123456.789]]).astype(np.float32)
    for x in xs:
        assert abs(L.orc_sin(float(x)) - math.sin(float(x))) < 2e-6
        assert abs(L.orc_cos(float(x)) - math.cos(float(x))) < 2e-6
    for x in np.linspace(-87, 0, 3000).astype(np.float32):
        e = math.exp(float(x))
        assert abs(L.orc_exp(float(x)) - e) <= 3e-7 * e + 1e-45
    assert L.orc_exp(-88.0) == 0.0 and L.orc_exp(0.0) == 1.0
    # hash11 = fract(sin(x) * 43758.5453123) stays in [0,1)
    for x in (0.0, 1.0, 57.0, 1919 + 57 * 1079 + 131 * 500):
        h = L.orc_hash11(float(x))
        assert 0.0 <= h < 1.0
    assert L.orc_smoothstep(0.02, 0.15, 0.02) == 0.0 and L.orc_smoothstep(0.02, 0.15, 0.15) == 1.0
    assert abs(L.orc_smoothstep(0.02, 0.15, 0.085) - 0.5) < 1e-6        # midpoint (pin 7: depthDiff 0.085)


def test_inverse_m4(oracle_mod):
    from vulkanrenderer_amd.scenes import look_at_rh, perspective_rh_zo_yflip, col_major
    L = oracle_mod.lib()
    for m in (look_at_rh((1, 2, 3), (4, 1, -2)), perspective_rh_zo_yflip(45, 16 / 9, 0.1, 1000)):
        a = col_major(m)
        out = np.zeros(16, dtype=np.float32)
        L.orc_inverse_m4(a.ctypes.data, out.ctypes.data)
        ident = out.reshape(4, 4).T.astype(np.float64) @ m.astype(np.float64)
        assert np.allclose(ident, np.eye(4), atol=2e-5)


# ---- samplers
def test_samplers(oracle_mod):
    L = oracle_mod.lib()
    W, H = 4, 3
    img = half(np.arange(12, dtype=np.float32).reshape(H, W))
    p = img.ctypes.data
    for y in range(H):
        for x in range(W):
            u, v = (x + 0.5) / W, (y + 0.5) / H
            assert L.orc_tex_nearest(p, W, H, u, v) == y * W + x
            assert L.orc_tex_linear(p, W, H, u, v) == y * W + x       # texel centre: bilinear == texel
    assert L.orc_tex_nearest(p, W, H, -0.1, 0.5) == 4 and L.orc_tex_nearest(p, W, H, 1.0, 1.0) == 11   # clamp-to-edge
    assert abs(L.orc_tex_linear(p, W, H, 0.5, 0.5) - 5.5) < 1e-6      # between texels 1,2 of rows 1: (5+6)/2
    assert L.orc_tex_linear(p, W, H, 0.0, 0.0) == 0.0                 # clamp: all four taps are texel (0,0)


# ---- pin 5: traceAoRay distance curve (:184-213) on a single quad at distance t
@pytest.mark.parametrize("brute", [True, False])
def test_trace_ao_ray_curve(oracle_mod, brute):
    O = oracle_mod
    tmin, tmax = 0.01, 1.0
    o, d = (0.1, 0.2, 0.0), (0.0, 0.0, 1.0)

    def vis_at(dist):
        s = O.OracleScene(quad_scene(z=dist), brute_force=brute)
        return s.trace_ao_ray(o, d, tmin, tmax)

    assert vis_at(2.0) == (1.0, 0)                       # beyond tMax: miss -> 1
    for dist in (0.25, 0.5, 0.9):
        v, occ = vis_at(dist)
        x = f32((f32(dist) - f32(tmin)) / f32(tmax - tmin))
        assert occ == 1 and abs(v - float(x) ** 3) < 1e-6
    # t <= 1.5*tMin: retried from origin + dir*max(2 tMin, 2e-4); the quad is behind the advanced origin -> miss -> 1
    assert vis_at(0.5 * tmin * 2.6) == (1.0, 0)          # t = 0.013 <= 0.015, advanced origin at 0.02 > 0.013
    # t below tMin is not an intersection at all
    assert vis_at(0.005) == (1.0, 0)
    # hit exactly inside (tMin, 1.5 tMin] but still in front of the advanced origin cannot happen (advance = 2 tMin);
    # a second surface behind the self-hit is what the retry finds:
    two = quad_scene(z=0.012)
    far = quad_scene(z=0.6)
    two.meshes.append(far.meshes[0])
    two.instance_mesh = np.array([0, 1], dtype=np.uint32)
    two.instance_transform = np.stack([trs((0, 0, 0.012)), trs((0, 0, 0.6))])
    v, occ = O.OracleScene(two, brute_force=brute).trace_ao_ray(o, d, tmin, tmax)
    x = f32((f32(0.6) - f32(tmin)) / f32(tmax - tmin))
    assert occ == 1 and abs(v - float(x) ** 3) < 1e-5    # t = retryT + advance = 0.6


# ---- pin 6: candidate filter of queryAoClosestHit (:141-175) vs the all-opaque shadow query (pbr.frag:199-215)
@pytest.mark.parametrize("brute", [True, False])
def test_material_matrix(oracle_mod, brute):
    s = oracle_mod.OracleScene(make_material_matrix_scene(), brute_force=brute)
    d = (0.0, 0.0, 1.0)
    expect_t = {0: 1.0, 1: 1.0, 2: 2.0, 3: 2.0}   # opaque | MASK alpha=1 | MASK alpha=0 -> wall | BLEND -> wall
    for col in range(4):
        o = (2.0 * col + 0.5, 0.5, 0.0)
        hit, t = s.query_ao(o, d, 1e-3, 10.0)
        assert hit and abs(t - expect_t[col]) < 1e-6, col
        assert s.shadow_ray(o, d) == 0.0           # gl_RayFlagsOpaqueEXT: every triangle occludes
    assert s.query_ao((0.5, 0.5, 2.5), d, 1e-3, 10.0) == (False, 10.0)     # miss: tHit = tMax (:176-179)
    assert s.shadow_ray((0.5, 0.5, 2.5), d) == 1.0
    # no back-face culling for AO (:131-134): the same quads from behind
    hit, t = s.query_ao((0.5, 0.5, 3.0), (0, 0, -1), 1e-3, 10.0)
    assert hit and abs(t - 1.0) < 1e-6


def test_brute_force_equals_bvh(oracle_mod):
    from vulkanrenderer_amd import make_street_scene
    scene = make_street_scene(3, 0.01, "t")
    a, b = oracle_mod.OracleScene(scene, True), oracle_mod.OracleScene(scene, False)
    rng = np.random.default_rng(0)
    for _ in range(300):
        o = rng.uniform((-5, 0.1, 0), (5, 6, 80)).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        d /= np.linalg.norm(d)
        assert a.query_ao(o, d, 1e-3, 3.0) == b.query_ao(o, d, 1e-3, 3.0)
        assert a.shadow_ray(o, d) == b.shadow_ray(o, d)


# ---- pin 4 + 7: K1 on a planar G-buffer with no geometry (every ray misses -> current = 1)
def _plane_gbuffer(W, H, fs, eye, tgt, plane_y=0.0):
    """Depth of the ground plane y = plane_y seen from the camera, quantised like the reference (R16F), sky = 0."""
    from vulkanrenderer_amd.scenes import look_at_rh, perspective_rh_zo_yflip
    ubo = fs.update(eye, tgt)
    view = look_at_rh(eye, tgt).astype(np.float64)
    proj = perspective_rh_zo_yflip(45.0, W / H, 0.1, 1000.0).astype(np.float64)
    inv_p, inv_v = np.linalg.inv(proj), np.linalg.inv(view)
    depth = np.zeros((H, W), dtype=np.float32)
    for y in range(H):
        for x in range(W):
            ndc = np.array([(x + 0.5) / W * 2 - 1, (y + 0.5) / H * 2 - 1, 1.0, 1.0])
            fv = inv_p @ ndc
            dv = fv[:3] / fv[3]
            dv /= np.linalg.norm(dv)
            dw = inv_v[:3, :3] @ dv
            if dw[1] < -1e-6:
                t = (plane_y - eye[1]) / dw[1]
                depth[y, x] = -(dv[2] * t)
    normal = np.zeros((H, W, 4), dtype=np.float32)
    normal[..., :] = (0.5, 1.0, 0.5, 1.0)
    return ubo, half(depth), half(normal)


def test_trace_plane_debug_taps_and_temporal(oracle_mod):
    O = oracle_mod
    W, H = 48, 27
    empty = quad_scene(z=1e4)       # far away: no AO ray can reach it
    osc = O.OracleScene(empty)
    eye, tgt = np.array((0.0, 2.0, 0.0), dtype=np.float32), np.array((0.0, 0.5, 6.0), dtype=np.float32)
    one = half(np.ones((H, W), dtype=np.float32))

    # debug 11 (:386-390): footprint/0.05 clamped; debug 10 (:381-385): bias/0.05
    for mode in (10, 11):
        fs = FrameState(W, H, ray_count=4, debug_view=mode)
        # mode 11 saturates at footprint 0.05: look at the ground from close up so the tap stays below 1
        e, t = (eye, tgt) if mode == 10 else (np.array((0.0, 0.3, 0.0), dtype=np.float32), np.array((0.0, 0.0, 1.2), dtype=np.float32))
        ubo, depth, normal = _plane_gbuffer(W, H, fs, e, t)
        out = unhalf(osc.trace(ubo, depth, one))
        ground = unhalf(depth) > 0
        assert ground.any() and (~ground).any()
        assert np.all(out[~ground] == 1.0)                       # sky early-out (:240-243)
        if mode == 10:
            # bias = clamp(max(5e-4, 0.05 fp), 5e-4, max(5e-4, min(0.02*1.2, 0.03))) in [5e-4, 0.024]
            assert np.all(out[ground] >= 0.0099) and np.all(out[ground] <= 0.4801)
        else:
            # footprint grows with view distance on a receding plane: rows nearer the horizon have larger values
            col = out[:, W // 2][ground[:, W // 2]]
            z = unhalf(depth)[:, W // 2][ground[:, W // 2]]
            order = np.argsort(z)
            assert np.all(np.diff(col[order]) >= -2e-3)
            assert col.max() > col.min()

    # debug 12 (:435-438): current without temporal; no occluder -> exactly 1
    fs = FrameState(W, H, ray_count=4, debug_view=12)
    ubo, depth, normal = _plane_gbuffer(W, H, fs, eye, tgt)
    assert np.all(unhalf(osc.trace(ubo, depth, one)) == 1.0)

    # temporal (:440-471): frame 0 ignores history (historyWeight = 0)
    fs = FrameState(W, H, ray_count=4)
    ubo, depth, normal = _plane_gbuffer(W, H, fs, eye, tgt)
    assert ubo[UBO_RTAO_PARAMS1 + 3] == 0.0
    garbage = half(np.full((H, W), 0.25, dtype=np.float32))
    assert np.all(unhalf(osc.trace(ubo, depth, garbage)) == 1.0)
    # frame 1, static camera: current = 1, sigma = 0 -> history clamped to current -> accum = 1 whatever history holds
    ubo1, depth, normal = _plane_gbuffer(W, H, fs, eye, tgt)
    assert ubo1[UBO_RTAO_PARAMS1 + 3] == 1.0
    assert np.all(unhalf(osc.trace(ubo1, depth, garbage)) == 1.0)


def test_temporal_blend_values(oracle_mod):
    """F16 with an occluder: current c in (0,1), history h: accum = mix(c, clamp(h, c +- 2.5 sigma), 0.8 (1 - reject))."""
    O = oracle_mod
    W, H = 32, 18
    # a ceiling 0.5 m above the ground plane occludes every AO ray partially
    ceil_scene = quad_scene(z=0.0, size=50.0)
    ceil_scene.instance_transform = np.stack([np.array([1, 0, 0, 0, 0, 0, 1, 0.5, 0, 1, 0, 5.0], dtype=np.float32)])
    osc = O.OracleScene(ceil_scene)
    eye, tgt = np.array((0.0, 0.3, 0.0), dtype=np.float32), np.array((0.0, 0.0, 4.0), dtype=np.float32)
    fs = FrameState(W, H, ray_count=8, debug_view=12)
    fs.update(eye, tgt)           # frame index 1 in both runs: the sample rotation depends on it (:395)
    ubo, depth, normal = _plane_gbuffer(W, H, fs, eye, tgt)
    one = half(np.ones((H, W), dtype=np.float32))
    current = unhalf(osc.trace(ubo, depth, one))
    ground = unhalf(depth) > 0
    assert ground.any() and current[ground].min() < 0.9
    # same frame without the debug view, frame index 1, static camera -> prevUv == uv, depthDiff == 0
    fs2 = FrameState(W, H, ray_count=8)
    fs2.update(eye, tgt)
    ubo1, depth, normal = _plane_gbuffer(W, H, fs2, eye, tgt)
    for hval in (0.0, 1.0):
        hist = half(np.full((H, W), hval, dtype=np.float32))
        acc = unhalf(osc.trace(ubo1, depth, hist))
        c = current.astype(np.float64)
        sigma = np.sqrt(np.maximum(c * (1 - c), 0) / 8)
        hc = np.clip(hval, c - 2.5 * sigma, c + 2.5 * sigma)
        thr = min(max(0.35 + 0.6 / math.sqrt(8), 0.0), 1.5)
        delta = np.abs(c - hc)
        t = np.clip((delta - thr) / (thr * 2 + 1e-5 - thr), 0, 1)
        reject = t * t * (3 - 2 * t)
        w = 0.8 * (1 - reject)
        expect = c * (1 - w) + hc * w
        # `current` was itself rounded to fp16 by the debug tap, so compare at fp16 resolution
        assert np.max(np.abs(acc[ground] - expect[ground])) < 3e-3, hval


# ---- pin 8: A-Trous (rtao_atrous.comp:48-120)
def _flat_inputs(W, H, z=5.0):
    depth = half(np.full((H, W), z, dtype=np.float32))
    normal = half(np.tile(np.array([0.5, 1.0, 0.5, 1.0], dtype=np.float32), (H, W, 1)))
    return depth, normal


def test_atrous_constant_image_unchanged(oracle_mod):
    O = oracle_mod
    W, H = 40, 24
    depth, normal = _flat_inputs(W, H)
    ubo = FrameState(W, H).update((0, 1, 0), (0, 1, 1))
    for val in (0.0, 0.37, 1.0):
        src = half(np.full((H, W), val, dtype=np.float32))
        for step in (1, 2, 4, 8, 16):
            out = O.atrous(ubo, step, src, depth, normal)
            assert np.array_equal(out, src), (val, step)      # sum(s w)/sum(w) of a constant is the constant at fp16


def test_atrous_edge_stopping(oracle_mod):
    O = oracle_mod
    W, H = 40, 24
    ubo = FrameState(W, H).update((0, 1, 0), (0, 1, 1))
    ao = np.full((H, W), 0.375, dtype=np.float32)
    ao[:, W // 2:] = 0.625
    src = half(ao)
    # (a) no geometric edge: the AO step is blurred across the boundary (range weight exp(-8|ds|) = exp(-2) only)
    depth, normal = _flat_inputs(W, H)
    out = unhalf(O.atrous(ubo, 1, src, depth, normal))
    assert 0.375 < out[H // 2, W // 2 - 1] < 0.5 and 0.5 < out[H // 2, W // 2] < 0.625
    # (b) a depth step far beyond sigma (max(dz/0.06, (dz/z)/0.0025) * 1.25 >> 1): no bleed at all
    d = np.full((H, W), 5.0, dtype=np.float32)
    d[:, W // 2:] = 8.0
    out = unhalf(O.atrous(ubo, 1, src, half(d), normal))
    assert np.array_equal(out, ao)
    # (c) a normal flip: pow(clamp(dot,0,1),32) = 0 across the edge: no bleed
    n = np.tile(np.array([0.5, 1.0, 0.5, 1.0], dtype=np.float32), (H, W, 1))
    n[:, W // 2:] = (1.0, 0.5, 0.5, 1.0)
    out = unhalf(O.atrous(ubo, 1, src, depth, half(n)))
    assert np.array_equal(out, ao)
    # (d) NaN / zero normals decode to (0,1,0) (:38-46) and behave like the flat case
    n2 = np.tile(np.array([0.5, 0.5, 0.5, 1.0], dtype=np.float32), (H, W, 1))
    out_d = O.atrous(ubo, 1, src, depth, half(n2))
    assert np.array_equal(out_d, O.atrous(ubo, 1, src, depth, normal))


def test_atrous_border_clamp_and_sky_depth(oracle_mod):
    O = oracle_mod
    W, H = 16, 12
    ubo = FrameState(W, H).update((0, 1, 0), (0, 1, 1))
    depth, normal = _flat_inputs(W, H)
    rng = np.random.default_rng(1)
    ao = rng.random((H, W)).astype(np.float32)
    src = half(ao)
    out = unhalf(O.atrous(ubo, 8, src, depth, normal))          # step 8: most taps clamp to the border (:81)
    assert out.min() >= unhalf(src).min() - 1e-3 and out.max() <= unhalf(src).max() + 1e-3
    # sky depth 0 is clamped to 1e-3 (:71,:93), not NaN/inf
    out0 = unhalf(O.atrous(ubo, 1, src, half(np.zeros((H, W), dtype=np.float32)), normal))
    assert np.isfinite(out0).all()


def test_pass_iteration_parity(oracle_mod):
    """Final ping-pong index = (N-1) & 1 (RtaoComputePass.cpp:134-135,560-561); N = 1 still runs one iteration."""
    O = oracle_mod
    W, H = 24, 16
    osc = O.OracleScene(quad_scene(z=1e4))
    eye, tgt = np.array((0.0, 2.0, 0.0), dtype=np.float32), np.array((0.0, 0.5, 6.0), dtype=np.float32)
    for n in (1, 2, 3, 4, 5):
        fs = FrameState(W, H, ray_count=4)
        ubo, depth, normal = _plane_gbuffer(W, H, fs, eye, tgt)
        p = O.OracleRtaoPass(osc, W, H, n)
        full = p.execute(ubo, depth, normal)
        assert np.array_equal(full, p.ping[(n - 1) & 1])
        assert p.parity == 1 and p.last_curr == 0          # history ping-pong flips per frame (FrameManager.cpp:1155-1165)
    # debug view 9 (rtao_upsample.comp:45-52): normal.y*0.5+0.5 of the decoded normal (0,1,0) -> 1
    fs = FrameState(W, H, ray_count=4, debug_view=9)
    ubo, depth, normal = _plane_gbuffer(W, H, fs, eye, tgt)
    full = O.OracleRtaoPass(osc, W, H, 2).execute(ubo, depth, normal)
    assert np.all(unhalf(full) == 1.0)


# ---- pin 9: sun-disc shadows (pbr.frag:217-260)
def test_sun_shadow_cases(oracle_mod):
    O = oracle_mod
    # occluder: a half-plane y = 1, x < 0 (a large quad whose +x edge sits at x = 0)
    occl = quad_scene(z=0.0, size=10.0)
    occl.instance_transform = np.stack([np.array([1, 0, 0, -10.0, 0, 0, 1, 1.0, 0, 1, 0, 0.0], dtype=np.float32)])
    osc = O.OracleScene(occl)
    L = np.array((0, 1, 0), dtype=np.float32)
    xs = np.array([-1.0, -0.05, 0.0, 0.05, 1.0], dtype=np.float32)
    wp = np.stack([xs, np.zeros_like(xs), np.zeros_like(xs)], axis=1).astype(np.float32)
    ng = np.tile(np.array((0, 1, 0), dtype=np.float32), (len(xs), 1))
    valid = np.ones(len(xs), dtype=np.uint8)
    # hard shadow (n <= 1 or radius <= 0): 1.0 lit / 0.2 shadowed (:232-235)
    hard = osc.shadow_sun(wp, ng, valid, L, 0.01, 1)
    assert list(hard) == [f32(0.2), f32(0.2), f32(1.0), f32(1.0), f32(1.0)] or list(hard)[:2] == [f32(0.2)] * 2
    assert hard[0] == f32(0.2) and hard[-1] == 1.0
    assert np.array_equal(osc.shadow_sun(wp, ng, valid, L, 0.0, 16), hard)
    # soft shadow: deep in shadow 0, fully lit 1, at the edge the fraction of the fixed Poisson taps with
    # offset.x > 0 (dir = (0,1,0): up = (0,0,1), T = cross(up,dir) = (-1,0,0), B = cross(dir,T) = (0,0,1):
    # a tap (px,py) leans towards -x*px; the ray from x=0 escapes iff its x-offset is >= 0 <=> px <= 0)
    poisson_x = [-0.613, 0.582, -0.451, 0.208, -0.263, -0.883, 0.759, 0.352, -0.713, 0.039, -0.189, 0.595, -0.515, 0.941,
                 -0.137, 0.169]
    for n in (8, 16):
        soft = osc.shadow_sun(wp, ng, valid, L, 0.2, n)
        assert soft[0] == 0.0 and soft[-1] == 1.0
        lit = sum(1 for p in poisson_x[:n] if p <= 0)
        assert abs(soft[2] - lit / n) <= 1.0 / n + 1e-6
    # |L|^2 < 1e-8 -> 1 (:219-222); invalid pixels -> 1
    assert np.all(osc.shadow_sun(wp, ng, valid, np.zeros(3, dtype=np.float32), 0.01, 8) == 1.0)
    assert np.all(osc.shadow_sun(wp, ng, np.zeros(len(xs), dtype=np.uint8), L, 0.01, 8) == 1.0)


def test_point_shadow_cases(oracle_mod):
    O = oracle_mod
    occl = quad_scene(z=0.0, size=1.0)
    occl.instance_transform = np.stack([np.array([1, 0, 0, 0.0, 0, 0, 1, 1.0, 0, 1, 0, 0.0], dtype=np.float32)])  # y = 1
    osc = O.OracleScene(occl)
    wp = np.array([[0, 0, 0], [5, 0, 0], [0, 2, 0]], dtype=np.float32)
    ng = np.tile(np.array((0, 1, 0), dtype=np.float32), (3, 1))
    valid = np.ones(3, dtype=np.uint8)
    vis = osc.shadow_point(wp, ng, valid, np.array((0, 3, 0), dtype=np.float32))
    assert list(vis) == [0.0, 1.0, 1.0]        # under the quad: blocked; far to the side: clear; above it: clear
    # light closer than 1e-4: visible (:173-175)
    assert osc.shadow_point(wp[:1], ng[:1], valid[:1], np.array((0, 5e-5, 0), dtype=np.float32))[0] == 1.0
    # tMax = dist - bias: an occluder beyond the light does not shadow
    assert osc.shadow_point(wp[:1], ng[:1], valid[:1], np.array((0, 0.5, 0), dtype=np.float32))[0] == 1.0


# ---------------------------------------------------------------------------------------------------------------
# G-buffer producer: interpolated vertex normals through the instance's normal matrix
# (depth_prepass.vert:47-48, depth_only.frag:31-33), Ng flipped on back faces (pbr.frag:339-346)
# ---------------------------------------------------------------------------------------------------------------
def test_gbuffer_vertex_normals_on_a_scaled_sphere(oracle_mod):
    from vulkanrenderer_amd.scenes import Material, SceneData, sphere_mesh, trs, FrameState
    O = oracle_mod
    sph = sphere_mesh(24, 16, 0)
    centre, scale = np.array([0.0, 1.0, 6.0]), np.array([2.0, 1.0, 1.5])
    for mirror in (1.0, -1.0):
        xf = trs(tuple(centre), (scale[0] * mirror, scale[1], scale[2]))
        scene = SceneData([sph], [Material()], np.array([0], dtype=np.uint32), np.stack([xf]).astype(np.float32), "s")
        osc = O.OracleScene(scene)
        W, H = 96, 64
        ubo = FrameState(W, H).update((0.0, 1.5, 0.0), tuple(centre + np.array([0.0, 0.5, 0.0])))
        depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
        v = valid > 0
        assert v.sum() > 200
        n = normal.view(np.float16).astype(np.float32)[..., :3] * 2.0 - 1.0
        # ellipsoid: the analytic normal at p is normalize((p - c) / s^2); the sphere mesh's own centre is its origin
        c = centre + scale * np.array([0.0, 0.5, 0.0])
        local = (wp[v] - c) / (scale ** 2)
        want = local / np.linalg.norm(local, axis=1, keepdims=True)
        # the generator's winding makes this mesh's vertex normals point inwards; the inverse-transpose must keep
        # that orientation for the mirrored instance too (sign of the determinant)
        sgn = np.sign(np.median(np.einsum("ij,ij->i", n[v], want)))
        assert sgn == -1.0, (mirror, sgn)
        err = np.abs(n[v] * sgn - want).max(axis=1)
        assert np.percentile(err, 95) < 0.06, (mirror, np.percentile(err, 95))      # smooth, not faceted
        # pbr.frag:339-346: Ng = interpolated normal, negated when !gl_FrontFacing.  This mesh is wound inside-out, so
        # from outside its faces are back faces (Ng = -n, towards the camera); mirroring the instance reverses the
        # screen-space winding, the faces become front faces and Ng = n points away -- the reference's own behaviour
        # for mirrored instances, reproduced as is.
        cam = np.array([0.0, 1.5, 0.0])
        rel = np.einsum("ij,ij->i", ng[v], n[v])
        assert np.abs(rel - (-1.0 if mirror > 0 else 1.0)).max() < 5e-3
        facing = np.einsum("ij,ij->i", ng[v], cam - wp[v]) > 0
        assert facing.mean() > 0.97 if mirror > 0 else facing.mean() < 0.03
