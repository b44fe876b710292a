"""Parity is unpinned (the reference holds no vectors), and in three places the contract the GPU kernels implement departs
from the literal shader text.  This CPU-only test quantifies each departure with oracle switches (oracle.set_variant) against
the north-star tolerance (per-ray hit/miss equal except grazing rays; final AO <= 0.5 % of pixels off by > 1/255), on BASELINE
configs[0] in full and on a row band of configs[3].  The numbers it prints are the ones quoted in DESIGN.md section 2.

  1  A-Trous per-tap quotients: IEEE `/` (rtao_atrous.comp:99,103) vs reciprocal multiplies (contract)
  2  A-Trous fetches: samplers at the computed uv (rtao_atrous.comp:55-56,80-89: LINEAR AO image, NEAREST depth/normal)
     vs exact texel fetches (contract)
  4  queryAoClosestHit: TerminateOnFirstHit taken literally, first ACCEPTED candidate in traversal order
     (rtao_trace_half.comp:131-139,176-181) vs closest committed hit (contract); 4|8 = the reverse traversal order
"""
import numpy as np
import pytest

from oracle import oracle as O
from vulkanrenderer_amd import FrameState, camera_path, make_bistro_class, make_sponza_class

TOL_FRAC = 0.005          # north star: <= 0.5 % of pixels may differ by more than 1/255
TOL_ABS = 1.0 / 255.0


def _f(a):
    return np.asarray(a).view(np.float16).astype(np.float32)


def _stats(ref, got):
    d = np.abs(_f(ref) - _f(got))
    return {"frac_over_1_255": float(np.mean(d > TOL_ABS)), "max_abs": float(d.max()),
            "frac_bits_differ": float(np.mean(np.asarray(ref) != np.asarray(got)))}


def _chain_band(osc, ubo, depth, normal, hist, iters, ya, yb):
    """K1 on the band + A-Trous halo, then the iterations with a shrinking row range -> (rtao_full rows, history rows, mask rows)."""
    H, W = depth.shape
    halo = 2 * ((1 << iters) - 1)
    t0, t1 = max(0, ya - halo), min(H, yb + halo)
    cur, mask = osc.trace(ubo, depth, hist, t0, t1, want_mask=True)
    src = np.asarray(cur).view(np.uint16).reshape(H, W)
    for i in range(iters):
        rem = 2 * ((1 << iters) - (1 << (i + 1)))
        src = O.atrous(ubo, 1 << i, src, depth, normal, max(0, ya - rem), min(H, yb + rem))
    return src[ya:yb].copy(), np.asarray(cur).view(np.uint16).reshape(H, W)[ya:yb].copy(), np.asarray(mask).reshape(H, W)[ya:yb].copy()


@pytest.fixture(scope="module")
def c0_runs():
    """configs[0]: sponza-class 640x360, 4 rays/px, temporal, 2 A-Trous iterations, two frames, every variant."""
    O.set_threads(min(8, O.max_threads()))
    scene = make_sponza_class()
    W, H = 640, 360
    osc = O.OracleScene(scene)
    fs = FrameState(W, H, ray_count=4)
    ubos = [fs.update(e, t) for e, t in camera_path(scene, 2, step=0.05)]
    gbs = [osc.gbuffer(u, W, H) for u in ubos]
    runs = {}
    try:
        for var in (0, 1, 2, 4, 12):
            O.set_variant(var)
            p = O.OracleRtaoPass(osc, W, H, 2)
            frames = []
            for u, g in zip(ubos, gbs):
                full, mask = p.execute(u, g[0], g[1], want_mask=True)
                frames.append((full.copy(), mask.copy(), p.hist[p.last_curr].copy()))
            runs[var] = frames
    finally:
        O.set_variant(0)
    return runs, 4 * W * H


@pytest.fixture(scope="module")
def c3_runs():
    """configs[3]: bistro-class 1920x1080, 8 rays/px, 4 A-Trous iterations; a 48-row band of one frame (cleared history)."""
    O.set_threads(min(8, O.max_threads()))
    scene = make_bistro_class()
    W, H = 1920, 1080
    osc = O.OracleScene(scene)
    fs = FrameState(W, H, ray_count=8)
    ubo = [fs.update(e, t) for e, t in camera_path(scene, 1, step=0.05)][0]
    g = osc.gbuffer(ubo, W, H)
    hist = np.full((H, W), np.float16(1.0).view(np.uint16), dtype=np.uint16)
    ya, yb = 600, 648
    runs = {}
    try:
        for var in (0, 1, 2, 4, 12):
            O.set_variant(var)
            runs[var] = _chain_band(osc, ubo, g[0], g[1], hist, 4, ya, yb)
    finally:
        O.set_variant(0)
    return runs, 8 * W * (yb - ya)


def _report(tag, var, ref, got, n_rays):
    full, hist = _stats(ref[0], got[0]), _stats(ref[2] if len(ref) == 3 and ref[2].dtype == np.uint16 else ref[1], got[2] if len(got) == 3 and got[2].dtype == np.uint16 else got[1])
    m_ref = ref[1] if ref[1].dtype == np.uint32 else ref[2]
    m_got = got[1] if got[1].dtype == np.uint32 else got[2]
    ray_bits = int(np.unpackbits((m_ref ^ m_got).view(np.uint8)).sum())
    print(f"[departure] {tag} variant {var}: rtao_full {100 * full['frac_over_1_255']:.4f} % px > 1/255, max-abs {full['max_abs']:.4f}, "
          f"{100 * full['frac_bits_differ']:.3f} % px differ at all | K1 history {100 * hist['frac_over_1_255']:.4f} % px > 1/255, "
          f"max-abs {hist['max_abs']:.4f} | rays with a different hit/miss bit {ray_bits} of {n_rays} ({100.0 * ray_bits / n_rays:.5f} %)")
    return full, hist, ray_bits


@pytest.mark.parametrize("var", [1, 2])
def test_atrous_departures_are_inside_the_tolerance_c0(c0_runs, var):
    """Reciprocal-vs-division and texel-fetch-vs-sampler change single fp16 ulps: 0 pixels beyond 1/255, K1 untouched."""
    runs, n_rays = c0_runs
    for f in range(2):
        full, hist, ray_bits = _report(f"c0 frame {f}", var, runs[0][f], runs[var][f], n_rays)
        assert full["frac_over_1_255"] <= TOL_FRAC and full["max_abs"] <= 2.0e-3
        assert hist["frac_bits_differ"] == 0.0 and ray_bits == 0          # A-Trous switches do not touch K1


@pytest.mark.parametrize("var", [1, 2])
def test_atrous_departures_are_inside_the_tolerance_c3_band(c3_runs, var):
    runs, n_rays = c3_runs
    full, hist, ray_bits = _report("c3 band", var, runs[0], runs[var], n_rays)
    assert full["frac_over_1_255"] <= TOL_FRAC and full["max_abs"] <= 4.0e-3
    assert hist["frac_bits_differ"] == 0.0 and ray_bits == 0


@pytest.mark.parametrize("var", [4, 12])
def test_first_found_hit_keeps_hit_miss_but_not_the_ao_value(c0_runs, c3_runs, var):
    """TerminateOnFirstHit taken literally: hit/miss agrees with the closest-hit contract except for a handful of grazing
    rays (the north star's first criterion holds), but the distance-weighted AO value does NOT stay inside 0.5 % / 1/255 --
    it depends on which hit the traversal meets first, i.e. on the driver.  The closest hit is one legal outcome; no
    tolerance on the AO VALUE can hold across all of them, which is why parity stays 'unpinned' (DESIGN.md 2, 7)."""
    for tag, (runs, n_rays), ref, got in (("c0 frame 0", c0_runs, c0_runs[0][0][0], c0_runs[0][var][0]),
                                          ("c3 band", c3_runs, c3_runs[0][0], c3_runs[0][var])):
        full, hist, ray_bits = _report(tag, var, ref, got, n_rays)
        assert ray_bits <= 2.0e-4 * n_rays, "hit/miss must agree except for grazing rays"
        assert hist["frac_over_1_255"] > TOL_FRAC, "if this ever holds, the AO value has become order-independent: update DESIGN.md 2"


def test_two_legal_traversal_orders_differ_from_each_other(c0_runs):
    runs, n_rays = c0_runs
    full, hist, ray_bits = _report("c0 frame 0, first-found forward vs reverse order", "4 vs 12", runs[4][0], runs[12][0], n_rays)
    assert hist["frac_over_1_255"] > TOL_FRAC
