#!/usr/bin/env python3
"""Generates tests/golden/rtao_64x36.npz with the CPU oracle (SURVEY.md 8c pin 10).

The reference cannot run here and holds no golden vectors of its own (parity unpinned), so these fixtures freeze
the oracle's current behaviour: they catch any drift of the oracle, and the GPU tests check the HIP path against the
same bytes without needing the oracle at all.   Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O                                   # noqa: E402
from vulkanrenderer_amd import FrameState, camera_path, make_street_scene   # noqa: E402

SEED, DETAIL, W, H, RAYS, ITERS, FRAMES = 20261111, 0.02, 64, 36, 4, 2, 2


def generate():
    scene = make_street_scene(SEED, DETAIL, "golden")
    osc = O.OracleScene(scene)
    out = {"meta": np.array([SEED, W, H, RAYS, ITERS, FRAMES], dtype=np.int64), "detail": np.array([DETAIL])}
    fs = FrameState(W, H, ray_count=RAYS)
    p = O.OracleRtaoPass(osc, W, H, ITERS)
    for f, (eye, tgt) in enumerate(camera_path(scene, FRAMES)):
        ubo = fs.update(eye, tgt)
        depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
        full, mask = p.execute(ubo, depth, normal, want_mask=True)
        out[f"ubo{f}"], out[f"depth{f}"], out[f"normal{f}"] = ubo, depth, normal
        out[f"world_pos{f}"], out[f"geom_normal{f}"], out[f"valid{f}"] = wp, ng, valid
        out[f"rtao_full{f}"], out[f"hit_mask{f}"], out[f"history{f}"] = full, mask, p.hist[p.last_curr].copy()
        out[f"sun{f}"] = osc.shadow_sun(wp, ng, valid, FrameState.sun_dir_to_light(ubo), 0.01, 8)
        out[f"point{f}"] = osc.shadow_point(wp, ng, valid, ubo[60:63])
        # debug taps of K1 (bias / footprint / current) on the same inputs, history all ones
        one = np.full((H, W), np.float16(1.0).view(np.uint16), dtype=np.uint16)
        for mode in (10, 11, 12):
            u = ubo.copy()
            u[95] = float(mode)
            out[f"tap{mode}_{f}"] = osc.trace(u, depth, one)
    return out


if __name__ == "__main__":
    data = generate()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rtao_64x36.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes")
