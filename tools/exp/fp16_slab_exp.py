#!/usr/bin/env python3
"""Offline experiment (GPU box; numpy after the BVH download): what a packed-fp16 slab test would cost K1 in extra node visits.

The fp16 form of the node step (bytes -> fp16 through v_perm with the 0x64 magic byte, one v_pk_fma_f16 per plane pair, packed
min/max) saves about 20 of the ~130 VALU instructions of a node step, but it is only conservative with wider boxes: the magic-number
conversion leaves (1024 + q) * a + (b - 1024 a) with an absolute error of up to ~0.6 cell per plane, and the distances themselves
carry 2^-11 relative error.  This script counts node visits and leaf items of AO-like rays (8 per valid pixel of sampled 8x8 tiles of
the C3 frame, length = radius) against the wide tree (a) as it is and (b) with every plane moved outward by PAD cells and the
[tnear, tfar] interval widened by REL relatively -- the padding that form needs.  No closest-hit culling on either side."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from vulkanrenderer_amd import make_bistro_class, FrameState, camera_path
from vulkanrenderer_amd.renderer import Context, RayTracingContext, GBuffer, generate_gbuffer

PAD = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
REL = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0 ** -9
scene = make_bistro_class()
W, H = 1920, 1080
ctx = Context(W, H); rt = RayTracingContext(ctx); rt.init(scene)
gb = GBuffer(ctx, with_shading_inputs=True)
fs = FrameState(W, H)
eye, tgt = camera_path(scene, 8, step=0.05)[7]
ubo = fs.update(eye, tgt)
generate_gbuffer(ctx, ubo, gb); ctx.sync()
wp = gb.world_pos.cpu().numpy().reshape(H, W, 3).astype(np.float64)
nrm = gb.geom_normal.cpu().numpy().reshape(H, W, 3).astype(np.float64)
valid = gb.valid.cpu().numpy().reshape(H, W) > 0
nodes, tris = rt.download()
nu = nodes.view(np.uint32)
N = len(nodes)
org = nodes[:, 0:3].astype(np.float64)
sc = np.stack([nodes[:, 10], nodes[:, 11], nodes[:, 3]], axis=1).astype(np.float64)


def planes(cols):
    out = np.zeros((N, 4, 3))
    for a, col in enumerate(cols):
        w = nu[:, col]
        for c in range(4):
            out[:, c, a] = ((w >> (8 * c)) & 0xff).astype(np.float64)
    return out


qmin, qmax = planes([4, 5, 6]), planes([7, 8, 9])
refs = nu[:, 12:16].copy()
is_leaf = (refs & 0x80000000) != 0
is_none = refs == 0x7fffffff


def traverse(ro, rd, tmax, pad, rel):
    bmin = org[:, None, :] + (qmin - pad) * sc[:, None, :]
    bmax = org[:, None, :] + (qmax + pad) * sc[:, None, :]
    inv = 1.0 / np.where(np.abs(rd) < 1e-12, 1e-12, rd)
    rays, nds = np.arange(len(ro)), np.zeros(len(ro), dtype=np.int64)
    visits = leaves = 0
    while len(rays):
        visits += len(rays)
        t0 = (bmin[nds] - ro[rays][:, None, :]) * inv[rays][:, None, :]
        t1 = (bmax[nds] - ro[rays][:, None, :]) * inv[rays][:, None, :]
        tn = np.minimum(t0, t1).max(axis=2); tf = np.maximum(t0, t1).min(axis=2)
        tn = tn - np.abs(tn) * rel; tf = tf + np.abs(tf) * rel
        hit = (np.maximum(tn, 0.0) <= np.minimum(tf, tmax[rays][:, None])) & ~is_none[nds]
        leaves += int((hit & is_leaf[nds]).sum())
        r_idx, c_idx = np.nonzero(hit & ~is_leaf[nds])
        rays, nds = rays[r_idx], refs[nds[r_idx], c_idx].astype(np.int64)
    return visits, leaves


rng = np.random.default_rng(3)
tot = [0, 0, 0, 0, 0]
tiles = tries = 0
while tiles < 120 and tries < 5000:
    tries += 1
    ty, tx = rng.integers(0, H // 8), rng.integers(0, W // 8)
    v = valid[ty * 8:(ty + 1) * 8, tx * 8:(tx + 1) * 8]
    if v.sum() < 32: continue
    tiles += 1
    p = wp[ty * 8:(ty + 1) * 8, tx * 8:(tx + 1) * 8][v]
    n = nrm[ty * 8:(ty + 1) * 8, tx * 8:(tx + 1) * 8][v]
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-9)
    P = len(p)
    u1, u2 = rng.random((P, 8)), rng.random((P, 8))
    r = np.sqrt(u1); phi = 2 * np.pi * u2
    lx, ly, lz = r * np.cos(phi), r * np.sin(phi), np.sqrt(np.maximum(0, 1 - u1))
    a = np.where(np.abs(n[:, 0:1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    t = np.cross(a, n); t /= np.linalg.norm(t, axis=1, keepdims=True); b = np.cross(n, t)
    rd = (lx[..., None] * t[:, None, :] + ly[..., None] * b[:, None, :] + lz[..., None] * n[:, None, :]).reshape(-1, 3)
    ro = np.repeat(p + n * 1e-3, 8, axis=0)
    tmax = np.full(len(ro), 1.2 * 1.3)
    v0, l0 = traverse(ro, rd, tmax, 0.0, 0.0)
    v1, l1 = traverse(ro, rd, tmax, PAD, REL)
    tot[0] += len(ro); tot[1] += v0; tot[2] += l0; tot[3] += v1; tot[4] += l1
R = tot[0]
print(f"fp16 slab padding experiment: PAD {PAD} cells, REL {REL:.2e}, tiles {tiles}, rays {R}")
print(f"  as built : {tot[1] / R:.2f} node visits per ray, {tot[2] / R:.2f} leaf items per ray")
print(f"  padded   : {tot[3] / R:.2f} node visits per ray ({100.0 * (tot[3] / tot[1] - 1):+.1f} %), {tot[4] / R:.2f} leaf items per ray ({100.0 * (tot[4] / tot[2] - 1):+.1f} %)")
