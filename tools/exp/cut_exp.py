#!/usr/bin/env python3
"""Offline experiment (GPU box; numpy after the BVH download): what a per-tile *cut* of the wide BVH would save K1.
For sample 8x8 tiles of the C3 frame: AO-like rays (8 per valid pixel, cosine hemisphere, length = radius); node visits
per ray (a) from the root, (b) from a cut = the nodes reached by descending, from the root, through every child whose box
overlaps the AABB of the tile's ray segments, until the cut would exceed FMAX entries or the nodes are no larger than the
AABB.  No closest-hit culling in either variant (upper bounds on the visits, same rule for both)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from vulkanrenderer_amd import make_bistro_class, FrameState, camera_path
from vulkanrenderer_amd.renderer import Context, RayTracingContext, GBuffer, generate_gbuffer

FMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SMALL = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0     # stop descending below half-area <= SMALL * AABB half-area
TILE = int(sys.argv[3]) if len(sys.argv) > 3 else 8
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
bmin = org[:, None, :] + planes([4, 5, 6]) * sc[:, None, :]
bmax = org[:, None, :] + planes([7, 8, 9]) * sc[:, None, :]
refs = nu[:, 12:16].copy()
NONE = 0x7fffffff
is_leaf = (refs & 0x80000000) != 0
is_none = refs == NONE
harea = lambda mn, mx: (lambda d: d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0])(np.maximum(mx - mn, 0))

def traverse(ro, rd, tmax, start_nodes, start_rays):
    """pairs (ray, node) -> total node visits and leaf items"""
    inv = 1.0 / np.where(np.abs(rd) < 1e-12, 1e-12, rd)
    rays, nds = start_rays, start_nodes
    visits = 0; leaves = 0
    while len(rays):
        visits += len(rays)
        mn, mx = bmin[nds], bmax[nds]                    # (P,4,3)
        t0 = (mn - ro[rays][:, None, :]) * inv[rays][:, None, :]
        t1 = (mx - ro[rays][:, None, :]) * inv[rays][:, None, :]
        tn = np.minimum(t0, t1).max(axis=2); tf = np.maximum(t0, t1).min(axis=2)
        hit = (np.maximum(tn, 0.0) <= np.minimum(tf, tmax[rays][:, None])) & ~is_none[nds]
        lf = hit & is_leaf[nds]
        leaves += int(lf.sum())
        go = hit & ~is_leaf[nds]
        r_idx, c_idx = np.nonzero(go)
        rays, nds = rays[r_idx], refs[nds[r_idx], c_idx].astype(np.int64)
    return visits, leaves

rng = np.random.default_rng(3)
tot = dict(rays=0, base=0, base_leaves=0, cut=0, cut_leaves=0, inject=0, F=[], rounds=[], cutvisits=0)
tiles = 0
tries = 0
while tiles < 120 and tries < 5000:
    tries += 1
    ty, tx = rng.integers(0, H // TILE), rng.integers(0, W // TILE)
    v = valid[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
    if v.sum() < TILE * TILE // 2: continue
    tiles += 1
    p = wp[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE][v]
    n = nrm[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE][v]
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-9)
    P = len(p)
    # 8 cosine-weighted directions per pixel
    u1, u2 = rng.random((P, 8)), rng.random((P, 8))
    r = np.sqrt(u1); phi = 2 * np.pi * u2
    lx, ly, lz = r * np.cos(phi), r * np.sin(phi), np.sqrt(np.maximum(0, 1 - u1))
    a = np.where(np.abs(n[:, 0:1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    t = np.cross(a, n); t /= np.linalg.norm(t, axis=1, keepdims=True); b = np.cross(n, t)
    rd = (lx[..., None] * t[:, None, :] + ly[..., None] * b[:, None, :] + lz[..., None] * n[:, None, :]).reshape(-1, 3)
    ro = np.repeat(p + n * 1e-3, 8, axis=0)
    R = 1.2 * 1.3
    tmax = np.full(len(ro), R)
    nr = len(ro)
    bv, bl = traverse(ro, rd, tmax, np.zeros(nr, dtype=np.int64), np.arange(nr))
    # cut
    end = ro + rd * (R * 1.01)
    amn, amx = np.minimum(ro, end).min(axis=0) - 1e-4, np.maximum(ro, end).max(axis=0) + 1e-4
    aarea = harea(amn, amx)
    inv = 1.0 / np.where(np.abs(rd) < 1e-12, 1e-12, rd)
    starts_r, starts_n = [], []
    open_nodes = np.array([0], dtype=np.int64); fn = 0; leaves_hit = 0; rounds = 0; cutvisits = 0
    while len(open_nodes):
        rounds += 1; cutvisits += len(open_nodes)
        mn, mx = bmin[open_nodes], bmax[open_nodes]
        ov = (mn <= amx).all(axis=2) & (mx >= amn).all(axis=2) & ~is_none[open_nodes]
        lf = ov & is_leaf[open_nodes]
        inner = ov & ~is_leaf[open_nodes]
        small = inner & (harea(mn, mx) <= SMALL * aarea)
        big = inner & ~small
        if fn + int(lf.sum() + small.sum()) + int(big.sum()) > FMAX:
            # stop: the open nodes themselves join the cut (their own boxes were tested by the round that opened them;
            # here every ray starts at each of them: pessimistic)
            for nd in open_nodes:
                starts_r.append(np.arange(nr)); starts_n.append(np.full(nr, nd, dtype=np.int64))
            fn += len(open_nodes)
            break
        for sel, leaf in ((small, False), (lf, True)):
            ri, ci = np.nonzero(sel)
            for k in range(len(ri)):
                bmn, bmx = mn[ri[k], ci[k]], mx[ri[k], ci[k]]
                t0 = (bmn - ro) * inv; t1 = (bmx - ro) * inv
                tn = np.minimum(t0, t1).max(axis=1); tf = np.maximum(t0, t1).min(axis=1)
                h = np.nonzero(np.maximum(tn, 0) <= np.minimum(tf, tmax))[0]
                fn += 1
                if leaf: leaves_hit += len(h)
                else:
                    starts_r.append(h); starts_n.append(np.full(len(h), refs[open_nodes[ri[k]], ci[k]], dtype=np.int64))
        ri, ci = np.nonzero(big)
        open_nodes = refs[open_nodes[ri], ci].astype(np.int64)
    cv = cl = 0
    if starts_r:
        cv, cl = traverse(ro, rd, tmax, np.concatenate(starts_n), np.concatenate(starts_r))
    tot['F'].append(fn); tot['rounds'].append(rounds); tot['cutvisits'] += cutvisits
    tot['rays'] += nr; tot['base'] += bv; tot['base_leaves'] += bl; tot['inject'] += fn * nr
    tot['cut'] += cv; tot['cut_leaves'] += cl + leaves_hit
R_ = tot['rays']
print(f"FMAX {FMAX} SMALL {SMALL} TILE {TILE}: tiles {tiles}, rays {R_}")
print(f"  from the root : {tot['base'] / R_:.2f} node visits per ray, {tot['base_leaves'] / R_:.2f} leaf items per ray")
print(f"  from the cut  : {tot['cut'] / R_:.2f} node visits per ray, {tot['cut_leaves'] / R_:.2f} leaf items per ray, "
      f"{tot['inject'] / R_:.1f} frontier box tests per ray; cut size mean {np.mean(tot['F']):.1f} (p90 {np.percentile(tot['F'], 90):.0f}), "
      f"rounds mean {np.mean(tot['rounds']):.1f}, cut-build node visits per tile {tot['cutvisits'] / tiles:.1f}")
