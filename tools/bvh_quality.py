#!/usr/bin/env python3
"""Diagnostics (GPU box): per-level overlap of the wide LBVH at sample surface points.
For each tree level: how many child boxes contain a sample point inflated by the AO radius (ideal: ~1-2)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanrenderer_amd import make_bistro_class, make_sponza_class, FrameState, camera_path
from vulkanrenderer_amd.renderer import Context, RayTracingContext, GBuffer, generate_gbuffer

scene = make_bistro_class() if len(sys.argv) < 2 or sys.argv[1] == "bistro" else make_sponza_class()
W, H = 480, 270
ctx = Context(W, H); rt = RayTracingContext(ctx); rt.init(scene)
gb = GBuffer(ctx)
fs = FrameState(W, H)
eye, tgt = camera_path(scene, 1)[0]
ubo = fs.update(eye, tgt)
generate_gbuffer(ctx, ubo, gb); ctx.sync()
wp = gb.world_pos.cpu().numpy().reshape(-1, 3)
valid = gb.valid.cpu().numpy().reshape(-1) > 0
rng = np.random.default_rng(1)
pts = wp[valid][rng.choice(valid.sum(), 256, replace=False)].astype(np.float64)
R = 1.2
nodes, tris = rt.download()
nu = nodes.view(np.uint32)
N = len(nodes)
org = nodes[:, 0:3].astype(np.float64)
sc = np.stack([nodes[:, 10], nodes[:, 11], nodes[:, 3]], axis=1).astype(np.float64)   # scale x, y | z stored as floats
def planes(wordcols):
    out = np.zeros((N, 4, 3))
    for a, col in enumerate(wordcols):
        w = nu[:, col]
        for c in range(4):
            out[:, c, a] = ((w >> (8 * c)) & 0xff).astype(np.float64)
    return out
bmin = org[:, None, :] + planes([4, 5, 6]) * sc[:, None, :]
bmax = org[:, None, :] + planes([7, 8, 9]) * sc[:, None, :]
refs = nu[:, 12:16].view(np.int32)
# levels by BFS
level = np.full(N, -1, dtype=np.int32); level[0] = 0
frontier = np.array([0])
L = 0
while len(frontier):
    ch = refs[frontier].reshape(-1)
    ch = ch[(ch >= 0) & (ch != 0x7fffffff)]
    L += 1
    level[ch] = L
    frontier = ch
print("nodes", N, "levels", level.max() + 1, "tris", len(tris))
tm = np.minimum(np.minimum(tris[:, 0:3], tris[:, 0:3] + tris[:, 4:7]), tris[:, 0:3] + tris[:, 8:11])
tM = np.maximum(np.maximum(tris[:, 0:3], tris[:, 0:3] + tris[:, 4:7]), tris[:, 0:3] + tris[:, 8:11])
ext = (tM - tm).max(axis=1)
print("triangle max-extent percentiles (m): 50%% %.3f 90%% %.3f 99%% %.3f 99.9%% %.3f max %.2f" % tuple(np.percentile(ext, [50, 90, 99, 99.9, 100])))
print("scene bounds", tm.min(axis=0), tM.max(axis=0))
tot = 0.0
for l in range(level.max() + 1):
    idx = np.where(level == l)[0]
    mn, mx = bmin[idx], bmax[idx]          # (n,4,3)
    used = (refs[idx] != 0x7fffffff)
    cnt_box = 0; cnt_node = 0
    for p in pts:
        ov = np.all((mn <= p + R) & (mx >= p - R), axis=2) & used      # (n,4): child boxes within reach of the point
        cnt_box += ov.sum()
    isleaf = refs[idx] < 0
    print(f"level {l:2d}: nodes {len(idx):7d}  child boxes within {R} m of a surface point: {cnt_box / len(pts):7.2f}  leaf-children share {isleaf[used].mean():.2f}  mean child extent {np.mean((mx - mn)[used]):8.3f}")
    tot += cnt_box / len(pts)
print("sum over levels (upper bound on child boxes a 1.2 m neighbourhood touches):", tot)
