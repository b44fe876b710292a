#!/usr/bin/env python3
"""Diagnostics (GPU box): child-count and leaf-size histograms of the wide LBVH."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanrenderer_amd import make_bistro_class
from vulkanrenderer_amd.renderer import Context, RayTracingContext
scene = make_bistro_class()
ctx = Context(64, 64); rt = RayTracingContext(ctx); rt.init(scene)
nodes, tris = rt.download()
nu = nodes.view(np.uint32)
refs = nu[:, 12:16].view(np.int32)
used = refs != 0x7fffffff
nch = used.sum(axis=1)
print("nodes", len(nodes), "children/node hist", np.bincount(nch, minlength=5), "mean", nch.mean())
leaf = used & (refs < 0)
x = refs[leaf].astype(np.uint32)          # leaf: bit 31 set, bits 30..1 first triangle, bit 0 count - 1
cnt = (x & 1) + 1
print("leaves", leaf.sum(), "tris/leaf hist", np.bincount(cnt, minlength=5), "mean", cnt.mean())
nl = leaf.sum(axis=1); ni = (used & (refs >= 0)).sum(axis=1)
print("nodes with only leaf children", (ni == 0).sum(), "their child hist", np.bincount(nch[ni == 0], minlength=5))
