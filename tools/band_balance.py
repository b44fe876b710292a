#!/usr/bin/env python3
"""GPU box: per-band cost of the C4 frame (3840x2160, 8 AO + 16 sun rays) for N = 2, 4, 8 row bands, run one band at a
time on one GPU -> load imbalance max/mean that a strong-scaling run will see."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanrenderer_amd import FrameState, camera_path, make_bistro_class, bands
from vulkanrenderer_amd.renderer import Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass, ShadowRays, generate_gbuffer
W, H = 3840, 2160
scene = make_bistro_class()
ctx = Context(W, H); rt = RayTracingContext(ctx); rt.init(scene)
fs = FrameState(W, H, ray_count=8, sun_samples=16)
ubos = [fs.update(e, t) for e, t in camera_path(scene, 4)]
gb = GBuffer(ctx, with_shading_inputs=True)
rp = RtaoComputePass(ctx, gb, atrous_iterations=4)
sh = ShadowRays(ctx, gb)
generate_gbuffer(ctx, ubos[0], gb); ctx.sync()
rows_valid = gb.valid.view(H, W).sum(dim=1).cpu().numpy()
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
NS = [int(x) for x in os.environ.get('VKR_BANDS', '1,2,4,8').split(',')]
for n in NS:
    ts = []
    for r in range(n):
        y0, y1 = bands.band_rows(H, n, r)
        rp.band = (y0, y1) if n > 1 else None
        def run():
            if os.environ.get("VKR_SCHEDULE", "stripes") == "bands" or n == 1:
                rp.execute(PassExecuteContext(ubo=ubos[1]))
            else:
                rp.band = None
                rp.traceStripes(ubos[1], r, n)
                rp.denoiseBand(ubos[1], y0, y1)
            sh.sun(ubos[1], rows=(y0, y1))
        run()
        ts.append(min(timed(run) for _ in range(3)))
    ts = np.array(ts)
    print(f"N={n}: per-band ms {np.round(ts, 2).tolist()}  max {ts.max():.2f} mean {ts.mean():.2f} imbalance {ts.max() / ts.mean():.2f}  speedup vs N=1 {'' if n == 1 else ''}", flush=True)
    vb = [int(rows_valid[bands.band_rows(H, n, r)[0]:bands.band_rows(H, n, r)[1]].sum()) for r in range(n)]
    print("   valid px per band", vb, flush=True)
