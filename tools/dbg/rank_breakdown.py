"""Development aid (GPU box): per-phase cost of one rank's share of the C4 frame at N ranks (K1 stripes / denoise band / sun band)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vulkanrenderer_amd import FrameState, camera_path, make_bistro_class, bands
from vulkanrenderer_amd.renderer import Context, GBuffer, RayTracingContext, RtaoComputePass, ShadowRays, generate_gbuffer
W, H = 3840, 2160
scene = make_bistro_class()
ctx = Context(W, H); rt = RayTracingContext(ctx); rt.init(scene)
fs = FrameState(W, H, ray_count=8, sun_samples=16)
ubos = [fs.update(e, t) for e, t in camera_path(scene, 4)]
gb = GBuffer(ctx, with_shading_inputs=True)
rp = RtaoComputePass(ctx, gb, atrous_iterations=4)
sh = ShadowRays(ctx, gb)
generate_gbuffer(ctx, ubos[0], gb); ctx.sync()
def timed(fn):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) * 1e3)
    return best
for n in (1, 2, 4, 8):
    r = n // 2
    y0, y1 = bands.band_rows(H, n, r)
    rp.band = None
    def full():
        rp.traceStripes(ubos[1], r, n); rp.denoiseBand(ubos[1], y0, y1); sh.sun(ubos[1], rows=(y0, y1))
    full()
    t_all = timed(full)
    side = torch.cuda.Stream()
    ev = torch.cuda.Event()
    def full_side():
        rp.traceStripes(ubos[1], r, n)
        with torch.cuda.stream(side):
            sh.sun(ubos[1], rows=(y0, y1)); ev.record(side)
        rp.denoiseBand(ubos[1], y0, y1)
        torch.cuda.current_stream().wait_event(ev)
    full_side()
    t_side = timed(full_side)
    t_k1 = timed(lambda: rp.traceStripes(ubos[1], r, n))
    rp.traceStripes(ubos[1], r, n)
    t_den = timed(lambda: rp.denoiseBand(ubos[1], y0, y1))
    t_sun = timed(lambda: sh.sun(ubos[1], rows=(y0, y1)))
    print(f"N={n} rank {r}: all {t_all:.3f} ms (sun on a 2nd stream {t_side:.3f}) | K1 stripes {t_k1:.3f} | denoise band {t_den:.3f} | sun band {t_sun:.3f}", flush=True)
