"""Worker of tests/test_bands_gloo.py: one rank of the row-band schedule, on the CPU oracle over gloo.

Mirrors what bench.py does per frame on N GPUs (band + halo through K1/K2/K3, ONE packed all-gather of
[history | rtao_full | sun], unpack) with the oracle standing in for the HIP kernels, and checks on every rank that
the assembled frame equals the unsharded oracle frame bit for bit.  Usage: python _band_worker.py OUT_DIR
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O                                            # noqa: E402
from vulkanrenderer_amd import FrameState, bands, camera_path, make_test_scene   # noqa: E402

W, H, RAYS, ITERS, FRAMES = 48, 37, 2, 3, 3      # ragged height: the last band is short


def band_frame(osc, ubo, depth, normal, hist_prev, y0, y1, iters):
    """vkr_rtao_execute_band's schedule (vkr_api.hip) on the oracle.  Returns (history rows written, lo, hi, full)."""
    Hh = depth.shape[0]
    lo = lambda h: max(0, y0 - h)                   # noqa: E731
    hi = lambda h: min(Hh, y1 + h)                  # noqa: E731
    halo_all = bands.atrous_halo(iters)
    cur = osc.trace(ubo, depth, hist_prev, lo(halo_all), hi(halo_all))
    src = cur
    for i in range(iters):
        halo_out = 2 * ((1 << iters) - (1 << (i + 1)))
        src = O.atrous(ubo, 1 << i, src, depth, normal, lo(halo_out), hi(halo_out))
    full = O.upsample(ubo, src, normal)
    return cur, lo(halo_all), hi(halo_all), full


def stripe_frame(osc, ubo, depth, normal, hist_prev, hist_curr, rank, world, plan, send, recv, y0, y1, iters):
    """The two-phase schedule of bench.py (vkr_rtao_trace_stripes -> gather history -> vkr_rtao_denoise_band) on the
    oracle.  hist_curr is completed in place; returns rtao_full with rows [y0, y1) filled."""
    Hh, Ww = depth.shape
    total = (Hh + bands.STRIPE - 1) // bands.STRIPE
    for s in range(rank, total, world):                    # dispatchTrace on this rank's stripes only
        a, b = s * bands.STRIPE, min(Hh, (s + 1) * bands.STRIPE)
        hist_curr[a:b] = osc.trace(ubo, depth, hist_prev, a, b)[a:b]
    t = torch.from_numpy(hist_curr.view(np.int16))
    plan.pack(t, rank, send)
    bands.all_gather_bytes(send, recv)
    plan.unpack(recv, t)
    lo = lambda h: max(0, y0 - h)                   # noqa: E731
    hi = lambda h: min(Hh, y1 + h)                  # noqa: E731
    src = hist_curr
    for i in range(iters):
        halo_out = 2 * ((1 << iters) - (1 << (i + 1)))
        src = O.atrous(ubo, 1 << i, src, depth, normal, lo(halo_out), hi(halo_out))
    return O.upsample(ubo, src, normal)


# ---- the C-ABI collective (vulkanrenderer_amd/csrc/comm.hip) on the CPU: the index arithmetic of its four kernels, line by line,
# with gloo carrying the bytes -- what bench.py's default N > 1 path runs per frame:
#     vkr_rtao_trace_stripes -> vkr_rtao_gather_history -> vkr_rtao_denoise_band -> vkr_gather_bands([rtao_full, sun])
def capi_stripes_per_rank(Hh, n):
    return (((Hh + 7) // 8) + n - 1) // n


def capi_pack_stripes(plane, rank, n):                       # k_pack_stripes
    Hh = plane.shape[0]
    j = np.arange(capi_stripes_per_rank(Hh, n) * 8)
    row = np.minimum(((j >> 3) * n + rank) * 8 + (j & 7), Hh - 1)
    return np.ascontiguousarray(plane[row])


def capi_unpack_stripes(recv, plane, n):                     # k_unpack_stripes: every image row exactly once
    row = np.arange(plane.shape[0])
    s = row >> 3
    plane[row] = recv[s % n, (s // n) * 8 + (row & 7)]


def capi_pack_bands(planes, rank, n):                        # k_pack_bands: slot = [plane][Hb rows][W], pad rows stay as they are
    Hh, Ww = planes[0].shape
    hb = (Hh + n - 1) // n
    send = np.full((len(planes), hb, Ww), 0x5a5a, dtype=np.uint16)
    for k, p in enumerate(planes):
        rows = np.arange(rank * hb, min(Hh, (rank + 1) * hb))
        send[k, : len(rows)] = p[rows]
    return send


def capi_unpack_bands(recv, planes, n):                      # k_unpack_bands
    Hh = planes[0].shape[0]
    hb = (Hh + n - 1) // n
    row = np.arange(Hh)
    for k, p in enumerate(planes):
        p[row] = recv[row // hb, k, row - (row // hb) * hb]


def gloo_gather(send):
    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(send).view(np.uint8).reshape(-1))
    out = torch.empty(world * t.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, t)
    return out.numpy().view(np.uint16).reshape((world,) + send.shape)


def capi_frame(osc, ubo, depth, normal, hist_prev, hist_curr, rank, world, y0, y1, iters, sun_band):
    Hh, Ww = depth.shape
    for s in range(rank, (Hh + 7) // 8, world):             # vkr_rtao_trace_stripes
        a, b = s * 8, min(Hh, (s + 1) * 8)
        hist_curr[a:b] = osc.trace(ubo, depth, hist_prev, a, b)[a:b]
    capi_unpack_stripes(gloo_gather(capi_pack_stripes(hist_curr, rank, world)), hist_curr, world)    # vkr_rtao_gather_history
    lo = lambda h: max(0, y0 - h)                   # noqa: E731
    hi = lambda h: min(Hh, y1 + h)                  # noqa: E731
    src = hist_curr
    for i in range(iters):                                  # vkr_rtao_denoise_band
        halo_out = 2 * ((1 << iters) - (1 << (i + 1)))
        src = O.atrous(ubo, 1 << i, src, depth, normal, lo(halo_out), hi(halo_out))
    full = np.full((Hh, Ww), 0x4321, dtype=np.uint16)       # rows of other ranks must arrive through the gather
    full[y0:y1] = O.upsample(ubo, src, normal)[y0:y1]
    sun = np.full((Hh, Ww), 0x4321, dtype=np.uint16)
    sun[y0:y1] = sun_band.view(np.uint16)
    planes = [full, sun]
    capi_unpack_bands(gloo_gather(capi_pack_bands(planes, rank, world)), planes, world)               # vkr_gather_bands
    return full, sun


def main():
    out_dir = sys.argv[1]
    schedule = sys.argv[2] if len(sys.argv) > 2 else "bands"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O.set_threads(2)
    scene = make_test_scene()
    osc = O.OracleScene(scene)
    fs = FrameState(W, H, ray_count=RAYS)
    ref = O.OracleRtaoPass(osc, W, H, ITERS)
    one = np.float16(1.0).view(np.uint16)
    hist = [np.full((H, W), one, dtype=np.uint16) for _ in range(2)]
    parity = 0
    y0, y1 = bands.band_rows(H, world, rank)
    hb = bands.band_height(H, world)
    packed = torch.zeros((bands.PLANES, hb, W), dtype=torch.int16)
    gathered = torch.zeros((world, bands.PLANES, hb, W), dtype=torch.int16)
    plan = bands.StripePlan(H, world)
    h_send = torch.zeros((plan.slot_rows, W), dtype=torch.int16)
    h_recv = torch.zeros((world, plan.slot_rows, W), dtype=torch.int16)
    ok = True
    for f, (eye, tgt) in enumerate(camera_path(scene, FRAMES)):
        ubo = fs.update(eye, tgt)
        depth, normal, wp, ng, valid = osc.gbuffer(ubo, W, H)
        want_full = ref.execute(ubo, depth, normal)
        want_hist = ref.hist[ref.last_curr]
        want_sun = osc.shadow_sun(wp, ng, valid, FrameState.sun_dir_to_light(ubo), 0.01, 4).astype(np.float16)

        if schedule == "capi":
            hist[parity][:] = 0x1234
            sun_band = osc.shadow_sun(wp[y0:y1], ng[y0:y1], valid[y0:y1], FrameState.sun_dir_to_light(ubo), 0.01, 4).astype(np.float16)
            full, sun16 = capi_frame(osc, ubo, depth, normal, hist[parity ^ 1], hist[parity], rank, world, y0, y1, ITERS, sun_band)
            ok &= np.array_equal(hist[parity], want_hist)
            ok &= np.array_equal(full, want_full)
            ok &= np.array_equal(sun16, want_sun.view(np.uint16))
            parity ^= 1
            continue
        if schedule == "stripes":
            hist[parity][:] = 0x1234                          # rows of other ranks must arrive through the gather
            full = stripe_frame(osc, ubo, depth, normal, hist[parity ^ 1], hist[parity], rank, world, plan, h_send, h_recv,
                                y0, y1, ITERS)
        else:
            cur, lo, hi, full = band_frame(osc, ubo, depth, normal, hist[parity ^ 1], y0, y1, ITERS)
            hist[parity][lo:hi] = cur[lo:hi]
        sun = np.zeros((H, W), dtype=np.float16)
        sun[y0:y1] = osc.shadow_sun(wp[y0:y1], ng[y0:y1], valid[y0:y1], FrameState.sun_dir_to_light(ubo), 0.01,
                                    4).astype(np.float16)
        rt_full = np.zeros((H, W), dtype=np.uint16)
        rt_full[y0:y1] = full[y0:y1]
        planes = [torch.from_numpy(hist[parity].view(np.int16)), torch.from_numpy(rt_full.view(np.int16)),
                  torch.from_numpy(sun.view(np.int16))]
        bands.pack_band(planes, H, world, rank, packed)
        bands.all_gather_frame(packed, gathered)
        bands.unpack_frame(gathered, planes, H, world)

        ok &= np.array_equal(hist[parity], want_hist)
        ok &= np.array_equal(rt_full, want_full)
        ok &= np.array_equal(sun.view(np.uint16), want_sun.view(np.uint16))
        parity ^= 1
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as fh:
        fh.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
