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
