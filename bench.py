#!/usr/bin/env python3
"""bench.py -- headline benchmark of the RTAO / shadow-ray path (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one frame of the hot path over resident inputs:
  workload c3 (default at N=1, BASELINE configs[3]): bistro-class scene, 1920x1080, 8 AO rays/px,
      trace+temporal + 4 A-Trous iterations (+ upsample folded into the last one), camera dolly 5 cm/frame.
  workload c4 (default at N>1, BASELINE configs[4]): same scene, 3840x2160, 8 AO rays + 16 sun-disc shadow
      rays per pixel, row-band sharded over N ranks, one RCCL all-gather per frame (strong scaling).
G-buffers for all frames are generated on the GPU BEFORE the timed region (inputs resident in HBM).
value = Mrays/s = rays actually traced (AO + self-hit retries + shadow) / wall time of the K steps, max over ranks.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

T_START = time.perf_counter()

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {
    # name: (W, H, ao_rays, atrous_iters, sun_samples, scene)
    "c0": (640, 360, 4, 2, 0, "sponza"),
    "c3": (1920, 1080, 8, 4, 0, "bistro"),
    "c4": (3840, 2160, 8, 4, 16, "bistro"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="auto", choices=["auto", "c0", "c1", "c2", "c3", "c4"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip per-kernel timing / counters")
    return ap.parse_args()


def make_scene(kind):
    from vulkanrenderer_amd import make_bistro_class, make_sponza_class
    return make_bistro_class() if kind == "bistro" else make_sponza_class()


def event_time_ms(fn, stream_sync):
    """Time one enqueue of `fn` with HIP events on torch's current stream (the stream the kernels are launched on)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1)


def _oracle_chain_band(O, osc, ubo, depth, normal, hist_prev, iters, ya, yb):
    """RtaoComputePass on the CPU for output rows [ya, yb): K1 on the rows + the A-Trous halo, then `iters` A-Trous
    iterations whose row range shrinks by 2*step each (the last one = rtao_full, upsample folded).  Returns
    (rtao_full rows, history rows, seconds K1, seconds A-Trous)."""
    H, W = depth.shape
    halo = 2 * ((1 << iters) - 1)
    t0, t1 = max(0, ya - halo), min(H, yb + halo)
    a = time.perf_counter()
    cur = osc.trace(ubo, depth, hist_prev, t0, t1)
    b = time.perf_counter()
    src = np.asarray(cur).view(np.uint16).reshape(H, W)
    for i in range(iters):
        rem = 2 * ((1 << iters) - (1 << (i + 1)))          # halo rows the later iterations still need
        src = O.atrous(ubo, 1 << i, src, depth, normal, max(0, ya - rem), min(H, yb + rem))
    c = time.perf_counter()
    return src[ya:yb], np.asarray(cur).view(np.uint16).reshape(H, W)[ya:yb], b - a, c - b


def cpu_baseline(scene, wl, ubo, gb0_depth, gb0_normal, target_seconds, c0=None):
    """Time the CPU oracle (kind "port") on bounded samples of the same workload, whole chain (K1 trace+temporal + N x A-Trous,
    upsample folded into the last store): all host threads the box grants us AND one thread.
    Sample 1: a row band of the first timed C3 frame (sized for ~target_seconds).  Sample 2 (c0 = (scene, ubo, depth, normal)):
    BASELINE configs[0] in full -- sponza-class 640x360, 4 rays/px, 2 A-Trous iterations."""
    from oracle import oracle as O
    W, H, rays, iters, sun, _ = wl
    threads = min(16, len(os.sched_getaffinity(0)))
    O.set_threads(threads)
    t0 = time.perf_counter()
    osc = O.OracleScene(scene)
    build_s = time.perf_counter() - t0
    hist = np.full((H, W), np.float16(1.0).view(np.uint16), dtype=np.uint16)
    ymid = H // 2
    # calibrate on 8 rows, then size the band: ~60 % of the budget all-core, ~40 % single-thread
    t0 = time.perf_counter()
    osc.trace(ubo, gb0_depth, hist, ymid, ymid + 8)
    dt8 = time.perf_counter() - t0
    halo = 2 * ((1 << iters) - 1)
    rows = int(max(8, min(H - ymid - halo, 8 * 0.6 * target_seconds / max(dt8, 1e-6) - 2 * halo)))
    osc.stats(reset=True)
    full_rows, hist_rows, t_k1, t_at = _oracle_chain_band(O, osc, ubo, gb0_depth, gb0_normal, hist, iters, ymid, ymid + rows)
    st = osc.stats(reset=True)
    out = {"value": st["rays"] / (t_k1 + t_at) / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"oracle whole chain (K1 trace+temporal on rows [{max(0, ymid - halo)},{min(H, ymid + rows + halo)}) + {iters} A-Trous iterations "
                     f"-> rtao_full rows [{ymid},{ymid + rows})) of the first timed frame at {W}x{H}, {rays} rays/px: {st['rays']} rays, "
                     f"K1 {t_k1:.2f} s + A-Trous {t_at:.2f} s on {threads} threads; CPU BVH build {build_s:.1f} s not counted",
           "band_ms": {"rows": rows, "k1": t_k1 * 1e3, "atrous": t_at * 1e3},
           "frame_ms_extrapolated_by_rows": (t_k1 + t_at) * 1e3 * H / rows}
    # one thread: a band 1/threads as tall for about the same wall time
    O.set_threads(1)
    rows1 = int(max(4, min(rows, rows * 0.7 / threads + 1)))
    osc.stats(reset=True)
    _, _, t1_k1, t1_at = _oracle_chain_band(O, osc, ubo, gb0_depth, gb0_normal, hist, iters, ymid, ymid + rows1)
    st1 = osc.stats(reset=True)
    out["single_thread"] = {"value": st1["rays"] / (t1_k1 + t1_at) / 1e6, "unit": "Mrays/s", "cores": 1,
                            "band_ms": {"rows": rows1, "k1": t1_k1 * 1e3, "atrous": t1_at * 1e3},
                            "frame_ms_extrapolated_by_rows": (t1_k1 + t1_at) * 1e3 * H / rows1}
    if c0 is not None:
        sc0, ubo0, d0, n0 = c0
        H0, W0 = d0.shape
        o0 = O.OracleScene(sc0)
        res = {}
        for nthr in (threads, 1):
            O.set_threads(nthr)
            p0 = O.OracleRtaoPass(o0, W0, H0, 2)
            o0.stats(reset=True)
            t0 = time.perf_counter()
            p0.execute(ubo0, d0, n0)
            dt = time.perf_counter() - t0
            r0 = o0.stats(reset=True)["rays"]
            res["all_cores" if nthr != 1 else "one_core"] = {"cores": nthr, "ms_per_frame": dt * 1e3, "mrays_per_s": r0 / dt / 1e6}
        out["c0_full_frame"] = {"workload": f"c0: {sc0.name} {W0}x{H0}, 4 AO rays/px + temporal + 2 A-Trous iters (BASELINE configs[0]), one frame from cleared history", **res}
    O.set_threads(threads)
    return (ymid, ymid + rows, full_rows, hist_rows), out


def log(msg):
    if os.environ.get("VKR_BENCH_VERBOSE"):
        print(f"[bench rank {os.environ.get('RANK', '0')} +{time.perf_counter() - T_START:.1f}s] {msg}", file=sys.stderr, flush=True)


def main_c1(args):
    """Workload c1 (BASELINE configs[1]): the IBL runtime precompute at the reference's sizes --
    HDR equirect -> 512^2 RGBA32F env cube (EquirectToCubemap::convert), then IblPrecompute::compute: 32^2 irradiance
    (252 x 63 samples/texel), 128^2 GGX prefilter with all 8 mips (1024 samples/texel; shading uses mips 0..4), 512^2
    BRDF LUT (1024 samples/texel).  One-shot work, not sharded: single GPU."""
    from vulkanrenderer_amd import Context, EquirectToCubemap, IblPrecompute, make_sky_hdr
    torch.cuda.set_device(0)
    K, Wu = args.steps, args.warmup
    ctx = Context(64, 64, 0)
    sky_np = make_sky_hdr(2048, 1024)
    sky = torch.from_numpy(sky_np).to(ctx.device)
    ENV, IRR, PRE, LUT = 512, 32, 128, 512

    def step():
        cube = EquirectToCubemap.convert(ctx, sky, ENV)
        return cube, IblPrecompute.compute(ctx, cube, IRR, PRE, LUT)

    for _ in range(Wu):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        cube, res = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lib, h, st = ctx.lib, ctx.handle, ctx.stream()
    pre_texels = int(lib.vkr_ibl_prefilter_texels(PRE))
    t_eq = np.mean([event_time_ms(lambda: EquirectToCubemap.convert(ctx, sky, ENV), None) for _ in range(5)])
    t_irr = np.mean([event_time_ms(lambda: lib.vkr_ibl_irradiance(h, cube.data_ptr(), ENV, res.irradiance.data_ptr(), IRR, st), None) for _ in range(5)])
    t_pre = np.mean([event_time_ms(lambda: lib.vkr_ibl_prefilter(h, cube.data_ptr(), ENV, res.prefilter.data_ptr(), PRE, st), None) for _ in range(5)])
    t_lut = np.mean([event_time_ms(lambda: lib.vkr_ibl_brdf_lut(h, res.brdf_lut.data_ptr(), LUT, st), None) for _ in range(5)])
    samples = {"equirect": 6 * ENV * ENV, "irradiance": 6 * IRR * IRR * 252 * 63, "prefilter": pre_texels * 1024,
               "brdf_lut": LUT * LUT * 1024}
    total_samples = sum(samples.values())
    # dominant kernel by time; algorithmic bytes: 64 B of texels per bilinear cube sample (4 x RGBA32F) + 8 B per output texel
    times = {"k_ibl_irradiance": t_irr, "k_ibl_prefilter (8 mips)": t_pre, "k_ibl_brdf": t_lut, "k_ibl_equirect": t_eq}
    dom = max(times, key=times.get)
    alg = {"k_ibl_irradiance": samples["irradiance"] * 64.0 + 6 * IRR * IRR * 8.0,
           "k_ibl_prefilter (8 mips)": samples["prefilter"] * 64.0 + pre_texels * 8.0,
           "k_ibl_brdf": LUT * LUT * 4.0, "k_ibl_equirect": samples["equirect"] * (64.0 + 16.0)}[dom]
    achieved = alg / (times[dom] * 1e-3) / 1e9
    compulsory = 6 * ENV * ENV * 16.0
    out = {
        "metric": "IBL precompute Msamples/s (environment/BRDF samples integrated / wall time of equirect->cube + IblPrecompute::compute)",
        "value": total_samples * K / elapsed / 1e6, "unit": "Msamples/s", "n_gpus": 1, "steps": K, "warmup": Wu,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"c1: HDR equirect 2048x1024 -> {ENV}^2 env cube, {IRR}^2 irradiance, {PRE}^2 GGX prefilter x 8 mips, {LUT}^2 BRDF LUT",
                   "samples_per_step": total_samples, "parallelism": "single"},
        "kernels": {"equirect_ms": float(t_eq), "irradiance_ms": float(t_irr), "prefilter_ms": float(t_pre), "brdf_lut_ms": float(t_lut)},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launch_ms": float(times[dom]),
                     "algorithmic_bytes_per_launch": alg, "compulsory_bytes_per_launch": compulsory,
                     "note": "algorithmic = 64 B of RGBA32F texels per bilinear cube sample; the 25 MB environment cube is read from HBM once and then "
                             "served by L2 / Infinity Cache, so the figure is cache bandwidth, not HBM traffic: these kernels are ALU/latency bound"},
    }
    if args.cpu_seconds > 0:
        from oracle import oracle
        t1 = time.perf_counter()
        c_np = oracle.ibl_equirect_to_cube(sky_np, ENV)
        oracle.ibl_irradiance(c_np, IRR, 0)
        oracle.ibl_prefilter(c_np, PRE, 0)
        oracle.ibl_brdf_lut(LUT)
        t_cpu = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": total_samples / t_cpu / 1e6, "unit": "Msamples/s", "cores": oracle.max_threads(), "kind": "port",
                               "sample": "one full precompute of the same workload (oracle, the shaders' running-sum order)"}
    print(json.dumps(out))


def main_c2(args):
    """Workload c2 (BASELINE configs[2]): 1M-node synthetic model, per frame world matrices +
    frustum cull + DrawIndexedIndirectCommand compaction (FrameManager::prepareSharedOpaqueIndirect on the device).
    Not sharded: N > 1 would be replicas only, so this workload runs at N = 1."""
    from vulkanrenderer_amd import Context, SharedOpaqueIndirect, make_draw_model
    from vulkanrenderer_amd.scenes import look_at_rh, perspective_rh_zo_yflip, col_major
    torch.cuda.set_device(0)
    K, Wu = args.steps, args.warmup
    n_nodes, n_meshes, n_mats = 1_000_000, 20_000, 256
    m = make_draw_model(n_nodes, n_meshes, n_mats, seed=15, invalid_fraction=0.005)
    n_slots = len(m["slots"])
    ctx = Context(64, 64, 0)
    ib = SharedOpaqueIndirect(ctx)
    ib.buildSharedDrawSlots(m["node_parent"], m["node_local"], m["slots"], m["mesh_infos"], 1 << 21, mesh_bounds=m["mesh_bounds"])
    proj = perspective_rh_zo_yflip(60.0, 16 / 9, 0.1, 400.0)

    def frame_inputs(i):
        a = 0.05 * i
        vp = col_major((proj @ look_at_rh((0, 4, 0), (60 * np.sin(a), 0, 60 * np.cos(a)))).astype(np.float32))   # inside the scene, panning
        model = np.eye(4, dtype=np.float32)
        model[0, 3] = 0.01 * i
        return col_major(model), vp

    inputs = [frame_inputs(i) for i in range(K + Wu)]
    for i in range(Wu):
        ib.prepareSharedOpaqueIndirect(*inputs[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(Wu, Wu + K):
        ib.prepareSharedOpaqueIndirect(*inputs[i])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    draws = []
    t_dev = []
    for i in range(Wu, Wu + K):
        t_dev.append(event_time_ms(lambda: ib.prepareSharedOpaqueIndirect(*inputs[i]), None))
        draws.append(ib.result()[0])
    dev_ms = float(np.mean(t_dev))
    mean_draws = float(np.mean(draws))
    # algorithmic bytes of one prepare: per node parent 4 + local 64 read, world 64 written;
    # per slot 16 read + world 64 + bounds 24 read for the test; per surviving draw world 64 + info 12 read,
    # drawData 64 + command 20 written
    bytes_per_launch = n_nodes * 132.0 + n_slots * 104.0 + mean_draws * 160.0
    achieved = bytes_per_launch / (dev_ms * 1e-3) / 1e9
    out = {
        "metric": "indirect-build Mslots/s (draw slots processed by prepareSharedOpaqueIndirect / wall time)",
        "value": n_slots * K / elapsed / 1e6, "unit": "Mslots/s", "n_gpus": 1, "steps": K, "warmup": Wu,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32+u32", "data": "synthetic",
        "config": {"workload": f"c2: synthetic {n_nodes}-node model, {n_slots} draw slots, {n_meshes} meshes, {n_mats} materials: "
                               "world matrices + frustum cull + DrawIndexedIndirectCommand compaction",
                   "mean_draws_after_cull": mean_draws, "parallelism": "single"},
        "roofline": {"bound": "hbm", "kernel": "vkr_indirect_prepare (k_ib_world + k_ib_count + k_ib_emit + spans)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "launch_ms": dev_ms, "algorithmic_bytes_per_launch": bytes_per_launch},
    }
    if args.cpu_seconds > 0:
        from oracle import oracle
        slots = oracle.sort_draw_slots(m["slots"])
        reps, t_cpu = 0, 0.0
        while t_cpu < min(args.cpu_seconds, 10.0) and reps < K:
            model, vp = inputs[Wu + reps]
            t1 = time.perf_counter()
            oracle.prepare_indirect(m["node_parent"], m["node_local"], slots, m["mesh_infos"], model, 1 << 21,
                                    view_proj=vp, mesh_bounds=m["mesh_bounds"])
            t_cpu += time.perf_counter() - t1
            reps += 1
        out["cpu_baseline"] = {"value": n_slots * reps / t_cpu / 1e6, "unit": "Mslots/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} full frames of the same workload (oracle restatement of FrameManager.cpp:118-175 + frustum test)"}
    print(json.dumps(out))


def _hip_runtime():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return hip


def fs_pose_ubo(W, H, rays, sun_samples, pose):
    """A UBO for generating the G-buffer of one camera pose (the producer reads view / proj / camPos only)."""
    from vulkanrenderer_amd import FrameState
    return FrameState(W, H, ray_count=rays, sun_samples=max(1, sun_samples)).update(*pose)


def main():
    args = parse_args()
    if args.workload == "c1":
        return main_c1(args)
    if args.workload == "c2":
        return main_c2(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = max(1, args.gpus)
    # VKR_BENCH_REHEARSE=1: functional rehearsal of the N>1 path on a one-GPU box (all ranks share cuda:0, gloo
    # instead of RCCL).  Its timings mean nothing; the driver's multi-GPU runs never set it.
    rehearse = os.environ.get("VKR_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    # --gpus N is a statement about the launcher: N ranks (torch.distributed.run sets WORLD_SIZE).  Without them the N>1
    # schedule would trace 1/N of the frame and report it as the whole job, so refuse.
    if world != n:
        sys.exit(f"bench.py: --gpus {n} but WORLD_SIZE is {world}: launch with python -m torch.distributed.run --nproc-per-node {n} "
                 f"--master-addr 127.0.0.1 bench.py --gpus {n} ...")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from vulkanrenderer_amd import FrameState, camera_path
    from vulkanrenderer_amd import bands
    from vulkanrenderer_amd.renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, RtaoComputePass,
                                             ShadowRays, generate_gbuffer, trace_stats, trace_stats_enable)

    wl_name = args.workload if args.workload != "auto" else ("c3" if n == 1 else "c4")
    wl = WORKLOADS[wl_name]
    W, H, rays, iters, sun_samples, scene_kind = wl
    K, Wu = args.steps, args.warmup
    frames = K + Wu

    log(f"workload {wl_name}, generating scene")
    scene = make_scene(scene_kind)
    log("scene ready")
    if os.environ.get("VKR_BENCH_ALL_OPAQUE"):      # experiment switch: cost of the MASK/BLEND candidate path
        for m in scene.materials:
            m.alpha_mode = 0
    ctx = Context(W, H, local_rank)
    rt = RayTracingContext(ctx)
    t0 = time.perf_counter()
    rt.init(scene)
    ctx.sync()
    first_build_s = time.perf_counter() - t0
    info = rt.info()
    log(f"BVH built: {info.triangle_count} triangles")
    # LBVH build, timed separately (config names it): rebuild 3x on the resident scene, HIP-event timed
    build_ms = min(event_time_ms(rt.rebuild, None) for _ in range(3))

    fs = FrameState(W, H, ray_count=rays, sun_samples=max(1, sun_samples))
    # N = 1: one G-buffer per frame along a straight dolly (resident before the timed region).  N > 1 (4K): a 4K G-buffer with
    # its shading inputs is 290 MB, so only RESIDENT camera poses are kept and the camera dollies forth and back over them
    # (0 1 2 3 2 1 0 1 ...): the motion stays continuous -- every frame's prevViewProj is the pose rendered before it -- and
    # a rank holds 1.2 GB of inputs instead of 7 GB.
    resident = frames if n == 1 else max(2, min(frames, int(os.environ.get("VKR_BENCH_RESIDENT_GBUFFERS", "4"))))
    poses = camera_path(scene, resident, step=0.05)
    period = max(1, 2 * (resident - 1))
    pose_of = [i if n == 1 else ((i % period) if (i % period) < resident else period - (i % period)) for i in range(frames)]
    ubos = [fs.update(*poses[pose_of[i]]) for i in range(frames)]
    pose_gbs = []
    for k in range(resident):
        gb = GBuffer(ctx, with_shading_inputs=(sun_samples > 0))
        generate_gbuffer(ctx, fs_pose_ubo(W, H, rays, sun_samples, poses[k]), gb)
        pose_gbs.append(gb)
    gbs = [pose_gbs[pose_of[i]] for i in range(frames)]
    ctx.sync()

    log("G-buffers ready")
    rpass = RtaoComputePass(ctx, gbs[0], atrous_iterations=iters)
    y0, y1 = bands.band_rows(H, n, rank)
    shadows = ShadowRays(ctx, gbs[0]) if sun_samples > 0 else None
    hb = bands.band_height(H, n)
    # N > 1: two-phase schedule.  K1 on interleaved 8-row stripes (evens out the scene-dependent traversal cost and
    # needs no halo re-trace), all-gather of the history stripes, A-Trous + sun shadows on contiguous bands (their halo
    # rows come from the gathered history), all-gather of [rtao_full | sun].
    plan = bands.StripePlan(H, n, dev) if n > 1 else None
    hist_send = torch.zeros((plan.slot_rows, W), dtype=torch.int16, device=dev) if n > 1 else None
    hist_recv = torch.zeros((n, plan.slot_rows, W), dtype=torch.int16, device=dev) if n > 1 else None
    packed = torch.zeros((2, hb, W), dtype=torch.int16, device=dev)
    gathered = torch.zeros((n, 2, hb, W), dtype=torch.int16, device=dev)
    sun16 = torch.zeros((H, W), dtype=torch.float16, device=dev)

    # VKR_BENCH_CAPI_COMM=1: the same schedule with both gathers behind the C boundary (vkr_comm_attach + vkr_rtao_gather_history
    # + vkr_gather_bands: fused pack, ncclAllGather on the frame's stream, fused unpack) -- what a C++ host would run.  Under
    # RCCL the process group's own communicator is attached; in the gloo rehearsal the custom-transport hook carries it.
    # Default for N > 1: the collective behind the C boundary (what a C++ host runs), on a communication stream of its own,
    # ordered against the frame's stream by events, so that the history gather overlaps the sun kernel and the output gather
    # overlaps the next frame's K1.  VKR_BENCH_TORCH_COMM=1 selects the torch.distributed gathers (index_select packing) instead.
    capi_comm = n > 1 and os.environ.get("VKR_BENCH_TORCH_COMM") != "1"
    if capi_comm:
        if rehearse:
            def gloo_allgather(user, send, recv, nbytes, stream):
                torch.cuda.synchronize()
                h_in = torch.empty(nbytes, dtype=torch.uint8)
                h_out = torch.empty(n * nbytes, dtype=torch.uint8)
                hip = _hip_runtime()
                if hip.hipMemcpy(h_in.data_ptr(), send, nbytes, 2) != 0:
                    return 1
                dist.all_gather_into_tensor(h_out, h_in)
                return hip.hipMemcpy(recv, h_out.data_ptr(), n * nbytes, 1)
            ctx.attachCommCustom(gloo_allgather, rank, n)
        else:
            dist.barrier()                                                       # the communicator exists after the first collective
            attached = 1
            try:
                backend = dist.distributed_c10d._get_default_group()._get_backend(dev)
                ctx.attachComm(int(backend._comm_ptr()), rank, n)
            except Exception as e:                                               # no raw communicator in this torch build, or RCCL refused it
                attached = 0
                print(f"[bench rank {rank}] vkr_comm_attach unavailable ({e!r})", file=sys.stderr, flush=True)
            flag = torch.tensor([attached], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)                          # every rank takes the same path
            if int(flag.item()) == 0:
                if attached:
                    ctx.attachComm(0, 0, 0)                                      # detach
                capi_comm = False                                                # the torch.distributed gathers, like VKR_BENCH_TORCH_COMM=1

    pending = []   # the [rtao_full | sun] gather of the previous frame, still in flight on the communicator's stream

    def finish_output_gather():
        if capi_comm:
            torch.cuda.current_stream().wait_event(ev_out)
        if pending:
            work, planes = pending.pop()
            work.wait()
            bands.unpack_frame(gathered, planes, H, n)

    # The sun-shadow kernel shares nothing with the RTAO kernels (vkr_rt.h), so it is issued on a second stream right after
    # K1: its persistent waves fill the CUs as K1's waves drain, instead of the two launches each paying its own tail.
    side = torch.cuda.Stream(device=dev) if (shadows is not None and os.environ.get("VKR_BENCH_SUN_STREAM", "1") == "1") else None
    ev_sun, ev_sun_read = torch.cuda.Event(), torch.cuda.Event()
    comm = torch.cuda.Stream(device=dev) if capi_comm else None
    ev_k1, ev_hist, ev_dn, ev_out = (torch.cuda.Event() for _ in range(4))
    if capi_comm:
        ev_out.record(torch.cuda.current_stream())                             # nothing to wait for before the first frame

    def issue_sun(i, rows):
        shadows.gb = gbs[i]
        if side is None:
            shadows.sun(ubos[i], rows=rows)
            return
        side.wait_event(ev_sun_read)          # the previous frame's reader of sun_vis (recorded on the main stream) is done
        with torch.cuda.stream(side):
            shadows.sun(ubos[i], rows=rows)
            ev_sun.record(side)

    def join_sun():
        if side is not None:
            torch.cuda.current_stream().wait_event(ev_sun)

    def step(i):
        rpass.gbuffer = gbs[i]
        if n == 1:
            rpass.execute(PassExecuteContext(ubo=ubos[i]))
            if shadows is not None:
                issue_sun(i, None)                                             # queued behind K1: K1 holds the CUs until it drains
                join_sun()
                ev_sun_read.record(torch.cuda.current_stream())
            return
        if capi_comm:
            main = torch.cuda.current_stream()
            rpass.traceStripes(ubos[i], rank, n)                               # overlaps the previous frame's output gather
            ev_k1.record(main)
            if shadows is not None:
                issue_sun(i, (y0, y1))                                         # side stream: fills K1's drain, then overlaps the history gather
            comm.wait_event(ev_k1)
            with torch.cuda.stream(comm):
                rpass.gatherHistory()                                          # pack + ncclAllGather + unpack on the communication stream
                ev_hist.record(comm)
            main.wait_event(ev_out)                                            # frame i-1 is assembled before its planes are rewritten
            planes = [rpass.rtao_full]
            if shadows is not None:
                join_sun()
                sun16[y0:y1].copy_(shadows.sun_vis[y0:y1])
                ev_sun_read.record(main)
                planes.append(sun16.view(torch.int16))
            main.wait_event(ev_hist)                                           # A-Trous reads its halo rows from the gathered history
            rpass.denoiseBand(ubos[i], y0, y1)
            ev_dn.record(main)
            comm.wait_event(ev_dn)
            with torch.cuda.stream(comm):
                ctx.gatherBands(planes)
                ev_out.record(comm)
            return
        rpass.traceStripes(ubos[i], rank, n)                                  # overlaps the previous frame's output gather
        if shadows is not None:
            issue_sun(i, (y0, y1))                                             # fills K1's drain, then overlaps the history gather
        hc = rpass.historyCurr()
        plan.pack(hc, rank, hist_send)
        work = bands.all_gather_bytes(hist_send, hist_recv, async_op=True)
        finish_output_gather()                                                 # frame i-1 is assembled before its planes are rewritten
        planes = [rpass.rtao_full]
        if shadows is not None:
            join_sun()
            sun16[y0:y1].copy_(shadows.sun_vis[y0:y1])
            ev_sun_read.record(torch.cuda.current_stream())
            planes.append(sun16.view(torch.int16))
        work.wait()
        plan.unpack(hist_recv, hc)
        rpass.denoiseBand(ubos[i], y0, y1)
        bands.pack_band(planes, H, n, rank, packed)
        pending.append((bands.all_gather_frame(packed, gathered, async_op=True), planes))

    for i in range(Wu):
        step(i)
        log(f"warmup step {i} issued")
    finish_output_gather()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(Wu, Wu + K):
        step(i)
    finish_output_gather()                 # the last frame is assembled inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed:.3f} s")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the sharded run's last frame, kept for the comparison with the unsharded replay below
    sharded_last = None
    if n > 1 and not args.no_extra:
        sharded_last = (rpass.rtao_full.clone(), rpass.image(0).clone(), sun16.clone() if shadows is not None else None)

    # ---- untimed: count the rays of the timed frames (AO incl. retries via counters; shadow = valid px * samples)
    trace_stats_enable(ctx, True)
    trace_stats(ctx, reset=True)
    rpass.band = None
    for i in range(Wu, Wu + K):
        rpass.gbuffer = gbs[i]
        rpass.dispatchTrace(ubos[i])
        rpass.endFrame()
    st = trace_stats(ctx, reset=True)
    trace_stats_enable(ctx, False)
    ao_rays = int(st.rays)
    shadow_rays = 0
    if sun_samples > 0:
        shadow_rays = sum(int(gbs[i].valid.sum().item()) for i in range(Wu, Wu + K)) * sun_samples
    total_rays = ao_rays + shadow_rays
    value = total_rays / elapsed / 1e6

    out = {
        "metric": "RTAO Mrays/s (AO + retry + shadow rays traced / wall time of trace+temporal+A-Trous[+sun] frames)",
        "value": value, "unit": "Mrays/s", "n_gpus": n, "steps": K, "warmup": Wu,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
        "scaling": "strong" if n > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{wl_name}: {scene.name} {W}x{H}, {rays} AO rays/px + temporal + {iters} A-Trous iters"
                               + (f" + {sun_samples} sun-disc shadow rays/px" if sun_samples else "")
                               + (f", sharded over {n} GPUs (K1: interleaved 8-row stripes, A-Trous/sun: row bands) + 2 RCCL all-gathers/frame" if n > 1 else ""),
                   "triangles": info.triangle_count, "instances": int(len(scene.instance_mesh)),
                   "bvh_levels": info.height, "bvh_nodes": info.node_count, "binary_height": info.binary_height, "lbvh_build_ms": build_ms,
                   "rays_per_frame": total_rays / K, "parallelism": f"rowband{n}" if n > 1 else "single",
                   "collective": ("none" if n == 1 else "c-abi (vkr_rtao_gather_history + vkr_gather_bands: fused pack, ncclAllGather, fused unpack) on a "
                                  "communication stream" if capi_comm else "torch.distributed all_gather"),
                   "resident_gbuffers": resident},
    }

    if rank == 0 and not args.no_extra and n > 1:
        # ---- untimed reference for the strong-scaling figure: the SAME frames, whole frame on this rank alone (the
        # default N=1 run measures workload c3, so value(N)/value(1) across runs would compare different workloads)
        rpass.band = None
        rpass.reset()

        def whole_frame(i):
            rpass.gbuffer = gbs[i]
            rpass.execute(PassExecuteContext(ubo=ubos[i]))
            if shadows is not None:
                shadows.gb = gbs[i]
                shadows.sun(ubos[i])

        for i in range(Wu):                                                # the same frames from the same cleared history
            whole_frame(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(Wu, Wu + K):
            whole_frame(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t1
        # the invariant of SURVEY 8e on the real transport: the frame the N ranks assembled == the unsharded frame, bit for bit
        same = {"rtao_full": bool(torch.equal(sharded_last[0], rpass.rtao_full)), "history": bool(torch.equal(sharded_last[1], rpass.image(0)))}
        if shadows is not None:
            same["sun"] = bool(torch.equal(sharded_last[2], shadows.sun_vis.to(torch.float16)))
        out["sharded_equals_unsharded"] = all(same.values())
        out["sharded_equals_unsharded_planes"] = same
        out["strong_scaling_reference"] = {
            "n_gpus": 1, "ms_per_step": t1 / K * 1e3, "value": total_rays / t1 / 1e6, "unit": "Mrays/s",
            "note": "same workload and frames, unsharded, on rank 0 alone after the timed region (other ranks idle); "
                    "speed-up = value / this value"}
        out["speedup_vs_one_gpu_same_workload"] = value / (total_rays / t1 / 1e6)

    if rank == 0 and not args.no_extra:
        # ---- per-kernel timing, HIP events on the launch stream, over the timed frames (single-GPU full frame)
        rpass.band = None
        rpass.reset()
        t_trace, t_atrous, t_up = [], [], []
        for i in range(Wu, Wu + K):
            rpass.gbuffer = gbs[i]
            t_trace.append(event_time_ms(lambda: rpass.dispatchTrace(ubos[i]), None))
            t_atrous.append(event_time_ms(lambda: rpass.dispatchAtrous(ubos[i]), None))
            t_up.append(event_time_ms(lambda: rpass.dispatchUpsample(ubos[i]), None))
            rpass.endFrame()
        trace_ms = float(np.mean(t_trace))
        atrous_ms = float(np.mean(t_atrous))
        px = W * H
        # ---- K1 roofline.  The kernel is bound by vector-instruction issue (and, next to it, by the L1 return path): see
        # DESIGN.md 4.  achieved = wave-level VALU instructions of one launch / launch time (HIP events, this run);
        # peak = 1024 SIMDs x 2.4 GHz / (calibrated issue cycles per instruction for K1's instruction mix).  The
        # instruction count and the mix are properties of the code + workload, measured once with rocprofv3 --pmc /
        # the ISA census and read from profiles/ (named in `source`); the issue costs come from tools/calib/issue_rates*.
        cal, model_fresh, model_name = {}, None, os.environ.get("VKR_BENCH_MODEL", "r3_k1_valu_model.json")
        try:
            with open(os.path.join(ROOT, "profiles", model_name)) as f:
                model = json.load(f)
            cal = model.get(wl_name, {})
            from vulkanrenderer_amd.fingerprint import kernel_source_fingerprint
            model_fresh = model.get("source_fingerprint") == kernel_source_fingerprint()
        except Exception:
            cal = {}
        # SURVEY 8(d)'s byte figure (8 B/px + 64 B per node visit + 48 B per triangle test): bytes served by L1/L2, NOT HBM
        cache_side = 8.0 * px + (st.nodes_visited * 64.0 + st.tris_tested * 48.0) / K
        compulsory = 8.0 * px + info.node_count * 64.0 + info.triangle_count * 48.0
        roof = {"kernel": "k_trace_q (K1 trace+temporal)", "launch_ms": trace_ms,
                "nodes_per_ray": st.nodes_visited / max(1, ao_rays), "tris_per_ray": st.tris_tested / max(1, ao_rays),
                "lane_util_node_loop": st.nodes_visited / max(1, 64 * st.wave_node_iters),
                "lane_util_tri_loop": st.tris_tested / max(1, 64 * st.wave_tri_iters),
                "cache_side_bytes_per_launch": cache_side, "cache_side_gbs": cache_side / (trace_ms * 1e-3) / 1e9,
                "compulsory_bytes_per_launch": compulsory,
                "hbm_frac_compulsory": compulsory / (trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if cal:
            # instructions of THIS launch = measured instructions per ray x the rays this launch traced (a rank of an N>1 run
            # traces its stripes only); peak from the lower-bound price of the mix, so frac <= the true VALU-issue utilisation
            rays_launch = ao_rays / K
            insts = float(cal["valu_insts_per_ray"]) * rays_launch
            cyc = float(cal["mean_issue_cycles_lower"])
            peak = 1024 * 2.4e9 / cyc / 1e9                      # G wave-instructions/s the chip can issue for this mix
            ach = insts / (trace_ms * 1e-3) / 1e9
            share = rays_launch / float(cal["rays_per_launch"])
            traffic = float(cal["hbm_bytes_per_launch"]) * share
            roof.update({"bound": "valu", "achieved": ach, "peak": peak, "unit": "G wave-inst/s", "frac": ach / peak,
                         "frac_estimate": ach / (1024 * 2.4e9 / float(cal["mean_issue_cycles_estimate"]) / 1e9),
                         "valu_insts_per_launch": insts, "issue_cycles_per_inst": {"lower": cyc, "estimate": cal["mean_issue_cycles_estimate"]},
                         "traffic": traffic, "traffic_frac_of_hbm_peak": traffic / (trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "td_busy_frac": cal.get("td_busy_frac"), "ta_busy_frac": cal.get("ta_busy_frac"),
                         "l1_hit_rate": cal.get("l1_hit_rate"), "l2_hit_rate": cal.get("l2_hit_rate"),
                         "source": f"profiles/{model_name} <- " + str(cal.get("source")),
                         "model_fresh": model_fresh,
                         "note": "achieved/peak/frac: VALU issue, co-binding with the L1 return path (td_busy_frac), DESIGN.md 4.  peak prices the "
                                 "instruction classes no counter resolves at the fast class, so frac is a lower bound of the issue utilisation; "
                                 "frac_estimate prices them by the static mix of the queue loop.  Instructions per ray, mix, traffic, TA/TD busy "
                                 "and hit rates were measured with rocprofv3 --pmc on this workload (file in `source`), not by this run; launch_ms "
                                 "and the ray count are this run's.  model_fresh: the kernel sources in the tree are the ones the counters were "
                                 "measured on (false = re-profile with tools/refresh_r3.sh; the fractions then describe an older kernel).  "
                                 "cache_side_bytes is SURVEY 8(d)'s figure: bytes served by L1/L2, not HBM."})
        else:
            roof.update({"bound": "hbm", "achieved": compulsory / (trace_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": compulsory / (trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "note": "no PMC model committed for this workload: compulsory bytes (images + the whole BVH once) against the HBM peak"})
        out["roofline"] = roof
        # ---- K2 roofline: HBM, 14 B/px per iteration
        atrous_bytes = 14.0 * px * iters
        k2 = {}
        if cal and cal.get("k2_valu_insts_per_iteration") and cal.get("k2_mean_issue_cycles_lower"):
            # the bound K2 actually sits on: VALU issue (two software exp + pow32 per tap are the contract).  Instructions per
            # iteration from the PMC model x this run's iteration count / this run's HIP-event time, against the issue rate the
            # chip sustains for K2's own instruction mix (lower-bound price: frac_valu is a lower bound of the utilisation)
            k2_insts = float(cal["k2_valu_insts_per_iteration"]) * iters * (px / float(cal.get("k2_pixels", px)))
            k2_ach = k2_insts / (atrous_ms * 1e-3) / 1e9
            k2_peak = 1024 * 2.4e9 / float(cal["k2_mean_issue_cycles_lower"]) / 1e9
            k2 = {"frac_valu": k2_ach / k2_peak, "valu_achieved_ginst_s": k2_ach, "valu_peak_ginst_s": k2_peak,
                  "frac_valu_estimate": k2_ach / (1024 * 2.4e9 / float(cal.get("k2_mean_issue_cycles_estimate", cal["k2_mean_issue_cycles_lower"])) / 1e9),
                  "valu_issue_cycles_per_inst": {"lower": cal["k2_mean_issue_cycles_lower"], "estimate": cal.get("k2_mean_issue_cycles_estimate")}}
        out["roofline_k2"] = {"bound": "hbm", "kernel": f"{iters} x k_atrous_tile (K2: operands decoded while a tile is staged, upsample folded into the last store)",
                              "achieved": atrous_bytes / (atrous_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": atrous_bytes / (atrous_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "traffic": (cal["k2_hbm_bytes_per_iteration"] * iters) if cal else None,
                              "launch_ms": atrous_ms, "algorithmic_bytes": atrous_bytes,
                              "valu_insts": (cal["k2_valu_insts_per_iteration"] * iters) if cal else None, **k2,
                              "source": (f"profiles/{model_name} <- " + str(cal.get("source"))) if cal else None,
                              "note": "two fractions: frac = SURVEY 8(d)'s algorithmic 14 B/px/iteration against the HBM peak (the roofline the survey "
                                      "assigns); frac_valu = the bound the kernel is on -- VALU issue: two software exp (bit-exact against the oracle) + "
                                      "pow32 per tap, ~57 instructions x 25 taps per pixel; the v_exp_f32 build (VKR_ATROUS=tol, outside the contract) is "
                                      "timed in DESIGN.md"}
        out["kernels"] = {"trace_ms": trace_ms, "atrous_total_ms": atrous_ms, "upsample_ms": float(np.mean(t_up)),
                          "ao_mrays_per_s_kernel": ao_rays / K / (trace_ms * 1e-3) / 1e6,
                          "atrous_gbs_algorithmic": atrous_bytes / (atrous_ms * 1e-3) / 1e9,
                          "first_build_incl_upload_s": first_build_s}
        # The reference's own default configuration (AppConfig.h:88-96: RTAO_RAY_COUNT 8, RTAO_ATROUS_ITERATIONS 5, steps 1..16) on the
        # same frames: BASELINE configs[3] names 4 iterations, which is what `value` is quoted on; this is the pass as the reference ships it
        if n == 1 and iters != 5:
            rp5 = RtaoComputePass(ctx, gbs[0], atrous_iterations=5)
            t5 = []
            for i in range(Wu, Wu + min(K, 10)):
                rp5.gbuffer = gbs[i]
                t5.append(event_time_ms(lambda: rp5.execute(PassExecuteContext(ubo=ubos[i])), None))
            t5a = [event_time_ms(lambda: rp5.dispatchAtrous(ubos[Wu]), None) for _ in range(3)]
            out["kernels"]["reference_default_8rays_5iters"] = {"frame_ms": float(np.mean(t5[1:] if len(t5) > 1 else t5)),
                                                                "atrous_total_ms": float(np.mean(t5a)), "atrous_steps": [1, 2, 4, 8, 16]}
            rpass.reset()
        # RayTracingContext::updateTopLevelAS on unchanged transforms: re-transform + refit of the resident BVH
        from vulkanrenderer_amd.renderer import instances_to_array
        inst_arr = instances_to_array(scene.instance_mesh, scene.instance_transform)
        out["kernels"]["bvh_refit_ms"] = min(event_time_ms(
            lambda: ctx.lib.vkr_bvh_refit(ctx.handle, inst_arr, len(scene.instance_mesh), ctx.stream()), None)
            for _ in range(3))
        if shadows is not None:
            ts = [event_time_ms(lambda: (setattr(shadows, "gb", gbs[i]), shadows.sun(ubos[i])), None) for i in range(Wu, Wu + min(K, 5))]
            out["kernels"]["sun_shadow_ms_full_frame"] = float(np.mean(ts))
            # K5, not part of the timed workload (the reference's default point-light count is 0, AppConfig.h:99): one
            # point light's shadow rays over the whole frame (pbr.frag:169-196, light 0 of FrameManager.cpp:477-483)
            tp = [event_time_ms(lambda: (setattr(shadows, "gb", gbs[i]), shadows.point(ubos[i], 0)), None) for i in range(Wu, Wu + min(K, 5))]
            out["kernels"]["point_shadow_ms_full_frame_per_light"] = float(np.mean(tp))

    if rank == 0 and n == 1 and args.cpu_seconds > 0:
        depth0 = gbs[Wu].linear_depth.cpu().numpy().view(np.uint16)
        normal0 = gbs[Wu].normal.cpu().numpy().view(np.uint16)
        # BASELINE configs[0] in full for the CPU leg: sponza-class 640x360, 4 rays, 2 iterations (G-buffer from the GPU producer)
        c0 = None
        if wl_name == "c3" and not os.environ.get("VKR_BENCH_NO_C0"):
            W0, H0 = WORKLOADS["c0"][0], WORKLOADS["c0"][1]
            sc0 = make_scene("sponza")
            ctx0 = Context(W0, H0, local_rank)
            rt0 = RayTracingContext(ctx0)
            rt0.init(sc0)
            fs0 = FrameState(W0, H0, ray_count=4)
            e0, t0_ = camera_path(sc0, 1, step=0.05)[0]
            ubo0 = fs0.update(e0, t0_)
            gb0 = GBuffer(ctx0)
            generate_gbuffer(ctx0, ubo0, gb0)
            ctx0.sync()
            c0 = (sc0, ubo0, gb0.linear_depth.cpu().numpy().view(np.uint16), gb0.normal.cpu().numpy().view(np.uint16))
        (ya, yb, want_full, want_hist), out["cpu_baseline"] = cpu_baseline(scene, wl, ubos[Wu], depth0, normal0, args.cpu_seconds, c0)
        # the oracle is the checker here too: the GPU's whole pass (K1 + A-Trous x N, upsample folded) for the same frame, from
        # the same cleared history, against the rows the CPU leg just computed (the parity tests proper are tests/ -m gpu)
        rpass.band = None
        rpass.reset()
        rpass.gbuffer = gbs[Wu]
        rpass.execute(PassExecuteContext(ubo=ubos[Wu]))
        ctx.sync()
        got_full = rpass.rtao_full.cpu().numpy().view(np.uint16).reshape(H, W)[ya:yb]
        got_hist = rpass.image(0).cpu().numpy().view(np.uint16).reshape(H, W)[ya:yb]
        ref_full = np.asarray(want_full).view(np.uint16)
        ref_hist = np.asarray(want_hist).view(np.uint16)
        diff = np.abs(got_full.view(np.float16).astype(np.float32) - ref_full.view(np.float16).astype(np.float32))
        out["parity"] = {"checked": f"rtao_full (K1 trace+temporal + {iters} A-Trous iterations) and the K1 history, rows [{ya},{yb}) of the first timed "
                                    "frame, GPU vs CPU oracle (R16F bits)",
                         "pixels": int(got_full.size), "mismatching_pixels": int((got_full != ref_full).sum()),
                         "mismatching_history_pixels": int((got_hist != ref_hist).sum()),
                         "pixels_over_1_255": int((diff > 1.0 / 255.0).sum()), "max_abs_diff": float(diff.max())}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
