"""GPU tests of the collective behind the C boundary (comm.hip): vkr_comm_attach / vkr_comm_attach_custom,
vkr_rtao_gather_history, vkr_gather_bands.  No reference counterpart (the reference is single-GPU, Renderer.cpp:359);
the property tested is 'sharded == unsharded, bit for bit'.

N ranks are emulated with N contexts on ONE GPU: the custom all-gather callback keeps one device slot per rank.  A real
all-gather is a rendezvous and the emulation is sequential, so every gather is issued in two rounds: round one deposits
every rank's slot (ranks see incomplete data, but their own rows come back unchanged), round two hands every rank the
complete set.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import to_np_u16

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _ctx(scene, W, H):
    from vulkanrenderer_amd.renderer import Context, GBuffer, RayTracingContext
    ctx = Context(W, H)
    rt = RayTracingContext(ctx)
    rt.init(scene)
    return ctx, rt, GBuffer(ctx)


class SlotExchange:
    """One slot per rank in device memory; the callback of rank r stores its send block and reads all slots."""

    def __init__(self, torch, n, device):
        self.torch, self.n, self.device = torch, n, device
        self.slots = None
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.calls = []

    def callback(self, rank):
        def fn(user, send, recv, nbytes, stream):
            if self.slots is None or self.slots.numel() != self.n * nbytes:
                self.slots = self.torch.zeros(self.n * nbytes, dtype=self.torch.uint8, device=self.device)
            base = self.slots.data_ptr()
            self.calls.append((rank, nbytes))
            if self.hip.hipMemcpyAsync(base + rank * nbytes, send, nbytes, 3, stream) != 0:
                return 1
            return self.hip.hipMemcpyAsync(recv, base, self.n * nbytes, 3, stream)
        return fn


def _comm_vs_full(torch, scene, W, H, rays, iters, n, frames, step=0.3):
    from vulkanrenderer_amd import FrameState, bands, camera_path
    from vulkanrenderer_amd.renderer import PassExecuteContext, RtaoComputePass, generate_gbuffer
    full_ctx, _, full_gb = _ctx(scene, W, H)
    full_pass = RtaoComputePass(full_ctx, full_gb, atrous_iterations=iters)
    xch = SlotExchange(torch, n, full_ctx.device)
    ranks = []
    for r in range(n):
        ctx, rt, gb = _ctx(scene, W, H)
        ctx.attachCommCustom(xch.callback(r), r, n)
        ranks.append((ctx, rt, gb, RtaoComputePass(ctx, gb, atrous_iterations=iters)))
    extra = [torch.zeros((H, W), dtype=torch.int16, device=full_ctx.device) for _ in range(n)]   # a second plane
    fs = FrameState(W, H, ray_count=rays)
    for f, (eye, tgt) in enumerate(camera_path(scene, frames, step=step)):
        ubo = fs.update(eye, tgt)
        generate_gbuffer(full_ctx, ubo, full_gb)
        full_pass.execute(PassExecuteContext(ubo=ubo))
        full_ctx.sync()
        for r, (ctx, rt, gb, p) in enumerate(ranks):
            gb.linear_depth.copy_(full_gb.linear_depth)
            gb.normal.copy_(full_gb.normal)
            p.historyCurr().fill_(0x1234)                   # rows this rank does not trace must come from the gather
            p.traceStripes(ubo, r, n)
        for _round in range(2):
            for r, (ctx, rt, gb, p) in enumerate(ranks):
                p.gatherHistory()
        for r, (ctx, rt, gb, p) in enumerate(ranks):
            y0, y1 = bands.band_rows(H, n, r)
            p.denoiseBand(ubo, y0, y1)
            extra[r].fill_(-1)
            extra[r][y0:y1] = full_pass.rtao_full[y0:y1] ^ 0x55
        for _round in range(2):
            for r, (ctx, rt, gb, p) in enumerate(ranks):
                ctx.gatherBands([p.rtao_full, extra[r]])
        torch.cuda.synchronize()
        for r, (ctx, rt, gb, p) in enumerate(ranks):
            assert np.array_equal(to_np_u16(p.image(0)), to_np_u16(full_pass.image(0))), f"frame {f} rank {r} history"
            assert np.array_equal(to_np_u16(p.rtao_full), to_np_u16(full_pass.rtao_full)), f"frame {f} rank {r} rtao_full"
            assert torch.equal(extra[r], full_pass.rtao_full ^ 0x55), f"frame {f} rank {r} second plane"
    assert len(xch.calls) == frames * n * 4
    for ctx, *_ in ranks:
        ctx.close()
    full_ctx.close()


@pytest.mark.parametrize("n,W,H,iters", [(2, 160, 90, 2), (3, 161, 91, 3), (8, 160, 37, 5), (5, 33, 8, 2), (1, 160, 90, 4)])
def test_capi_gathers_equal_full_frame(torch_mod, n, W, H, iters):
    """Ragged stripes and bands, odd widths (the 2-byte path of the stripe kernels), more ranks than stripes."""
    from vulkanrenderer_amd import make_street_scene
    _comm_vs_full(torch_mod, make_street_scene(21, 0.02, "band"), W, H, 4, iters, n, 3)


def test_capi_gathers_equal_full_frame_1080p_8_ranks(torch_mod):
    """BASELINE.json full size over 8 emulated ranks through the C-ABI gathers."""
    from vulkanrenderer_amd import make_sponza_class
    _comm_vs_full(torch_mod, make_sponza_class(), 1920, 1080, 8, 4, 8, 2, step=0.05)


def test_comm_errors(torch_mod):
    from vulkanrenderer_amd import _capi, make_street_scene
    from vulkanrenderer_amd.renderer import RtaoComputePass
    ctx, rt, gb = _ctx(make_street_scene(5, 0.02, "e"), 64, 32)
    p = RtaoComputePass(ctx, gb, atrous_iterations=1)
    lib = ctx.lib
    # nothing attached
    assert lib.vkr_rtao_gather_history(ctx.handle, None) == _capi.VKR_ERR_STATE
    assert b"no communicator" in lib.vkr_last_error(ctx.handle)
    arr = (C.c_void_p * 1)(p.rtao_full.data_ptr())
    assert lib.vkr_gather_bands(ctx.handle, arr, 1, None) == _capi.VKR_ERR_STATE
    # bad ranks / plane counts
    fn = _capi.ALLGATHER_FN(lambda *a: 7)
    assert lib.vkr_comm_attach_custom(ctx.handle, fn, None, 2, 2) == _capi.VKR_ERR_INVALID_ARG
    assert lib.vkr_comm_attach_custom(ctx.handle, fn, None, 0, 0) == _capi.VKR_ERR_INVALID_ARG
    assert lib.vkr_comm_attach_custom(ctx.handle, fn, None, 0, 1) == _capi.VKR_OK
    assert lib.vkr_gather_bands(ctx.handle, arr, 0, None) == _capi.VKR_ERR_INVALID_ARG
    assert lib.vkr_gather_bands(ctx.handle, arr, 5, None) == _capi.VKR_ERR_INVALID_ARG
    null = (C.c_void_p * 1)(None)
    assert lib.vkr_gather_bands(ctx.handle, null, 1, None) == _capi.VKR_ERR_INVALID_ARG
    # a failing transport surfaces as VKR_ERR_RCCL with the callback's code in the message
    assert lib.vkr_rtao_gather_history(ctx.handle, None) == _capi.VKR_ERR_RCCL
    assert b"returned 7" in lib.vkr_last_error(ctx.handle)
    # detach
    assert lib.vkr_comm_attach_custom(ctx.handle, _capi.ALLGATHER_FN(0), None, 0, 1) == _capi.VKR_OK
    assert lib.vkr_rtao_gather_history(ctx.handle, None) == _capi.VKR_ERR_STATE
    torch_mod.cuda.synchronize()
    ctx.close()


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_rccl_world_size_1_communicator(torch_mod):
    """A real RCCL communicator (one rank: the GPU box has one GPU) attached with vkr_comm_attach: the gathers run
    pack -> ncclAllGather -> unpack on the caller's stream and reproduce the unsharded frame; a communicator whose
    rank / size differ from the arguments is rejected."""
    torch = torch_mod
    from vulkanrenderer_amd import FrameState, _capi, camera_path, make_street_scene
    from vulkanrenderer_amd.renderer import PassExecuteContext, RtaoComputePass, generate_gbuffer
    rccl = C.CDLL("librccl.so.1")               # by soname: the copy this process already holds (torch's)
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    torch.cuda.set_device(0)
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    scene = make_street_scene(21, 0.02, "band")
    W, H = 160, 90
    full_ctx, _, full_gb = _ctx(scene, W, H)
    full_pass = RtaoComputePass(full_ctx, full_gb, atrous_iterations=3)
    ctx, _, gb = _ctx(scene, W, H)
    p = RtaoComputePass(ctx, gb, atrous_iterations=3)
    assert ctx.lib.vkr_comm_attach(ctx.handle, comm, 1, 2) == _capi.VKR_ERR_INVALID_ARG
    assert b"rank 0 of 1" in ctx.lib.vkr_last_error(ctx.handle)
    ctx.attachComm(comm.value, 0, 1)
    fs = FrameState(W, H, ray_count=4)
    for f, (eye, tgt) in enumerate(camera_path(scene, 3, step=0.3)):
        ubo = fs.update(eye, tgt)
        generate_gbuffer(full_ctx, ubo, full_gb)
        full_pass.execute(PassExecuteContext(ubo=ubo))
        gb.linear_depth.copy_(full_gb.linear_depth)
        gb.normal.copy_(full_gb.normal)
        p.traceStripes(ubo, 0, 1)
        p.gatherHistory()
        p.denoiseBand(ubo, 0, H)
        ctx.gatherBands([p.rtao_full])
        torch.cuda.synchronize()
        assert np.array_equal(to_np_u16(p.image(0)), to_np_u16(full_pass.image(0))), f"frame {f} history"
        assert np.array_equal(to_np_u16(p.rtao_full), to_np_u16(full_pass.rtao_full)), f"frame {f} rtao_full"
    ctx.attachComm(0, 0, 0)                       # detach before the communicator goes away
    ctx.close()
    full_ctx.close()
    assert rccl.ncclCommDestroy(comm) == 0
