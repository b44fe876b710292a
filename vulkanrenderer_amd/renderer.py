"""Host-side mirror of the reference's pass interface for the RTAO / shadow-ray path, above the C ABI.

Names and call shapes follow the reference so tests read like its own code would:
  RenderPass / PassExecuteContext / RenderStats  app/include/Rendering/core/RenderPass.h:12-67
  Rendergraph (AddPass / Compile / Execute)      app/src/Rendering/core/Rendergraph.cpp:150-204,252-348
  RayTracingContext (init / updateTopLevelAS)    app/src/Rendering/RHI/Vulkan/RayTracingContext.cpp:124-197
  RtaoComputePass (inputs {"depth"} -> outputs {"rtao_full"})   app/src/Rendering/pass/RtaoComputePass.cpp:16-25,102-156
Errors surface as RuntimeError, as the reference throws std::runtime_error.

torch is used only for device memory and streams; all compute goes through libvkr_rt.so.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _capi
from .scenes import SceneData, VERTEX_STRIDE, UV_OFFSET_FLOATS, NORMAL_OFFSET_FLOATS, MAX_MATERIALS


def _check(ctx_handle, rc: int, what: str):
    if rc != _capi.VKR_OK:
        msg = _capi.lib().vkr_last_error(ctx_handle)
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def _stream_ptr(device: torch.device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else int(t.data_ptr())


def scene_to_desc(scene: SceneData):
    """SceneData -> vkr_scene_desc (+ the list of arrays that must stay alive during the call)."""
    keep = []
    meshes = (_capi.MeshDesc * max(1, len(scene.meshes)))()
    opaque = scene.mesh_opaque_flags()
    for i, m in enumerate(scene.meshes):
        v = np.ascontiguousarray(m.vertices, dtype=np.float32)
        idx = np.ascontiguousarray(m.indices, dtype=np.uint32)
        keep += [v, idx]
        meshes[i].positions = v.ctypes.data
        meshes[i].position_stride = VERTEX_STRIDE
        meshes[i].vertex_count = v.shape[0]
        meshes[i].indices = idx.ctypes.data
        meshes[i].index_count = idx.shape[0]
        meshes[i].uvs = v.ctypes.data + 4 * UV_OFFSET_FLOATS
        meshes[i].uv_stride = VERTEX_STRIDE
        meshes[i].material_index = int(m.material_index)
        meshes[i].opaque = int(opaque[i])
        meshes[i].normals = v.ctypes.data + 4 * NORMAL_OFFSET_FLOATS
        meshes[i].normal_stride = VERTEX_STRIDE
    n_inst = len(scene.instance_mesh)
    inst = (_capi.Instance * max(1, n_inst))()
    for i in range(n_inst):
        inst[i].mesh_index = int(scene.instance_mesh[i])
        inst[i].transform[:] = [float(x) for x in scene.instance_transform[i]]
    mp = np.ascontiguousarray(scene.material_params(), dtype=np.float32)
    tex = (_capi.AlphaTexture * MAX_MATERIALS)()
    for i, mat in enumerate(scene.materials[:MAX_MATERIALS]):
        if mat.alpha_texture is not None:
            a = np.ascontiguousarray(mat.alpha_texture, dtype=np.uint8)
            keep.append(a)
            tex[i].alpha = a.ctypes.data
            tex[i].width = a.shape[1]
            tex[i].height = a.shape[0]
    keep += [meshes, inst, mp, tex]
    desc = _capi.SceneDesc()
    desc.meshes = meshes
    desc.mesh_count = len(scene.meshes)
    desc.instances = inst
    desc.instance_count = n_inst
    desc.material_params = mp.ctypes.data
    desc.alpha_textures = tex
    return desc, keep


def instances_to_array(instance_mesh: np.ndarray, instance_transform: np.ndarray):
    n = len(instance_mesh)
    inst = (_capi.Instance * max(1, n))()
    for i in range(n):
        inst[i].mesh_index = int(instance_mesh[i])
        inst[i].transform[:] = [float(x) for x in instance_transform[i]]
    return inst


class Context:
    """One vkr_ctx = the RTAO slice of FrameManager + RtaoComputePipeline + RayTracingContext on one GPU."""

    def __init__(self, width: int, height: int, device: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device: the RTAO path has no CPU fallback")
        self.lib = _capi.lib()
        self.width, self.height = int(width), int(height)
        self.device = torch.device("cuda", device)
        desc = _capi.CreateDesc(device, width, height)
        handle = C.c_void_p()
        torch.cuda.set_device(self.device)
        _check(None, self.lib.vkr_create(C.byref(desc), C.byref(handle)), "vkr_create")
        self.handle = handle

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vkr_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self) -> int:
        return _stream_ptr(self.device)

    def sync(self):
        _check(self.handle, self.lib.vkr_sync(self.handle, self.stream()), "vkr_sync")

    # the collective of the row-band schedule behind the C boundary (comm.hip; include/vkr_rt.h "multi-GPU")
    def attachComm(self, nccl_comm: int, rank: int, n_ranks: int):
        """Attach an RCCL communicator (an ncclComm_t as an integer address); 0 detaches."""
        _check(self.handle, self.lib.vkr_comm_attach(self.handle, C.c_void_p(nccl_comm or None), rank, n_ranks), "vkr_comm_attach")

    def attachCommCustom(self, fn, rank: int, n_ranks: int, user: int = 0):
        """Attach the host's own all-gather: fn(user, send, recv, bytes_per_rank, stream) -> 0 on success."""
        self._comm_fn = _capi.ALLGATHER_FN(fn) if fn is not None else _capi.ALLGATHER_FN(0)   # keep the thunk alive
        _check(self.handle, self.lib.vkr_comm_attach_custom(self.handle, self._comm_fn, C.c_void_p(user or None), rank, n_ranks),
               "vkr_comm_attach_custom")

    def gatherBands(self, planes: Sequence[torch.Tensor]):
        """All-gather the contiguous row bands of up to 4 full-frame R16F planes in place (pack + all-gather + unpack)."""
        arr = (C.c_void_p * len(planes))(*[p.data_ptr() for p in planes])
        _check(self.handle, self.lib.vkr_gather_bands(self.handle, arr, len(planes), self.stream()), "vkr_gather_bands")


class RayTracingContext:
    """RayTracingContext.cpp:124-197: init() = BLAS+TLAS build, updateTopLevelAS() = refit."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.instance_mesh = None

    def init(self, scene: SceneData):
        c = self.ctx
        desc, keep = scene_to_desc(scene)
        _check(c.handle, c.lib.vkr_scene_upload(c.handle, C.byref(desc)), "vkr_scene_upload")
        del keep
        _check(c.handle, c.lib.vkr_bvh_build(c.handle, c.stream()), "vkr_bvh_build")
        self.instance_mesh = np.array(scene.instance_mesh, dtype=np.uint32)

    def rebuild(self):
        c = self.ctx
        _check(c.handle, c.lib.vkr_bvh_build(c.handle, c.stream()), "vkr_bvh_build")

    def updateTopLevelAS(self, instance_mesh: np.ndarray, instance_transform: np.ndarray):
        c = self.ctx
        inst = instances_to_array(instance_mesh, instance_transform)
        _check(c.handle, c.lib.vkr_bvh_refit(c.handle, inst, len(instance_mesh), c.stream()), "vkr_bvh_refit")

    def setHierarchy(self, parents, locals_, first_instance, instance_count):
        """Upload the model's node hierarchy (DFS pre-order) once: Renderer::rebuildRayTracingInstances
        (Renderer.cpp:503-527) then runs on the device."""
        c = self.ctx
        n = len(parents)
        nodes = (_capi.SceneNode * max(1, n))()
        for i in range(n):
            nodes[i].parent = int(parents[i])
            nodes[i].local[:] = [float(x) for x in np.asarray(locals_[i], dtype=np.float32).reshape(16)]
            nodes[i].first_instance = int(first_instance[i])
            nodes[i].instance_count = int(instance_count[i])
        _check(c.handle, c.lib.vkr_hierarchy_upload(c.handle, nodes, n), "vkr_hierarchy_upload")

    def updateFromModelMatrix(self, model_matrix) -> bool:
        """The TLAS cache of Renderer::recordCommandBuffer (Renderer.cpp:470-479): refit only when the model matrix
        changed.  Returns whether a refit was enqueued."""
        c = self.ctx
        m = (C.c_float * 16)(*[float(x) for x in np.asarray(model_matrix, dtype=np.float32).reshape(16)])
        did = C.c_int(0)
        _check(c.handle, c.lib.vkr_bvh_refit_model(c.handle, m, C.byref(did), c.stream()), "vkr_bvh_refit_model")
        return bool(did.value)

    def instanceTransforms(self) -> np.ndarray:
        c = self.ctx
        out = np.zeros((max(1, len(self.instance_mesh)), 12), dtype=np.float32)
        _check(c.handle, c.lib.vkr_instances_download(c.handle, out.ctypes.data), "vkr_instances_download")
        return out[:len(self.instance_mesh)]

    def info(self) -> _capi.BvhInfo:
        c = self.ctx
        out = _capi.BvhInfo()
        _check(c.handle, c.lib.vkr_bvh_info_get(c.handle, C.byref(out)), "vkr_bvh_info_get")
        return out

    def download(self):
        info = self.info()
        nodes = np.zeros((max(1, info.node_count), 16), dtype=np.float32)
        tris = np.zeros((max(1, info.triangle_count), 12), dtype=np.float32)
        c = self.ctx
        _check(c.handle, c.lib.vkr_bvh_download(c.handle, nodes.ctypes.data, tris.ctypes.data), "vkr_bvh_download")
        return nodes[:info.node_count], tris[:info.triangle_count]


SLOT_DTYPE = np.dtype([("node_linear_index", "<u4"), ("mesh_index", "<u4"), ("mat_index", "<u4"), ("double_sided", "<u4")])
MESH_INFO_DTYPE = np.dtype([("vertex_offset", "<u4"), ("first_index", "<u4"), ("index_count", "<u4")])
DRAW_CMD_DTYPE = np.dtype([("index_count", "<u4"), ("instance_count", "<u4"), ("first_index", "<u4"),
                           ("vertex_offset", "<i4"), ("first_instance", "<u4")])
SPAN_DTYPE = np.dtype([("double_sided", "<u4"), ("mat_index", "<u4"), ("first_command", "<u4"), ("draw_count", "<u4")])


class SharedOpaqueIndirect:
    """FrameManager::buildSharedDrawSlots + prepareSharedOpaqueIndirect (FrameManager.cpp:81-175) on the device:
    per frame, node world matrices -> drawData (mat4 per draw) + VkDrawIndexedIndirectCommand[] + bucket spans."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.max_draws = 1
        self.slot_count = 0
        self.draw_data = None        # torch float32 [max_draws, 16]  (drawDataMapped)
        self.commands = None         # torch uint8   [max_draws, 20]  (indirectMapped)

    def buildSharedDrawSlots(self, node_parent, node_local, slots, mesh_infos, max_draws: int, mesh_bounds=None):
        c = self.ctx
        par = np.ascontiguousarray(node_parent, dtype=np.int32)
        loc = np.ascontiguousarray(node_local, dtype=np.float32).reshape(-1, 16)
        sl = np.ascontiguousarray(slots, dtype=SLOT_DTYPE)
        mi = np.ascontiguousarray(mesh_infos, dtype=MESH_INFO_DTYPE)
        mb = None if mesh_bounds is None else np.ascontiguousarray(mesh_bounds, dtype=np.float32).reshape(-1, 6)
        if mb is not None and len(mb) != len(mi):
            raise ValueError("mesh_bounds needs one AABB per mesh")
        d = _capi.IndirectDesc(par.ctypes.data if len(par) else None, loc.ctypes.data if len(par) else None, len(par),
                               sl.ctypes.data if len(sl) else None, len(sl), mi.ctypes.data if len(mi) else None, len(mi),
                               mb.ctypes.data if mb is not None and len(mb) else None, int(max_draws))
        _check(c.handle, c.lib.vkr_indirect_upload(c.handle, C.byref(d)), "vkr_indirect_upload")
        self.max_draws = max(1, int(max_draws))
        self.slot_count = len(sl)
        self.draw_data = torch.zeros((self.max_draws, 16), dtype=torch.float32, device=c.device)
        self.commands = torch.zeros((self.max_draws, 20), dtype=torch.uint8, device=c.device)

    def sortedSlots(self) -> np.ndarray:
        c = self.ctx
        out = np.zeros(max(1, self.slot_count), dtype=SLOT_DTYPE)
        _check(c.handle, c.lib.vkr_indirect_slots_download(c.handle, out.ctypes.data), "vkr_indirect_slots_download")
        return out[:self.slot_count]

    def prepareSharedOpaqueIndirect(self, model_matrix, view_proj=None):
        """Enqueue one frame's build on the context's stream (asynchronous)."""
        c = self.ctx
        m = (C.c_float * 16)(*[float(x) for x in np.asarray(model_matrix, dtype=np.float32).reshape(16)])
        vp = None
        if view_proj is not None:
            vp = (C.c_float * 16)(*[float(x) for x in np.asarray(view_proj, dtype=np.float32).reshape(16)])
        _check(c.handle, c.lib.vkr_indirect_prepare(c.handle, m, C.cast(vp, C.c_void_p) if vp is not None else None,
                                                    self.draw_data.data_ptr(), self.commands.data_ptr(),
                                                    min(self.draw_data.numel() // 16, self.commands.numel() // 20), c.stream()),
               "vkr_indirect_prepare")

    def result(self):
        """(sharedOpaqueDrawCount, sharedOpaqueBucketSpans); synchronises."""
        c = self.ctx
        cap = max(1, self.slot_count)
        spans = np.zeros(cap, dtype=SPAN_DTYPE)
        dc, sc = C.c_uint32(0), C.c_uint32(0)
        _check(c.handle, c.lib.vkr_indirect_result(c.handle, C.byref(dc), spans.ctypes.data, cap, C.byref(sc), c.stream()),
               "vkr_indirect_result")
        return int(dc.value), spans[:sc.value]

    def download(self):
        n, spans = self.result()
        dd = self.draw_data[:n].cpu().numpy()
        cm = self.commands[:n].cpu().numpy().reshape(-1).view(DRAW_CMD_DTYPE)
        return n, dd, cm, spans


class IblResult:
    """IblResult of IblPrecompute.h:7-22 as device tensors (views into packed buffers)."""

    def __init__(self, irradiance, prefilter, prefilter_size, brdf_lut):
        self.irradiance = irradiance            # int16 view of RGBA16F [6, S, S, 4]
        self.prefilter = prefilter              # int16 view of RGBA16F, mip-major packed
        self.prefilter_size = prefilter_size
        self.brdf_lut = brdf_lut                # int16 view of RG16F [S, S, 2]

    def prefilter_mip(self, mip: int) -> torch.Tensor:
        off = 0
        for m in range(mip):
            sm = max(1, self.prefilter_size >> m)
            off += 6 * sm * sm * 4
        sm = max(1, self.prefilter_size >> mip)
        return self.prefilter[off:off + 6 * sm * sm * 4].view(6, sm, sm, 4)


class EquirectToCubemap:
    """EquirectToCubemap::convert (EquirectToCubemap.cpp:66-398)."""

    @staticmethod
    def convert(ctx: Context, equirect: torch.Tensor, cube_size: int = 512) -> torch.Tensor:
        if equirect.dtype != torch.float32 or equirect.dim() != 3 or equirect.shape[2] != 4 or not equirect.is_cuda:
            raise ValueError("equirect must be a CUDA float32 [H, W, 4] tensor (RGBA32F)")
        eq = equirect.contiguous()
        cube = torch.empty((6, cube_size, cube_size, 4), dtype=torch.float32, device=ctx.device)
        _check(ctx.handle, ctx.lib.vkr_ibl_equirect_to_cube(ctx.handle, eq.data_ptr(), eq.shape[1], eq.shape[0],
                                                            cube.data_ptr(), cube_size, ctx.stream()), "vkr_ibl_equirect_to_cube")
        return cube


class IblPrecompute:
    """IblPrecompute::compute (IblPrecompute.cpp:75-561): irradiance 32^2, prefilter 128^2 with all mips, BRDF LUT 512^2."""

    @staticmethod
    def compute(ctx: Context, env_cube: torch.Tensor, irradiance_size: int = 32, prefilter_size: int = 128,
                brdf_lut_size: int = 512) -> IblResult:
        if env_cube.dtype != torch.float32 or env_cube.dim() != 4 or env_cube.shape[0] != 6 or env_cube.shape[3] != 4 \
                or env_cube.shape[1] != env_cube.shape[2] or not env_cube.is_cuda:
            raise ValueError("env_cube must be a CUDA float32 [6, S, S, 4] tensor (RGBA32F)")
        cube = env_cube.contiguous()
        lib = ctx.lib
        irr = torch.empty((6, irradiance_size, irradiance_size, 4), dtype=torch.int16, device=ctx.device)
        pre = torch.empty(int(lib.vkr_ibl_prefilter_texels(prefilter_size)) * 4, dtype=torch.int16, device=ctx.device)
        lut = torch.empty((brdf_lut_size, brdf_lut_size, 2), dtype=torch.int16, device=ctx.device)
        _check(ctx.handle, lib.vkr_ibl_compute(ctx.handle, cube.data_ptr(), cube.shape[1], irr.data_ptr(), irradiance_size,
                                               pre.data_ptr(), prefilter_size, lut.data_ptr(), brdf_lut_size, ctx.stream()),
               "vkr_ibl_compute")
        return IblResult(irr, pre, prefilter_size, lut)


@dataclass
class RenderStats:            # RenderPass.h:12-36 (only the field this path fills)
    rtaoMs: float = 0.0


@dataclass
class PassExecuteContext:     # RenderPass.h:38-44; the command buffer becomes a HIP stream
    stream: int = 0
    imageIndex: int = 0
    ubo: Optional[np.ndarray] = None
    stats: Optional[RenderStats] = None


class RenderPass:             # RenderPass.h:46-67, RenderPass.cpp:11-16
    def __init__(self, name: str, inputs: List[str], outputs: List[str]):
        self.name, self.inputs, self.outputs = name, list(inputs), list(outputs)

    def execute(self, ctx: PassExecuteContext):
        self.beginPass(ctx)
        self.render(ctx)
        self.endPass(ctx)

    def beginPass(self, ctx): ...
    def render(self, ctx): raise NotImplementedError
    def endPass(self, ctx): ...


class GBuffer:
    """What DepthPrepass leaves for the path: linearDepthResolve (R16F), normalResolve (RGBA16F), plus the
    worldPos / Ng varyings the shadow functions consume."""

    def __init__(self, ctx: Context, with_shading_inputs: bool = True):
        dev, W, H = ctx.device, ctx.width, ctx.height
        self.linear_depth = torch.zeros((H, W), dtype=torch.int16, device=dev)
        self.normal = torch.zeros((H, W, 4), dtype=torch.int16, device=dev)
        self.world_pos = torch.zeros((H, W, 3), dtype=torch.float32, device=dev) if with_shading_inputs else None
        self.geom_normal = torch.zeros((H, W, 3), dtype=torch.float32, device=dev) if with_shading_inputs else None
        self.valid = torch.zeros((H, W), dtype=torch.uint8, device=dev) if with_shading_inputs else None

    def upload(self, depth_u16: np.ndarray, normal_u16: np.ndarray, world_pos=None, geom_normal=None, valid=None):
        self.linear_depth.copy_(torch.from_numpy(depth_u16.view(np.int16)))
        self.normal.copy_(torch.from_numpy(normal_u16.view(np.int16)))
        if world_pos is not None:
            self.world_pos.copy_(torch.from_numpy(world_pos))
            self.geom_normal.copy_(torch.from_numpy(geom_normal))
            self.valid.copy_(torch.from_numpy(valid))


def generate_gbuffer(ctx: Context, ubo: np.ndarray, gb: GBuffer, samples: int = 1):
    """vkr_gbuffer_primary[_msaa]: the DepthPrepass stand-in; samples = the prepass's MSAA count."""
    u = (C.c_float * _capi.UBO_FLOATS)(*[float(x) for x in ubo])
    if samples == 1:
        _check(ctx.handle, ctx.lib.vkr_gbuffer_primary(ctx.handle, u, _ptr(gb.linear_depth), _ptr(gb.normal),
                                                       _ptr(gb.world_pos), _ptr(gb.geom_normal), _ptr(gb.valid),
                                                       ctx.stream()), "vkr_gbuffer_primary")
    else:
        _check(ctx.handle, ctx.lib.vkr_gbuffer_primary_msaa(ctx.handle, u, int(samples), _ptr(gb.linear_depth), _ptr(gb.normal),
                                                            _ptr(gb.world_pos), _ptr(gb.geom_normal), _ptr(gb.valid),
                                                            ctx.stream()), "vkr_gbuffer_primary_msaa")


class RtaoComputePass(RenderPass):
    """RtaoComputePass.cpp:16-25 (inputs {"depth"}, outputs {"rtao_full"}) and :102-156 (render)."""

    def __init__(self, ctx: Context, gbuffer: GBuffer, atrous_iterations: int = 5, denoise: bool = True,
                 want_hit_mask: bool = False):
        super().__init__("RtaoComputePass", ["depth"], ["rtao_full"])
        self.ctx, self.gbuffer = ctx, gbuffer
        # max(1, ENABLE_RTAO_SPATIAL_DENOISE ? RTAO_ATROUS_ITERATIONS : 1)  RtaoComputePass.cpp:529
        self.iterations = max(1, atrous_iterations if denoise else 1)
        self.rtao_full = torch.zeros((ctx.height, ctx.width), dtype=torch.int16, device=ctx.device)
        self.hit_mask = (torch.zeros((ctx.height, ctx.width), dtype=torch.int32, device=ctx.device)
                         if want_hit_mask else None)
        self.band = None      # (y0, y1) for row-band sharding, None = full frame
        # The synchronisation half of the pass (RtaoComputePass::beginPass / endPass record the image barriers around the three
        # dispatches, RtaoComputePass.cpp:27-100,158-255).  On HIP what is left of them is ordering against the producer of the
        # G-buffer and the consumer of rtao_full when those run on other streams / queues: two optional callables
        # (pass_ctx, stream), run where the reference records its barriers -- same contract as include/vkr_pass.hpp.
        self.before_pass = None
        self.after_pass = None

    def beginPass(self, pctx: PassExecuteContext):
        if self.before_pass is not None:
            self.before_pass(pctx, pctx.stream or self.ctx.stream())

    def endPass(self, pctx: PassExecuteContext):
        if self.after_pass is not None:
            self.after_pass(pctx, pctx.stream or self.ctx.stream())

    def _io(self, ubo: np.ndarray) -> _capi.RtaoFrameIO:
        io = _capi.RtaoFrameIO()
        io.ubo[:] = [float(x) for x in ubo]
        io.atrous_iterations = self.iterations
        io.linear_depth_r16f = _ptr(self.gbuffer.linear_depth)
        io.normal_rgba16f = _ptr(self.gbuffer.normal)
        io.rtao_full_r16f = _ptr(self.rtao_full)
        io.ray_hit_mask = _ptr(self.hit_mask)
        return io

    def render(self, pctx: PassExecuteContext):
        c = self.ctx
        io = self._io(pctx.ubo)
        stream = pctx.stream or c.stream()
        if self.band is None:
            _check(c.handle, c.lib.vkr_rtao_execute(c.handle, C.byref(io), stream), "vkr_rtao_execute")
        else:
            _check(c.handle, c.lib.vkr_rtao_execute_band(c.handle, C.byref(io), self.band[0], self.band[1], stream),
                   "vkr_rtao_execute_band")

    # the three dispatches individually (RtaoComputePass.cpp:503-571)
    def dispatchTrace(self, ubo):
        c = self.ctx
        io = self._io(ubo)
        _check(c.handle, c.lib.vkr_rtao_trace(c.handle, C.byref(io), c.stream()), "vkr_rtao_trace")

    def dispatchAtrous(self, ubo):
        c = self.ctx
        io = self._io(ubo)
        for i in range(self.iterations):
            _check(c.handle, c.lib.vkr_rtao_atrous(c.handle, C.byref(io), 1 << i, i, c.stream()), "vkr_rtao_atrous")

    def dispatchUpsample(self, ubo):
        c = self.ctx
        io = self._io(ubo)
        final_index = (self.iterations - 1) % 2      # RtaoComputePass.cpp:560-561
        _check(c.handle, c.lib.vkr_rtao_upsample(c.handle, C.byref(io), final_index, c.stream()), "vkr_rtao_upsample")

    # two-phase sharding (multi-GPU): K1 on interleaved stripes, gather history, A-Trous on a contiguous band
    def traceStripes(self, ubo, rank: int, nranks: int):
        c = self.ctx
        io = self._io(ubo)
        _check(c.handle, c.lib.vkr_rtao_trace_stripes(c.handle, C.byref(io), rank, nranks, c.stream()), "vkr_rtao_trace_stripes")

    def denoiseBand(self, ubo, y0: int, y1: int):
        c = self.ctx
        io = self._io(ubo)
        _check(c.handle, c.lib.vkr_rtao_denoise_band(c.handle, C.byref(io), y0, y1, c.stream()), "vkr_rtao_denoise_band")

    def gatherHistory(self):
        """All-gather the interleaved history stripes of the frame in flight (between traceStripes and denoiseBand)."""
        c = self.ctx
        _check(c.handle, c.lib.vkr_rtao_gather_history(c.handle, c.stream()), "vkr_rtao_gather_history")

    def historyCurr(self) -> torch.Tensor:
        """history[curr] of the frame in flight (what dispatchTrace writes), aliasing the library's memory."""
        c = self.ctx
        p = C.c_void_p()
        _check(c.handle, c.lib.vkr_rtao_history_curr(c.handle, C.byref(p)), "vkr_rtao_history_curr")
        return _alias_device_memory(p.value, (c.height, c.width), c.device)

    def endFrame(self):
        _check(self.ctx.handle, self.ctx.lib.vkr_rtao_end_frame(self.ctx.handle), "vkr_rtao_end_frame")

    def reset(self):
        c = self.ctx
        _check(c.handle, c.lib.vkr_rtao_reset(c.handle, c.stream()), "vkr_rtao_reset")

    def image(self, which: int) -> torch.Tensor:
        """Context-owned R16F image (0 history curr of the last frame, 1 history prev, 2 ping0, 3 ping1) as a
        torch tensor ALIASING the library's device memory (no copy): what the multi-GPU host all-gathers into."""
        c = self.ctx
        p = C.c_void_p()
        _check(c.handle, c.lib.vkr_rtao_image(c.handle, which, C.byref(p)), "vkr_rtao_image")
        return _alias_device_memory(p.value, (c.height, c.width), c.device)


class _DevMem:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<i2", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _alias_device_memory(ptr: int, shape, device) -> torch.Tensor:
    return torch.as_tensor(_DevMem(ptr, shape), device=device)


class ShadowRays:
    """pbr.frag:169-260 as screen-space passes over the G-buffer's worldPos / Ng."""

    def __init__(self, ctx: Context, gbuffer: GBuffer):
        self.ctx, self.gb = ctx, gbuffer
        self.sun_vis = torch.zeros((ctx.height, ctx.width), dtype=torch.float32, device=ctx.device)
        self.point_vis = torch.zeros((3, ctx.height, ctx.width), dtype=torch.float32, device=ctx.device)

    def sun(self, ubo: np.ndarray, rows=None) -> torch.Tensor:
        from .scenes import FrameState, UBO_DIR_LIGHT_PARAMS
        c, gb = self.ctx, self.gb
        L = (C.c_float * 3)(*[float(x) for x in FrameState.sun_dir_to_light(ubo)])
        y0, y1 = rows if rows is not None else (0, c.height)
        n = (y1 - y0) * c.width
        off = y0 * c.width
        _check(c.handle, c.lib.vkr_shadow_sun(
            c.handle, gb.world_pos.data_ptr() + off * 12, gb.geom_normal.data_ptr() + off * 12,
            gb.valid.data_ptr() + off, n, L, float(ubo[UBO_DIR_LIGHT_PARAMS]), float(ubo[UBO_DIR_LIGHT_PARAMS + 1]),
            self.sun_vis.data_ptr() + off * 4, c.stream()), "vkr_shadow_sun")
        return self.sun_vis

    def point(self, ubo: np.ndarray, light: int) -> torch.Tensor:
        from .scenes import UBO_LIGHT_POSITIONS
        c, gb = self.ctx, self.gb
        P = (C.c_float * 3)(*[float(x) for x in ubo[UBO_LIGHT_POSITIONS + 4 * light:UBO_LIGHT_POSITIONS + 4 * light + 3]])
        _check(c.handle, c.lib.vkr_shadow_point(c.handle, _ptr(gb.world_pos), _ptr(gb.geom_normal), _ptr(gb.valid),
                                                c.width * c.height, P, self.point_vis[light].data_ptr(), c.stream()),
               "vkr_shadow_point")
        return self.point_vis[light]


class Rendergraph:
    """Name-based pass DAG: AddPass, Compile (DFS topological sort on resource names, Rendergraph.cpp:150-204),
    Execute in that order with a per-pass CPU timer (Rendergraph.cpp:321-336)."""

    def __init__(self):
        self.passes: List[RenderPass] = []
        self.order: List[int] = []
        self.compiled = False

    def AddPass(self, p: RenderPass):
        self.passes.append(p)
        self.compiled = False

    def Compile(self):
        producers: Dict[str, int] = {}
        for i, p in enumerate(self.passes):
            for o in p.outputs:
                producers[o] = i
        state = [0] * len(self.passes)
        order: List[int] = []

        def visit(i):
            if state[i] == 2:
                return
            if state[i] == 1:
                raise RuntimeError("Rendergraph: cycle detected")
            state[i] = 1
            for name in self.passes[i].inputs:
                j = producers.get(name)
                if j is not None and j != i:
                    visit(j)
            state[i] = 2
            order.append(i)

        for i in range(len(self.passes)):
            visit(i)
        self.order, self.compiled = order, True

    def Execute(self, pctx: PassExecuteContext):
        if not self.compiled:
            raise RuntimeError("Rendergraph::Execute before Compile")
        for i in self.order:
            t0 = time.perf_counter()
            self.passes[i].execute(pctx)
            if pctx.stats is not None and isinstance(self.passes[i], RtaoComputePass):
                pctx.stats.rtaoMs = (time.perf_counter() - t0) * 1e3


def trace_stats(ctx: Context, reset: bool = True) -> _capi.TraceStats:
    out = _capi.TraceStats()
    _check(ctx.handle, ctx.lib.vkr_trace_stats_get(ctx.handle, C.byref(out), 1 if reset else 0), "vkr_trace_stats_get")
    return out


def trace_stats_enable(ctx: Context, enable: bool):
    _check(ctx.handle, ctx.lib.vkr_trace_stats_enable(ctx.handle, 1 if enable else 0), "vkr_trace_stats_enable")
