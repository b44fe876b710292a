"""ctypes binding of libvkr_rt.so (include/vkr_rt.h).  No fallback: if the HIP library is missing or does not
export a declared symbol, importing this module raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VKR_LIB_PATH: experiment hook (tools/variants.sh) to load a differently-built copy of the same library
LIB_PATH = os.environ.get("VKR_LIB_PATH") or os.path.join(_HERE, "lib", "libvkr_rt.so")

VKR_OK, VKR_ERR_INVALID_ARG, VKR_ERR_STATE, VKR_ERR_HIP, VKR_ERR_OOM, VKR_ERR_LIMIT, VKR_ERR_RCCL = range(7)
MAX_MATERIALS = 256
UBO_FLOATS = 120


class CreateDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32)]


class MeshDesc(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("position_stride", C.c_uint32), ("vertex_count", C.c_uint32),
                ("indices", C.c_void_p), ("index_count", C.c_uint32),
                ("uvs", C.c_void_p), ("uv_stride", C.c_uint32),
                ("material_index", C.c_int32), ("opaque", C.c_uint32),
                ("normals", C.c_void_p), ("normal_stride", C.c_uint32)]


class Instance(C.Structure):
    _fields_ = [("mesh_index", C.c_uint32), ("transform", C.c_float * 12)]


class AlphaTexture(C.Structure):
    _fields_ = [("alpha", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(MeshDesc)), ("mesh_count", C.c_uint32),
                ("instances", C.POINTER(Instance)), ("instance_count", C.c_uint32),
                ("material_params", C.c_void_p), ("alpha_textures", C.POINTER(AlphaTexture))]


class SceneNode(C.Structure):
    _fields_ = [("parent", C.c_int32), ("local", C.c_float * 16), ("first_instance", C.c_uint32),
                ("instance_count", C.c_uint32)]


class DrawSlot(C.Structure):               # SharedOpaqueDrawSlot
    _fields_ = [("node_linear_index", C.c_uint32), ("mesh_index", C.c_uint32), ("mat_index", C.c_uint32),
                ("double_sided", C.c_uint32)]


class MeshDrawInfo(C.Structure):           # GlobalMeshBuffer.h:12-16
    _fields_ = [("vertex_offset", C.c_uint32), ("first_index", C.c_uint32), ("index_count", C.c_uint32)]


class DrawIndexedIndirectCommand(C.Structure):
    _fields_ = [("index_count", C.c_uint32), ("instance_count", C.c_uint32), ("first_index", C.c_uint32),
                ("vertex_offset", C.c_int32), ("first_instance", C.c_uint32)]


class BucketSpan(C.Structure):             # SharedOpaqueBucketSpan
    _fields_ = [("double_sided", C.c_uint32), ("mat_index", C.c_uint32), ("first_command", C.c_uint32),
                ("draw_count", C.c_uint32)]


class IndirectDesc(C.Structure):
    _fields_ = [("node_parent", C.c_void_p), ("node_local", C.c_void_p), ("node_count", C.c_uint32),
                ("slots", C.c_void_p), ("slot_count", C.c_uint32), ("mesh_infos", C.c_void_p), ("mesh_count", C.c_uint32),
                ("mesh_bounds", C.c_void_p), ("max_draws", C.c_uint32)]


class BvhInfo(C.Structure):
    _fields_ = [("triangle_count", C.c_uint32), ("node_count", C.c_uint32), ("height", C.c_uint32),
                ("morton_bits", C.c_uint32), ("binary_height", C.c_uint32), ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3)]


class RtaoFrameIO(C.Structure):
    _fields_ = [("ubo", C.c_float * UBO_FLOATS), ("atrous_iterations", C.c_uint32),
                ("linear_depth_r16f", C.c_void_p), ("normal_rgba16f", C.c_void_p),
                ("rtao_full_r16f", C.c_void_p), ("ray_hit_mask", C.c_void_p)]


class TraceStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("wave_node_iters", C.c_uint64),
                ("wave_tri_iters", C.c_uint64), ("overflow_tiles", C.c_uint64)]


# every symbol include/vkr_rt.h declares: name -> (restype, argtypes)
_P = C.c_void_p
# vkr_allgather_fn: int (*)(void* user, const void* send, void* recv, size_t bytes_per_rank, void* stream)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
SYMBOLS = {
    "vkr_create": (C.c_int, [C.POINTER(CreateDesc), C.POINTER(_P)]),
    "vkr_destroy": (None, [_P]),
    "vkr_last_error": (C.c_char_p, [_P]),
    "vkr_sync": (C.c_int, [_P, _P]),
    "vkr_scene_upload": (C.c_int, [_P, C.POINTER(SceneDesc)]),
    "vkr_bvh_build": (C.c_int, [_P, _P]),
    "vkr_bvh_refit": (C.c_int, [_P, C.POINTER(Instance), C.c_uint32, _P]),
    "vkr_hierarchy_upload": (C.c_int, [_P, C.POINTER(SceneNode), C.c_uint32]),
    "vkr_bvh_refit_model": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int), _P]),
    "vkr_instances_download": (C.c_int, [_P, _P]),
    "vkr_bvh_info_get": (C.c_int, [_P, C.POINTER(BvhInfo)]),
    "vkr_bvh_download": (C.c_int, [_P, _P, _P]),
    "vkr_rtao_execute": (C.c_int, [_P, C.POINTER(RtaoFrameIO), _P]),
    "vkr_rtao_execute_band": (C.c_int, [_P, C.POINTER(RtaoFrameIO), C.c_uint32, C.c_uint32, _P]),
    "vkr_rtao_trace_stripes": (C.c_int, [_P, C.POINTER(RtaoFrameIO), C.c_uint32, C.c_uint32, _P]),
    "vkr_rtao_denoise_band": (C.c_int, [_P, C.POINTER(RtaoFrameIO), C.c_uint32, C.c_uint32, _P]),
    "vkr_rtao_history_curr": (C.c_int, [_P, C.POINTER(_P)]),
    "vkr_rtao_trace": (C.c_int, [_P, C.POINTER(RtaoFrameIO), _P]),
    "vkr_rtao_atrous": (C.c_int, [_P, C.POINTER(RtaoFrameIO), C.c_uint32, C.c_uint32, _P]),
    "vkr_rtao_upsample": (C.c_int, [_P, C.POINTER(RtaoFrameIO), C.c_uint32, _P]),
    "vkr_rtao_end_frame": (C.c_int, [_P]),
    "vkr_rtao_reset": (C.c_int, [_P, _P]),
    "vkr_rtao_image": (C.c_int, [_P, C.c_uint32, C.POINTER(_P)]),
    "vkr_comm_attach": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    "vkr_comm_attach_custom": (C.c_int, [_P, ALLGATHER_FN, _P, C.c_uint32, C.c_uint32]),
    "vkr_rtao_gather_history": (C.c_int, [_P, _P]),
    "vkr_gather_bands": (C.c_int, [_P, C.POINTER(_P), C.c_uint32, _P]),
    "vkr_shadow_sun": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.POINTER(C.c_float), C.c_float, C.c_float, _P, _P]),
    "vkr_shadow_sun_noalloc": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.POINTER(C.c_float), C.c_float, C.c_float, _P, _P]),
    "vkr_shadow_point": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.POINTER(C.c_float), _P, _P]),
    "vkr_gbuffer_primary": (C.c_int, [_P, C.POINTER(C.c_float), _P, _P, _P, _P, _P, _P]),
    "vkr_gbuffer_primary_msaa": (C.c_int, [_P, C.POINTER(C.c_float), C.c_uint32, _P, _P, _P, _P, _P, _P]),
    "vkr_indirect_upload": (C.c_int, [_P, C.POINTER(IndirectDesc)]),
    "vkr_indirect_slots_download": (C.c_int, [_P, _P]),
    "vkr_indirect_prepare": (C.c_int, [_P, C.POINTER(C.c_float), _P, _P, _P, C.c_uint32, _P]),
    "vkr_indirect_result": (C.c_int, [_P, C.POINTER(C.c_uint32), _P, C.c_uint32, C.POINTER(C.c_uint32), _P]),
    "vkr_ibl_equirect_to_cube": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, _P]),
    "vkr_ibl_irradiance": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P]),
    "vkr_ibl_prefilter": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P]),
    "vkr_ibl_brdf_lut": (C.c_int, [_P, _P, C.c_uint32, _P]),
    "vkr_ibl_compute": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P]),
    "vkr_ibl_prefilter_mip_count": (C.c_uint32, [C.c_uint32]),
    "vkr_ibl_prefilter_texels": (C.c_size_t, [C.c_uint32]),
    "vkr_trace_stats_enable": (C.c_int, [_P, C.c_int]),
    "vkr_trace_stats_get": (C.c_int, [_P, C.POINTER(TraceStats), C.c_int]),
}


def load_library(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C vulkanrenderer_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib
