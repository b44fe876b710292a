"""ctypes wrapper of the CPU oracle (oracle/rtao_oracle.cpp).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by vulkanrenderer_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


class _Mesh(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("position_stride", C.c_uint32), ("vertex_count", C.c_uint32),
                ("indices", C.c_void_p), ("index_count", C.c_uint32), ("uvs", C.c_void_p), ("uv_stride", C.c_uint32),
                ("material_index", C.c_int32), ("opaque", C.c_uint32),
                ("normals", C.c_void_p), ("normal_stride", C.c_uint32)]


class _Instance(C.Structure):
    _fields_ = [("mesh_index", C.c_uint32), ("transform", C.c_float * 12)]


class _Alpha(C.Structure):
    _fields_ = [("alpha", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class _Scene(C.Structure):
    _fields_ = [("meshes", C.POINTER(_Mesh)), ("mesh_count", C.c_uint32), ("instances", C.POINTER(_Instance)),
                ("instance_count", C.c_uint32), ("material_params", C.c_void_p), ("alpha_textures", C.POINTER(_Alpha))]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, "rtao_oracle.cpp"), os.path.join(_HERE, "ibl_oracle.cpp")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, f32, u32, i32 = C.c_void_p, C.c_float, C.c_uint32, C.c_int
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [C.POINTER(_Scene), i32]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_set_instances.restype = i32
        L.orc_scene_set_instances.argtypes = [vp, C.POINTER(_Instance), u32]
        L.orc_scene_triangle_count.restype = u32
        L.orc_scene_triangle_count.argtypes = [vp]
        L.orc_stats.argtypes = [vp, C.POINTER(C.c_uint64), i32]
        L.orc_set_threads.argtypes = [i32]
        L.orc_max_threads.restype = i32
        L.orc_rtao_trace.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, i32, i32]
        L.orc_rtao_atrous.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, i32]
        L.orc_rtao_upsample.argtypes = [vp, i32, i32, vp, vp, vp]
        L.orc_rtao_execute.argtypes = [vp, vp, i32, i32, u32, vp, vp, vp, vp, vp, vp, u32, vp, vp]
        L.orc_shadow_sun.argtypes = [vp, vp, vp, vp, u32, vp, f32, f32, vp]
        L.orc_shadow_point.argtypes = [vp, vp, vp, vp, u32, vp, vp]
        L.orc_gbuffer.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp]
        L.orc_radical_inverse.restype = f32
        L.orc_radical_inverse.argtypes = [u32]
        L.orc_hammersley.argtypes = [u32, u32, vp]
        L.orc_offset_ray.argtypes = [vp, vp, vp]
        L.orc_build_tbn.argtypes = [vp, vp, vp]
        for name in ("orc_sin", "orc_cos", "orc_exp", "orc_hash11"):
            getattr(L, name).restype = f32
            getattr(L, name).argtypes = [f32]
        L.orc_smoothstep.restype = f32
        L.orc_smoothstep.argtypes = [f32, f32, f32]
        L.orc_inverse_m4.argtypes = [vp, vp]
        L.orc_trace_ao_ray.restype = f32
        L.orc_trace_ao_ray.argtypes = [vp, vp, vp, f32, f32, C.POINTER(i32)]
        L.orc_query_ao.restype = i32
        L.orc_query_ao.argtypes = [vp, vp, vp, f32, f32, C.POINTER(f32)]
        L.orc_shadow_ray.restype = f32
        L.orc_shadow_ray.argtypes = [vp, vp, vp]
        L.orc_tex_linear.restype = f32
        L.orc_tex_linear.argtypes = [vp, i32, i32, f32, f32]
        L.orc_tex_nearest.restype = f32
        L.orc_tex_nearest.argtypes = [vp, i32, i32, f32, f32]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data


def _instances(instance_mesh, instance_transform):
    n = len(instance_mesh)
    inst = (_Instance * max(1, n))()
    for i in range(n):
        inst[i].mesh_index = int(instance_mesh[i])
        inst[i].transform[:] = [float(x) for x in instance_transform[i]]
    return inst


class OracleScene:
    """The oracle's own acceleration structure over a SceneData (median-split BVH, or brute force)."""

    def __init__(self, scene, brute_force: bool = False):
        L = lib()
        keep = []
        meshes = (_Mesh * max(1, len(scene.meshes)))()
        opaque = scene.mesh_opaque_flags()
        for i, m in enumerate(scene.meshes):
            v = _f32(m.vertices)
            idx = np.ascontiguousarray(m.indices, dtype=np.uint32)
            keep += [v, idx]
            meshes[i].positions = v.ctypes.data
            meshes[i].position_stride = v.shape[1] * 4
            meshes[i].vertex_count = v.shape[0]
            meshes[i].indices = idx.ctypes.data
            meshes[i].index_count = idx.shape[0]
            meshes[i].uvs = v.ctypes.data + 36
            meshes[i].uv_stride = v.shape[1] * 4
            meshes[i].material_index = int(m.material_index)
            meshes[i].opaque = int(opaque[i])
            meshes[i].normals = v.ctypes.data + 12
            meshes[i].normal_stride = v.shape[1] * 4
        inst = _instances(scene.instance_mesh, scene.instance_transform)
        mp = _f32(scene.material_params())
        tex = (_Alpha * 256)()
        for i, mat in enumerate(scene.materials[:256]):
            if mat.alpha_texture is not None:
                a = np.ascontiguousarray(mat.alpha_texture, dtype=np.uint8)
                keep.append(a)
                tex[i].alpha = a.ctypes.data
                tex[i].width, tex[i].height = a.shape[1], a.shape[0]
        d = _Scene(meshes, len(scene.meshes), inst, len(scene.instance_mesh), mp.ctypes.data, tex)
        self.handle = C.c_void_p(L.orc_scene_create(C.byref(d), 1 if brute_force else 0))
        self.L = L

    def __del__(self):
        try:
            if self.handle:
                self.L.orc_scene_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def set_instances(self, instance_mesh, instance_transform):
        rc = self.L.orc_scene_set_instances(self.handle, _instances(instance_mesh, instance_transform), len(instance_mesh))
        if rc != 0:
            raise RuntimeError("instance count mismatch: RayTracingContext must be re-initialized")

    def triangle_count(self) -> int:
        return int(self.L.orc_scene_triangle_count(self.handle))

    def stats(self, reset=True):
        out = (C.c_uint64 * 3)()
        self.L.orc_stats(self.handle, out, 1 if reset else 0)
        return {"rays": int(out[0]), "nodes": int(out[1]), "tris": int(out[2])}

    # ---- passes -----------------------------------------------------------------------------------
    def gbuffer(self, ubo, W, H, samples=1):
        depth = np.zeros((H, W), dtype=np.uint16)
        normal = np.zeros((H, W, 4), dtype=np.uint16)
        wp = np.zeros((H, W, 3), dtype=np.float32)
        ng = np.zeros((H, W, 3), dtype=np.float32)
        valid = np.zeros((H, W), dtype=np.uint8)
        u = _f32(ubo)
        if samples == 1:
            self.L.orc_gbuffer(self.handle, _p(u), W, H, _p(depth), _p(normal), _p(wp), _p(ng), _p(valid))
        else:
            self.L.orc_gbuffer_msaa.restype = None
            self.L.orc_gbuffer_msaa.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
            self.L.orc_gbuffer_msaa(self.handle, _p(u), W, H, int(samples), _p(depth), _p(normal), _p(wp), _p(ng), _p(valid))
        return depth, normal, wp, ng, valid

    def trace(self, ubo, depth, hist_prev, y0=0, y1=None, want_mask=False):
        H, W = depth.shape
        u = _f32(ubo)
        out = np.zeros((H, W), dtype=np.uint16)
        mask = np.zeros((H, W), dtype=np.uint32) if want_mask else None
        self.L.orc_rtao_trace(self.handle, _p(u), W, H, _p(depth), _p(hist_prev), _p(out), _p(mask), y0,
                              H if y1 is None else y1)
        return (out, mask) if want_mask else out

    def shadow_sun(self, wp, ng, valid, L3, radius, samples):
        n = valid.size
        out = np.zeros(valid.shape, dtype=np.float32)
        l = _f32(L3)
        wpa, nga, va = _f32(wp), _f32(ng), np.ascontiguousarray(valid, dtype=np.uint8)
        self.L.orc_shadow_sun(self.handle, _p(wpa), _p(nga), _p(va), n, _p(l), float(radius), float(samples), _p(out))
        return out

    def shadow_point(self, wp, ng, valid, P3):
        n = valid.size
        out = np.zeros(valid.shape, dtype=np.float32)
        p = _f32(P3)
        wpa, nga, va = _f32(wp), _f32(ng), np.ascontiguousarray(valid, dtype=np.uint8)
        self.L.orc_shadow_point(self.handle, _p(wpa), _p(nga), _p(va), n, _p(p), _p(out))
        return out

    def query_ao(self, o, d, tmin, tmax):
        t = C.c_float()
        oa, da = _f32(o), _f32(d)
        hit = self.L.orc_query_ao(self.handle, _p(oa), _p(da), tmin, tmax, C.byref(t))
        return bool(hit), float(t.value)

    def trace_ao_ray(self, o, d, tmin, tmax):
        occ = C.c_int()
        oa, da = _f32(o), _f32(d)
        v = self.L.orc_trace_ao_ray(self.handle, _p(oa), _p(da), tmin, tmax, C.byref(occ))
        return float(v), int(occ.value)

    def shadow_ray(self, o, d):
        oa, da = _f32(o), _f32(d)
        return float(self.L.orc_shadow_ray(self.handle, _p(oa), _p(da)))


class _Node(C.Structure):
    _fields_ = [("parent", C.c_int32), ("local", C.c_float * 16), ("first_instance", C.c_uint32),
                ("instance_count", C.c_uint32)]


def instances_from_nodes(parents, locals_, first_instance, instance_count, model):
    """Renderer::rebuildRayTracingInstances on the CPU -> (N_inst, 12) float32 row-major 3x4 transforms."""
    n = len(parents)
    nodes = (_Node * max(1, n))()
    for i in range(n):
        nodes[i].parent = int(parents[i])
        nodes[i].local[:] = [float(x) for x in np.asarray(locals_[i], dtype=np.float32).reshape(16)]
        nodes[i].first_instance = int(first_instance[i])
        nodes[i].instance_count = int(instance_count[i])
    total = int(np.sum(instance_count))
    out = np.zeros((total, 12), dtype=np.float32)
    m = _f32(np.asarray(model, dtype=np.float32).reshape(16))
    L = lib()
    L.orc_instances_from_nodes.restype = None
    L.orc_instances_from_nodes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.orc_instances_from_nodes(C.cast(nodes, C.c_void_p), n, _p(m), _p(out))
    return out


SLOT_DTYPE = np.dtype([("node_linear_index", "<u4"), ("mesh_index", "<u4"), ("mat_index", "<u4"), ("double_sided", "<u4")])
MESH_INFO_DTYPE = np.dtype([("vertex_offset", "<u4"), ("first_index", "<u4"), ("index_count", "<u4")])
DRAW_CMD_DTYPE = np.dtype([("index_count", "<u4"), ("instance_count", "<u4"), ("first_index", "<u4"),
                           ("vertex_offset", "<i4"), ("first_instance", "<u4")])
SPAN_DTYPE = np.dtype([("double_sided", "<u4"), ("mat_index", "<u4"), ("first_command", "<u4"), ("draw_count", "<u4")])


def sort_draw_slots(slots):
    """The std::stable_sort of FrameManager::buildSharedDrawSlots (FrameManager.cpp:110-115)."""
    L = lib()
    s = np.array(slots, dtype=SLOT_DTYPE, copy=True)
    L.orc_sort_draw_slots.restype = None
    L.orc_sort_draw_slots.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_sort_draw_slots(s.ctypes.data if len(s) else None, len(s))
    return s


def frustum_planes(view_proj):
    L = lib()
    m = _f32(np.asarray(view_proj).reshape(16))
    out = np.zeros(24, dtype=np.float32)
    L.orc_frustum_planes.restype = None
    L.orc_frustum_planes.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_frustum_planes(_p(m), _p(out))
    return out.reshape(6, 4)


def prepare_indirect(node_parent, node_local, sorted_slots, mesh_infos, model, max_draws, view_proj=None, mesh_bounds=None):
    """FrameManager::prepareSharedOpaqueIndirect (FrameManager.cpp:118-175); returns
    (draw_count, draw_data[n,16], commands[n], spans)."""
    L = lib()
    par = np.ascontiguousarray(node_parent, dtype=np.int32)
    loc = _f32(np.asarray(node_local, dtype=np.float32).reshape(-1, 16))
    sl = np.ascontiguousarray(sorted_slots, dtype=SLOT_DTYPE)
    mi = np.ascontiguousarray(mesh_infos, dtype=MESH_INFO_DTYPE)
    m = _f32(np.asarray(model).reshape(16))
    md = max(1, int(max_draws))
    dd = np.zeros((md, 16), dtype=np.float32)
    cm = np.zeros(md, dtype=DRAW_CMD_DTYPE)
    cap = max(1, len(sl))
    sp = np.zeros(cap, dtype=SPAN_DTYPE)
    vp = None if view_proj is None else _f32(np.asarray(view_proj).reshape(16))
    mb = None if mesh_bounds is None else _f32(np.asarray(mesh_bounds, dtype=np.float32).reshape(-1, 6))
    ns = C.c_uint32(0)
    L.orc_prepare_indirect.restype = C.c_uint32
    L.orc_prepare_indirect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                       C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_uint32, C.POINTER(C.c_uint32)]
    n = L.orc_prepare_indirect(par.ctypes.data if len(par) else None, _p(loc) if len(par) else None, len(par),
                               sl.ctypes.data if len(sl) else None, len(sl), mi.ctypes.data if len(mi) else None, len(mi),
                               _p(m), int(max_draws), _p(vp) if vp is not None else None,
                               _p(mb) if mb is not None else None, _p(dd), cm.ctypes.data, sp.ctypes.data, cap, C.byref(ns))
    return int(n), dd[:n], cm[:n], sp[:ns.value]


# ---------------------------------------------------------------- IBL precompute (oracle/ibl_oracle.cpp)
def prefilter_mip_count(size: int) -> int:
    return int(np.floor(np.log2(size))) + 1            # IblPrecompute.cpp:68-71


def prefilter_mip_offsets(size: int):
    """Offsets (in halves) of each mip in the mip-major packed RGBA16F prefilter buffer, plus the total."""
    offs, o = [], 0
    for m in range(prefilter_mip_count(size)):
        offs.append(o)
        sm = max(1, size >> m)
        o += 6 * sm * sm * 4
    return offs, o


def ibl_equirect_to_cube(equirect, cube_size):
    L = lib()
    eq = _f32(equirect)
    h, w = eq.shape[0], eq.shape[1]
    out = np.zeros((6, cube_size, cube_size, 4), dtype=np.float32)
    L.orc_ibl_equirect_to_cube.restype = None
    L.orc_ibl_equirect_to_cube.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.orc_ibl_equirect_to_cube(_p(eq), w, h, _p(out), cube_size)
    return out


def ibl_irradiance(cube, size, order=0):
    L = lib()
    c = _f32(cube)
    out = np.zeros((6, size, size, 4), dtype=np.uint16)
    L.orc_ibl_irradiance.restype = None
    L.orc_ibl_irradiance.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.orc_ibl_irradiance(_p(c), c.shape[1], _p(out), size, order)
    return out


def ibl_prefilter(cube, size, order=0):
    L = lib()
    c = _f32(cube)
    _, total = prefilter_mip_offsets(size)
    out = np.zeros(total, dtype=np.uint16)
    L.orc_ibl_prefilter.restype = None
    L.orc_ibl_prefilter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.orc_ibl_prefilter(_p(c), c.shape[1], _p(out), size, order)
    return out


def ibl_brdf_lut(size):
    L = lib()
    out = np.zeros((size, size, 2), dtype=np.uint16)
    L.orc_ibl_brdf_lut.restype = None
    L.orc_ibl_brdf_lut.argtypes = [C.c_void_p, C.c_int]
    L.orc_ibl_brdf_lut(_p(out), size)
    return out


def ibl_sample_cube(cube, direction):
    L = lib()
    c = _f32(cube)
    d = _f32(direction)
    out = np.zeros(3, dtype=np.float32)
    L.orc_ibl_sample_cube.restype = None
    L.orc_ibl_sample_cube.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_ibl_sample_cube(_p(c), c.shape[1], _p(d), _p(out))
    return out


def ibl_sample_equirect(equirect, direction):
    L = lib()
    eq = _f32(equirect)
    d = _f32(direction)
    out = np.zeros(3, dtype=np.float32)
    L.orc_ibl_sample_equirect.restype = None
    L.orc_ibl_sample_equirect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_ibl_sample_equirect(_p(eq), eq.shape[1], eq.shape[0], _p(d), _p(out))
    return out


def ibl_texel_dir(face, x, y, size):
    L = lib()
    out = np.zeros(3, dtype=np.float32)
    L.orc_ibl_texel_dir.restype = None
    L.orc_ibl_texel_dir.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_ibl_texel_dir(face, x, y, size, _p(out))
    return out


def det_atan2(y, x):
    L = lib()
    L.orc_atan2.restype = C.c_float
    L.orc_atan2.argtypes = [C.c_float, C.c_float]
    return float(L.orc_atan2(float(y), float(x)))


def det_asin(x):
    L = lib()
    L.orc_asin.restype = C.c_float
    L.orc_asin.argtypes = [C.c_float]
    return float(L.orc_asin(float(x)))


def atrous(ubo, step, src, depth, normal, y0=0, y1=None):
    H, W = depth.shape
    out = np.zeros((H, W), dtype=np.uint16)
    u = _f32(ubo)
    lib().orc_rtao_atrous(_p(u), W, H, int(step), _p(src), _p(depth), _p(normal), _p(out), y0, H if y1 is None else y1)
    return out


def upsample(ubo, src, normal):
    H, W = src.shape
    out = np.zeros((H, W), dtype=np.uint16)
    u = _f32(ubo)
    lib().orc_rtao_upsample(_p(u), W, H, _p(src), _p(normal), _p(out))
    return out


class OracleRtaoPass:
    """RtaoComputePass on the CPU: owns history x2 / ping-pong x2 (cleared to 1.0, FrameManager.cpp:1057) and the
    frame-in-flight index (FrameManager.cpp:1155-1165)."""

    ONE = np.float16(1.0).view(np.uint16)

    def __init__(self, scene: OracleScene, W: int, H: int, iterations: int):
        self.scene, self.W, self.H = scene, W, H
        self.iterations = max(1, iterations)
        self.reset()

    def reset(self):
        self.hist = [np.full((self.H, self.W), self.ONE, dtype=np.uint16) for _ in range(2)]
        self.ping = [np.full((self.H, self.W), self.ONE, dtype=np.uint16) for _ in range(2)]
        self.parity = 0
        self.last_curr = 0

    def execute(self, ubo, depth, normal, want_mask=False):
        u = _f32(ubo)
        full = np.zeros((self.H, self.W), dtype=np.uint16)
        mask = np.zeros((self.H, self.W), dtype=np.uint32) if want_mask else None
        c = self.parity
        lib().orc_rtao_execute(self.scene.handle, _p(u), self.W, self.H, self.iterations, _p(depth), _p(normal),
                               _p(self.hist[0]), _p(self.hist[1]), _p(self.ping[0]), _p(self.ping[1]), c, _p(full),
                               _p(mask))
        self.last_curr = c
        self.parity ^= 1
        return (full, mask) if want_mask else full


def set_variant(v: int):
    """Contract-departure switches of the oracle (tests only; 0 = the contract): 1 IEEE divisions in A-Trous, 2 A-Trous
    fetches through the samplers at the computed uv, 4 first-found AO hits, 8 (with 4) reversed traversal order."""
    lib().orc_set_variant(int(v))


def set_threads(n: int):
    lib().orc_set_threads(int(n))


def max_threads() -> int:
    return int(lib().orc_max_threads())
