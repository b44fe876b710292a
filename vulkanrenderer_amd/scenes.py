"""Deterministic procedural scenes + camera/UBO helpers.

The reference's assets (Bistro, Sponza) are not in the container, so workloads are seeded
procedural scenes with matched statistics ("bistro-class": ~2.8 M world-space triangles, ~2.7 k instances,
80/15/5 % opaque/MASK/BLEND by triangle; counts from docs/GPU_DRIVEN_RENDERING_DESIGN.md:9-10).

Data layout follows the reference's CPU-side model: one ``Mesh`` per glTF primitive with interleaved
60-byte ``Vertex`` records (app/include/Resource/model/Vertex.h:15-21: pos@0, normal@12, color@24,
texCoord@36, tangent@44), uint32 indices, a material index, and a flat instance list in the DFS order of
``Renderer::rebuildRayTracingInstances`` (Renderer.cpp:503-527).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

VERTEX_FLOATS = 15          # sizeof(Vertex) = 60 bytes
VERTEX_STRIDE = 60
UV_OFFSET_FLOATS = 9        # texCoord at byte 36
NORMAL_OFFSET_FLOATS = 3    # normal at byte 12 (Vertex.h: pos, normal, color, texCoord, tangent)
MAX_MATERIALS = 256         # AppConfig::MAX_REFLECTION_MATERIAL_COUNT

ALPHA_OPAQUE, ALPHA_MASK, ALPHA_BLEND = 0, 1, 2


@dataclass
class Mesh:
    vertices: np.ndarray        # (V, 15) float32, reference Vertex layout
    indices: np.ndarray         # (I,) uint32, mesh-local
    material_index: int
    keep_normals: bool = False  # True: vertices[:, 3:6] already holds the file's NORMAL attribute

    def __post_init__(self):
        # area-weighted vertex normals, as a glTF exporter would have written them (shared vertices come out smooth)
        v, idx = self.vertices, self.indices.reshape(-1, 3)
        if len(idx) == 0 or self.keep_normals:
            return
        p = v[:, :3].astype(np.float64)
        fn = np.cross(p[idx[:, 1]] - p[idx[:, 0]], p[idx[:, 2]] - p[idx[:, 0]])
        acc = np.zeros_like(p)
        for k in range(3):
            np.add.at(acc, idx[:, k], fn)
        ln = np.linalg.norm(acc, axis=1, keepdims=True)
        acc = np.where(ln > 1e-20, acc / np.maximum(ln, 1e-20), np.array([0.0, 1.0, 0.0]))
        v[:, 3:6] = acc.astype(np.float32)


@dataclass
class Material:
    alpha_mode: int = ALPHA_OPAQUE       # Material.h: AlphaMode
    alpha_cutoff: float = 0.5
    alpha_texture: Optional[np.ndarray] = None   # (H, W) uint8 = LOD-0 alpha plane of baseColor


@dataclass
class SceneData:
    meshes: List[Mesh]
    materials: List[Material]
    instance_mesh: np.ndarray            # (N,) uint32
    instance_transform: np.ndarray       # (N, 12) float32, 3x4 row-major world-from-mesh
    name: str = "scene"
    cameras: list = field(default_factory=list)   # suggested (eye, target) pairs

    # Renderer.cpp:80-87: opaque iff the material is missing or its alphaMode is Opaque
    def mesh_opaque_flags(self) -> np.ndarray:
        out = np.ones(len(self.meshes), dtype=np.uint32)
        for i, m in enumerate(self.meshes):
            has = 0 <= m.material_index < len(self.materials)
            out[i] = 1 if (not has or self.materials[m.material_index].alpha_mode == ALPHA_OPAQUE) else 0
        return out

    # FrameManager.cpp:1354-1371: vec4(alphaCutoff, alphaMode, 0, 0), defaults (0.5, 0)
    def material_params(self) -> np.ndarray:
        mp = np.zeros((MAX_MATERIALS, 4), dtype=np.float32)
        mp[:, 0] = 0.5
        for i, m in enumerate(self.materials[:MAX_MATERIALS]):
            mp[i, 0] = m.alpha_cutoff
            mp[i, 1] = float(m.alpha_mode)
        return mp

    def triangle_count(self) -> int:
        per_mesh = np.array([len(m.indices) // 3 for m in self.meshes], dtype=np.int64)
        return int(per_mesh[self.instance_mesh].sum()) if len(self.instance_mesh) else 0

    def class_fractions(self):
        per_mesh = np.array([len(m.indices) // 3 for m in self.meshes], dtype=np.int64)
        mode = np.array([self.materials[m.material_index].alpha_mode
                         if 0 <= m.material_index < len(self.materials) else 0 for m in self.meshes])
        tri = per_mesh[self.instance_mesh]
        md = mode[self.instance_mesh]
        tot = max(1, tri.sum())
        return {k: float(tri[md == v].sum()) / tot for k, v in (("opaque", 0), ("mask", 1), ("blend", 2))}


# --------------------------------------------------------------------------------------------------
# mesh primitives
# --------------------------------------------------------------------------------------------------
def _make_vertices(pos: np.ndarray, uv: np.ndarray) -> np.ndarray:
    v = np.zeros((pos.shape[0], VERTEX_FLOATS), dtype=np.float32)
    v[:, 0:3] = pos
    v[:, 3:6] = (0.0, 1.0, 0.0)
    v[:, 6:9] = 1.0
    v[:, 9:11] = uv
    return v


def _grid_indices(nx: int, ny: int, offset: int = 0) -> np.ndarray:
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    a = (j * (nx + 1) + i).ravel() + offset
    b, c, d = a + 1, a + (nx + 1), a + (nx + 2)
    return np.stack([a, c, b, b, c, d], axis=1).ravel().astype(np.uint32)


def grid_mesh(nx: int, ny: int, material: int, rng: Optional[np.random.Generator] = None, bump: float = 0.0) -> Mesh:
    """Unit square in the XZ plane ([0,1]^2, y = 0 + optional bump), nx*ny*2 triangles."""
    u, w = np.meshgrid(np.linspace(0, 1, nx + 1), np.linspace(0, 1, ny + 1), indexing="xy")
    y = np.zeros_like(u)
    if bump > 0.0 and rng is not None:
        y = (rng.random(u.shape).astype(np.float32) - 0.5) * bump
    pos = np.stack([u.ravel(), y.ravel(), w.ravel()], axis=1).astype(np.float32)
    uv = np.stack([u.ravel(), w.ravel()], axis=1).astype(np.float32)
    return Mesh(_make_vertices(pos, uv), _grid_indices(nx, ny), material)


def box_mesh(sub: int, material: int) -> Mesh:
    """Unit cube [0,1]^3, six faces of sub*sub quads."""
    verts, inds, off = [], [], 0
    u, w = np.meshgrid(np.linspace(0, 1, sub + 1), np.linspace(0, 1, sub + 1), indexing="xy")
    u, w = u.ravel(), w.ravel()
    zeros, ones = np.zeros_like(u), np.ones_like(u)
    faces = [(u, w, zeros), (u, w, ones), (u, zeros, w), (u, ones, w), (zeros, u, w), (ones, u, w)]
    for f in faces:
        pos = np.stack(f, axis=1).astype(np.float32)
        uv = np.stack([u, w], axis=1).astype(np.float32)
        verts.append(_make_vertices(pos, uv))
        inds.append(_grid_indices(sub, sub, off))
        off += pos.shape[0]
    return Mesh(np.concatenate(verts), np.concatenate(inds), material)


def cylinder_mesh(seg: int, rings: int, material: int) -> Mesh:
    """Radius 0.5, y in [0,1], open-ended tube + two fan caps."""
    th = np.linspace(0.0, 2.0 * math.pi, seg + 1)
    ys = np.linspace(0.0, 1.0, rings + 1)
    T, Y = np.meshgrid(th, ys, indexing="xy")
    pos = np.stack([0.5 * np.cos(T).ravel(), Y.ravel(), 0.5 * np.sin(T).ravel()], axis=1).astype(np.float32)
    uv = np.stack([(T / (2 * math.pi)).ravel(), Y.ravel()], axis=1).astype(np.float32)
    idx = [_grid_indices(seg, rings)]
    off = pos.shape[0]
    caps_pos, caps_uv = [], []
    for yv in (0.0, 1.0):
        ring = np.stack([0.5 * np.cos(th[:-1]), np.full(seg, yv), 0.5 * np.sin(th[:-1])], axis=1)
        cpos = np.concatenate([[[0.0, yv, 0.0]], ring]).astype(np.float32)
        caps_pos.append(cpos)
        caps_uv.append((cpos[:, [0, 2]] + 0.5).astype(np.float32))
        k = np.arange(seg)
        idx.append(np.stack([np.full(seg, off), off + 1 + k, off + 1 + (k + 1) % seg], axis=1).ravel().astype(np.uint32))
        off += cpos.shape[0]
    pos = np.concatenate([pos] + caps_pos)
    uv = np.concatenate([uv] + caps_uv)
    return Mesh(_make_vertices(pos, uv), np.concatenate(idx), material)


def sphere_mesh(seg: int, rings: int, material: int) -> Mesh:
    """UV sphere, radius 0.5 centred at (0, 0.5, 0)."""
    th = np.linspace(0.0, 2.0 * math.pi, seg + 1)
    ph = np.linspace(0.0, math.pi, rings + 1)
    T, P = np.meshgrid(th, ph, indexing="xy")
    pos = np.stack([0.5 * np.sin(P) * np.cos(T), 0.5 + 0.5 * np.cos(P), 0.5 * np.sin(P) * np.sin(T)], axis=-1)
    uv = np.stack([T / (2 * math.pi), P / math.pi], axis=-1)
    return Mesh(_make_vertices(pos.reshape(-1, 3).astype(np.float32), uv.reshape(-1, 2).astype(np.float32)),
                _grid_indices(seg, rings), material)


def foliage_mesh(cards: int, material: int, rng: np.random.Generator) -> Mesh:
    """`cards` random quads (2 triangles each) inside a unit ball centred at (0, 0.5, 0): leaf cards for MASK."""
    c = rng.normal(size=(cards, 3))
    c /= np.maximum(np.linalg.norm(c, axis=1, keepdims=True), 1e-6)
    c *= (rng.random((cards, 1)) ** (1.0 / 3.0)) * 0.5
    c[:, 1] += 0.5
    a = rng.normal(size=(cards, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.normal(size=(cards, 3)); b -= a * (a * b).sum(1, keepdims=True); b /= np.linalg.norm(b, axis=1, keepdims=True)
    s = (0.04 + 0.06 * rng.random((cards, 1)))
    corners = np.stack([c - a * s - b * s, c + a * s - b * s, c - a * s + b * s, c + a * s + b * s], axis=1)
    pos = corners.reshape(-1, 3).astype(np.float32)
    uv0 = rng.random((cards, 1, 2)) * 0.75
    quad = np.array([[0, 0], [0.25, 0], [0, 0.25], [0.25, 0.25]], dtype=np.float64)[None]
    uv = (uv0 + quad).reshape(-1, 2).astype(np.float32)
    k = (np.arange(cards) * 4)[:, None]
    idx = (k + np.array([0, 2, 1, 1, 2, 3])[None]).ravel().astype(np.uint32)
    return Mesh(_make_vertices(pos, uv), idx, material)


def alpha_blobs(size: int, rng: np.random.Generator, coverage: float = 0.55) -> np.ndarray:
    """Procedural leaf-like alpha plane: smooth noise thresholded softly (values span 0..255)."""
    n = rng.random((size // 8 + 2, size // 8 + 2))
    # bilinear upsample x8 (wrap-free) -> smooth blobs
    y, x = np.mgrid[0:size, 0:size] / 8.0
    x0, y0 = x.astype(int), y.astype(int)
    fx, fy = x - x0, y - y0
    v = (n[y0, x0] * (1 - fx) * (1 - fy) + n[y0, x0 + 1] * fx * (1 - fy) + n[y0 + 1, x0] * (1 - fx) * fy
         + n[y0 + 1, x0 + 1] * fx * fy)
    a = np.clip((v - (1.0 - coverage)) * 6.0 + 0.5, 0.0, 1.0)
    return np.round(a * 255.0).astype(np.uint8)


# --------------------------------------------------------------------------------------------------
# transforms
# --------------------------------------------------------------------------------------------------
def trs(translate=(0, 0, 0), scale=(1, 1, 1), yaw: float = 0.0) -> np.ndarray:
    """3x4 row-major world-from-mesh (what toVkTransformMatrix produces, RayTracingContext.cpp:28-44)."""
    c, s = math.cos(yaw), math.sin(yaw)
    r = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)
    m = np.zeros((3, 4), dtype=np.float64)
    m[:, :3] = r * np.asarray(scale, dtype=np.float64)[None, :]
    m[:, 3] = translate
    return m.astype(np.float32).reshape(12)


# --------------------------------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------------------------------
def make_street_scene(seed: int, detail: float = 1.0, name: str = "street") -> SceneData:
    """A street canyon: bumpy ground tiles, building blocks with ledges, columns, props, MASK tree canopies,
    BLEND glass panes.  `detail` scales both tessellation and instance counts; detail=1.0 is "bistro-class"
    (~2.8 M triangles, ~2.7 k instances), detail~0.3 is "sponza-class" (~260 k triangles)."""
    rng = np.random.default_rng(seed)
    d = float(detail)

    def n_(x, lo=1):
        return max(lo, int(round(x)))

    mats = [
        Material(ALPHA_OPAQUE, 0.5),                                               # 0 stone
        Material(ALPHA_OPAQUE, 0.5),                                               # 1 plaster
        Material(ALPHA_MASK, 0.5, alpha_blobs(256, rng, 0.55)),                    # 2 leaves
        Material(ALPHA_MASK, 0.35, alpha_blobs(256, rng, 0.45)),                   # 3 fence / sparse leaves
        Material(ALPHA_BLEND, 0.5),                                                # 4 glass
    ]
    tess = math.sqrt(d)
    meshes = [
        grid_mesh(n_(40 * tess, 2), n_(40 * tess, 2), 0, rng, bump=0.004),         # 0 ground tile (unit square)
        box_mesh(n_(22 * tess, 1), 1),                                             # 1 building block
        box_mesh(n_(5 * tess, 1), 0),                                              # 2 ledge / small box
        cylinder_mesh(n_(32 * tess, 6), n_(8 * tess, 1), 0),                       # 3 column
        sphere_mesh(n_(32 * tess, 6), n_(16 * tess, 3), 1),                        # 4 round prop
        foliage_mesh(n_(2800 * d, 16), 2, rng),                                    # 5 canopy (MASK)
        foliage_mesh(n_(1400 * d, 16), 3, rng),                                    # 6 hedge (MASK, lower cutoff)
        grid_mesh(n_(32 * tess, 2), n_(32 * tess, 2), 4),                          # 7 glass pane (BLEND)
        box_mesh(n_(14 * tess, 1), 1),                                             # 8 detailed prop box
        cylinder_mesh(n_(16 * tess, 5), n_(4 * tess, 1), 0),                       # 9 trunk
    ]
    inst_mesh, inst_xf = [], []

    def add(mesh, t, s, yaw=0.0):
        inst_mesh.append(mesh)
        inst_xf.append(trs(t, s, yaw))

    street_len, street_half = 80.0, 5.0
    # ground: tiles covering [-20, 20] x [0, street_len]
    gt = n_(9 * tess, 2)
    for i in range(gt):
        for j in range(gt * 2):
            sx, sz = 40.0 / gt, street_len / (gt * 2)
            add(0, (-20.0 + i * sx, 0.0, j * sz), (sx, 1.0, sz))
    # buildings on both sides, with ledges
    nb = n_(22 * tess, 2)
    for side in (-1.0, 1.0):
        z = 0.0
        for b in range(nb):
            wdt = 3.0 + 4.0 * rng.random()
            hgt = 6.0 + 10.0 * rng.random()
            dep = 6.0 + 3.0 * rng.random()
            x0 = side * street_half if side > 0 else -street_half - dep
            add(1, (x0, 0.0, z), (dep, hgt, wdt))
            nl = n_(14 * tess, 1)
            for l in range(nl):
                ly = 1.0 + (hgt - 1.5) * rng.random()
                lz = z + wdt * rng.random() * 0.9
                lx = (street_half - 0.3) if side > 0 else (-street_half - 0.05)
                add(2, (lx, ly, lz), (0.35, 0.12 + 0.2 * rng.random(), 0.5 + rng.random()))
            z += wdt + 0.2 * rng.random()
            if z > street_len:
                break
    # columns / poles
    for k in range(n_(330 * d, 2)):
        x = (rng.random() * 2 - 1) * (street_half - 0.4)
        add(3, (x, 0.0, rng.random() * street_len), (0.25 + 0.2 * rng.random(), 2.5 + 2.0 * rng.random(), 0.25 + 0.2 * rng.random()))
    # props
    for k in range(n_(420 * d, 2)):
        s = 0.3 + 0.6 * rng.random()
        add(4, ((rng.random() * 2 - 1) * street_half, 0.0, rng.random() * street_len), (s, s, s), rng.random() * 6.28)
    for k in range(n_(310 * d, 2)):
        s = 0.4 + 0.8 * rng.random()
        add(8, ((rng.random() * 2 - 1) * street_half, 0.0, rng.random() * street_len), (s, 0.4 + rng.random(), s), rng.random() * 6.28)
    # trees: trunk + MASK canopy ; hedges
    for k in range(n_(60 * tess, 1)):
        x, z = (rng.random() * 2 - 1) * (street_half - 1.0), rng.random() * street_len
        h = 2.0 + 1.5 * rng.random()
        add(9, (x, 0.0, z), (0.3, h, 0.3))
        s = 2.0 + 1.5 * rng.random()
        add(5, (x, h - 0.3, z), (s, s, s), rng.random() * 6.28)
    for k in range(n_(40 * tess, 1)):
        s = 1.0 + 0.5 * rng.random()
        add(6, ((rng.random() * 2 - 1) * street_half, 0.0, rng.random() * street_len), (s * 1.5, s * 0.8, s), rng.random() * 6.28)
    # glass panes (BLEND), standing quads: rotate the XZ unit square up by mapping z->y via scale trick:
    # use an explicit matrix (x, z) -> (x, y)
    for k in range(n_(70 * tess, 1)):
        x, z = (rng.random() * 2 - 1) * (street_half - 0.5), rng.random() * street_len
        w, h = 1.0 + rng.random(), 1.5 + rng.random()
        m = np.array([[w, 0, 0, x], [0, 0, h, 0.0], [0, 1, 0, z]], dtype=np.float32).reshape(12)
        inst_mesh.append(7)
        inst_xf.append(m)

    scene = SceneData(meshes, mats, np.asarray(inst_mesh, dtype=np.uint32), np.stack(inst_xf).astype(np.float32), name)
    scene.cameras = [((0.5, 1.7, 4.0), (0.0, 1.5, 40.0)), ((-2.0, 2.2, 30.0), (2.0, 1.0, 60.0)),
                     ((3.0, 6.0, 10.0), (-3.0, 0.5, 35.0))]
    return scene


def make_scene_graph(scene: SceneData, seed: int = 1, group_size: int = 6, max_depth: int = 4):
    """A glTF-like node hierarchy over the scene's instances, in DFS pre-order (the order
    Renderer::rebuildRayTracingInstances visits nodes, Renderer.cpp:510-524): nested group nodes with their own
    translations, leaf nodes carrying 1-3 consecutive instances.  With the identity model matrix the hierarchy
    reproduces scene.instance_transform up to float rounding.
    Returns (parent[int32], local[(N,16) float32 column-major], first_instance[uint32], instance_count[uint32])."""
    rng = np.random.default_rng(seed)
    n_inst = len(scene.instance_mesh)
    parents, locals_, firsts, counts = [], [], [], []

    def m4_of(xf12):
        m = np.eye(4, dtype=np.float64)
        m[:3, :4] = np.asarray(xf12, dtype=np.float64).reshape(3, 4)
        return m

    def emit(parent, local4, first, count):
        parents.append(parent)
        locals_.append(np.ascontiguousarray(local4.T, dtype=np.float32).reshape(16))   # column-major
        firsts.append(first)
        counts.append(count)
        return len(parents) - 1

    cursor = 0

    def build(parent, parent_world, lo, hi, depth):
        nonlocal cursor
        if hi - lo <= group_size or depth >= max_depth:
            i = lo
            while i < hi:
                c = int(min(hi - i, rng.integers(1, 4)))
                # instances of one node share the node's transform in the reference; keep the first one's
                world = m4_of(scene.instance_transform[i])
                emit(parent, np.linalg.inv(parent_world) @ world, cursor, c)
                cursor += c
                i += c
            return
        parts = int(rng.integers(2, 5))
        cuts = np.linspace(lo, hi, parts + 1).astype(int)
        for a, b in zip(cuts[:-1], cuts[1:]):
            if a == b:
                continue
            t = np.eye(4)
            t[:3, 3] = rng.uniform(-3.0, 3.0, 3)
            g = emit(parent, t, cursor, 0)
            build(g, parent_world @ t, a, b, depth + 1)

    build(-1, np.eye(4), 0, n_inst, 0)
    assert cursor == n_inst
    return (np.asarray(parents, dtype=np.int32), np.stack(locals_).astype(np.float32),
            np.asarray(firsts, dtype=np.uint32), np.asarray(counts, dtype=np.uint32))


def make_draw_model(n_nodes: int, n_meshes: int, n_materials: int = 64, seed: int = 3, max_depth: int = 8,
                    invalid_fraction: float = 0.0, extent: float = 60.0):
    """A synthetic Model for the GPU-driven indirect build (BASELINE configs[2]): linear nodes (parents first, like
    Model::getLinearNodes), per-node local matrices (rotation about Y, scale, translation), draw slots in the
    collection order of FrameManager::buildSharedDrawSlots (node order, one slot per (node, meshIndex)), MeshDrawInfo
    and object-space mesh AABBs.  `invalid_fraction` of the slots get an out-of-range node or mesh index (the
    `continue` of FrameManager.cpp:140).  Returns a dict of numpy arrays."""
    rng = np.random.default_rng(seed)
    parent = np.full(n_nodes, -1, dtype=np.int32)
    # level by level (parents first): level 0 = roots, every later node hangs under a random node of the level above
    n_roots = max(1, n_nodes // 64)
    bounds_lv = np.unique(np.linspace(n_roots, n_nodes, max(1, max_depth - 1) + 1).astype(np.int64)) if n_nodes > n_roots else []
    prev_lo, prev_hi = 0, min(n_roots, n_nodes)
    for a, b in zip(bounds_lv[:-1], bounds_lv[1:]):
        parent[a:b] = rng.integers(prev_lo, prev_hi, b - a)
        prev_lo, prev_hi = int(a), int(b)
    ang = rng.uniform(-np.pi, np.pi, n_nodes)
    sc = rng.uniform(0.6, 1.5, n_nodes)
    tr = rng.uniform(-extent, extent, (n_nodes, 3)) * np.where(parent[:, None] < 0, 1.0, 0.15)
    local = np.zeros((n_nodes, 4, 4), dtype=np.float64)      # [col][row]
    local[:, 0, 0] = np.cos(ang) * sc; local[:, 0, 2] = -np.sin(ang) * sc
    local[:, 1, 1] = sc
    local[:, 2, 0] = np.sin(ang) * sc; local[:, 2, 2] = np.cos(ang) * sc
    local[:, 3, :3] = tr; local[:, 3, 3] = 1.0
    local = local.reshape(n_nodes, 16).astype(np.float32)
    # meshes: packed into one global vertex/index buffer
    idx_count = (rng.integers(4, 4000, n_meshes) * 3).astype(np.uint32)
    vtx_count = rng.integers(8, 3000, n_meshes).astype(np.uint32)
    infos = np.zeros(n_meshes, dtype=[("vertex_offset", "<u4"), ("first_index", "<u4"), ("index_count", "<u4")])
    infos["index_count"] = idx_count
    infos["first_index"] = np.concatenate([[0], np.cumsum(idx_count[:-1], dtype=np.uint64)]).astype(np.uint32)
    infos["vertex_offset"] = np.concatenate([[0], np.cumsum(vtx_count[:-1], dtype=np.uint64)]).astype(np.uint32)
    half = rng.uniform(0.1, 2.5, (n_meshes, 3))
    cen = rng.uniform(-1.0, 1.0, (n_meshes, 3))
    bounds = np.concatenate([cen - half, cen + half], axis=1).astype(np.float32)
    mesh_mat = rng.integers(0, n_materials, n_meshes).astype(np.uint32)
    mat_ds = rng.random(n_materials) < 0.25
    # slots: ~1 per node on average, node order
    per_node = rng.choice([0, 1, 1, 1, 2, 3], n_nodes)
    node_of = np.repeat(np.arange(n_nodes, dtype=np.uint32), per_node)
    mesh_of = rng.integers(0, n_meshes, len(node_of)).astype(np.uint32)
    slots = np.zeros(len(node_of), dtype=[("node_linear_index", "<u4"), ("mesh_index", "<u4"), ("mat_index", "<u4"),
                                          ("double_sided", "<u4")])
    slots["node_linear_index"] = node_of
    slots["mesh_index"] = mesh_of
    slots["mat_index"] = mesh_mat[mesh_of]
    slots["double_sided"] = mat_ds[mesh_mat[mesh_of]].astype(np.uint32)
    if invalid_fraction > 0 and len(slots):
        bad = rng.random(len(slots)) < invalid_fraction
        which = rng.random(len(slots)) < 0.5
        slots["node_linear_index"][bad & which] = n_nodes + 7
        slots["mesh_index"][bad & ~which] = 0xFFFFFFFF
    return {"node_parent": parent, "node_local": local, "slots": slots, "mesh_infos": infos, "mesh_bounds": bounds}


def make_sky_hdr(width: int = 1024, height: int = 512, seed: int = 11, sun_peak: float = 3.0e4) -> np.ndarray:
    """A synthetic HDR equirect (RGBA32F [H, W, 4], row 0 = zenith) standing in for the reference's .hdr sky: horizon
    gradient, ground, a small very bright sun, soft clouds.  Dynamic range like an outdoor RGBE image, kept below the
    fp16 maximum so prefilter mip 0 stays finite."""
    rng = np.random.default_rng(seed)
    v = (np.arange(height, dtype=np.float64) + 0.5) / height
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    lat = (0.5 - v)[:, None] * np.pi                    # +pi/2 at row 0
    lon = (u[None, :] - 0.5) * 2.0 * np.pi
    d = np.stack([np.cos(lat) * np.cos(lon), np.sin(lat) * np.ones_like(lon), np.cos(lat) * np.sin(lon)], axis=-1)
    up = np.clip(d[..., 1], 0.0, 1.0)
    sky = np.stack([0.25 + 0.35 * (1 - up) ** 3, 0.45 + 0.35 * (1 - up) ** 3, 0.9 + 0.1 * (1 - up) ** 3], axis=-1)
    ground = np.array([0.18, 0.15, 0.12])
    img = np.where(d[..., 1:2] >= 0.0, sky, ground * (0.6 + 0.4 * np.clip(-d[..., 1:2] * 4, 0, 1)))
    sun = np.array([0.35, 0.62, -0.70]); sun /= np.linalg.norm(sun)
    cosang = np.clip(d @ sun, -1.0, 1.0)
    img = img + sun_peak * np.exp(-((1.0 - cosang) / 2.0e-4)) [..., None] * np.array([1.0, 0.93, 0.82])
    img = img + 4.0 * np.exp(-((1.0 - cosang) / 2.0e-2))[..., None] * np.array([1.0, 0.9, 0.7])
    clouds = np.zeros((height, width))
    for k in range(1, 6):
        ph = rng.uniform(0, 2 * np.pi, 3)
        clouds += np.sin(k * 3 * lon + ph[0]) * np.sin(k * 5 * lat + ph[1]) / k
    img = img + (np.clip(clouds, 0, None) * up)[..., None] * 0.6
    out = np.ones((height, width, 4), dtype=np.float32)
    out[..., :3] = img.astype(np.float32)
    return out


def scene_from_gltf_arrays(positions, normals, uvs, indices, node_local_col_major, name="gltf",
                           scale: float = 1.0, translate=(0.0, 0.0, 0.0)) -> SceneData:
    """One-mesh, one-node glTF (decoded arrays, e.g. tests/golden/viking_room_mesh.npz) as a SceneData: the instance
    transform is modelMatrix * node local matrix (Renderer::rebuildRayTracingInstances, Renderer.cpp:503-527) with
    modelMatrix = translate * uniform scale."""
    v = np.zeros((len(positions), VERTEX_FLOATS), dtype=np.float32)
    v[:, 0:3] = positions
    v[:, 3:6] = normals
    v[:, 6:9] = 1.0
    v[:, 9:11] = uvs
    mesh = Mesh(v, np.ascontiguousarray(indices, dtype=np.uint32), 0, keep_normals=True)
    local = np.asarray(node_local_col_major, dtype=np.float64).reshape(4, 4).T
    model = np.eye(4)
    model[:3, :3] *= scale
    model[:3, 3] = translate
    world = (model @ local).astype(np.float32)
    sc = SceneData([mesh], [Material()], np.zeros(1, dtype=np.uint32), world[:3, :4].reshape(1, 12).copy(), name)
    w = positions @ world[:3, :3].T + world[:3, 3]
    lo, hi = w.min(axis=0), w.max(axis=0)
    c, r = (lo + hi) * 0.5, float(np.linalg.norm(hi - lo)) * 0.5
    sc.cameras = [(tuple(c + np.array([r * 1.1, r * 0.7, r * 1.2])), tuple(c)),
                  (tuple(c + np.array([-r * 0.9, r * 0.5, r * 1.3])), tuple(c)),
                  (tuple(c + np.array([r * 0.2, r * 1.4, -r * 0.6])), tuple(c))]
    return sc


def make_bistro_class(seed: int = 0xB1572026) -> SceneData:
    return make_street_scene(seed, 1.0, "bistro-class")


def make_sponza_class(seed: int = 0x5907A) -> SceneData:
    return make_street_scene(seed, 0.26, "sponza-class")     # ~257 k triangles (Sponza glTF: ~262 k)


def make_test_scene(seed: int = 7, detail: float = 0.004) -> SceneData:
    """A few thousand triangles with every material class present; used by parity tests."""
    return make_street_scene(seed, detail, "test")


def make_material_matrix_scene() -> SceneData:
    """SURVEY 8c pin 6: opaque / MASK (alpha above and below cutoff) / BLEND quads in front of an opaque wall.
    Quads are unit squares in the XY plane facing -Z, at z = 1 (front) and z = 2 (wall), four columns in x."""
    def quad(material, uv_lo, uv_hi):
        pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], dtype=np.float32)
        uv = np.array([[uv_lo, uv_lo], [uv_hi, uv_lo], [uv_lo, uv_hi], [uv_hi, uv_hi]], dtype=np.float32)
        return Mesh(_make_vertices(pos, uv), np.array([0, 2, 1, 1, 2, 3], dtype=np.uint32), material)

    tex = np.zeros((8, 8), dtype=np.uint8)
    tex[:, 4:] = 255           # left half alpha 0, right half alpha 1
    mats = [Material(ALPHA_OPAQUE), Material(ALPHA_MASK, 0.5, tex), Material(ALPHA_BLEND)]
    meshes = [quad(0, 0.0, 1.0),              # 0 opaque
              quad(1, 0.70, 0.95),            # 1 MASK, uv in the alpha=1 half  -> occludes
              quad(1, 0.05, 0.30),            # 2 MASK, uv in the alpha=0 half  -> transparent
              quad(2, 0.0, 1.0),              # 3 BLEND                          -> never occludes AO
              quad(0, 0.0, 1.0)]              # 4 wall piece
    inst_mesh, inst_xf = [], []
    for col in range(4):
        inst_mesh.append(col); inst_xf.append(trs((2.0 * col, 0.0, 1.0)))
        inst_mesh.append(4); inst_xf.append(trs((2.0 * col, 0.0, 2.0)))
    return SceneData(meshes, mats, np.asarray(inst_mesh, dtype=np.uint32), np.stack(inst_xf), "material-matrix")


# --------------------------------------------------------------------------------------------------
# camera + UBO (Camera.cpp:142-152, GlmConfig.h:8-14, FrameManager.cpp:440-511, VulkanTypes.h:47-62)
# --------------------------------------------------------------------------------------------------
UBO_FLOATS = 120
UBO_VIEW, UBO_PROJ = 16, 32
UBO_DIR_LIGHT_DIR, UBO_DIR_LIGHT_COLOR, UBO_DIR_LIGHT_PARAMS = 48, 52, 56
UBO_LIGHT_POSITIONS, UBO_LIGHT_COLORS = 60, 72
UBO_CAM_POS, UBO_PARAMS, UBO_IBL_PARAMS = 84, 88, 92
UBO_RTAO_PARAMS0, UBO_RTAO_PARAMS1, UBO_PREV_VIEW_PROJ = 96, 100, 104


def look_at_rh(eye, center, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """glm::lookAt (RH); returned as a 4x4 array indexed [row, col]."""
    eye, center, up = (np.asarray(v, dtype=np.float64) for v in (eye, center, up))
    f = center - eye; f /= np.linalg.norm(f)
    s = np.cross(f, up); s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -s.dot(eye), -u.dot(eye), f.dot(eye)
    return m.astype(np.float32)


def perspective_rh_zo_yflip(fovy_deg: float, aspect: float, near: float, far: float) -> np.ndarray:
    """glm::perspective under GLM_FORCE_DEPTH_ZERO_TO_ONE (RH, depth 0..1) with proj[1][1] *= -1
    (Camera.cpp:147-152).  [row, col]."""
    t = math.tan(math.radians(fovy_deg) / 2.0)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = -1.0 / t
    m[2, 2] = far / (near - far)
    m[3, 2] = -1.0
    m[2, 3] = -(far * near) / (far - near)
    return m.astype(np.float32)


def col_major(m: np.ndarray) -> np.ndarray:
    """[row, col] 4x4 -> 16 floats in GLM/GLSL column-major order."""
    return np.ascontiguousarray(m.T, dtype=np.float32).reshape(16)


class FrameState:
    """The per-frame UBO bookkeeping of FrameManager (FrameManager.cpp:440-511): prevViewProj = last frame's
    proj*view (identity on the first frame, :28,57), rtaoParams1.w = uniformFrameIndex incremented after the
    copy (:463,510).  AppConfig.h:88-99 defaults."""

    def __init__(self, width: int, height: int, ray_count: int = 8, radius: float = 1.2, bias: float = 0.0005,
                 strength: float = 1.0, temporal_alpha: float = 0.8, disocclusion: float = 0.35,
                 sun_radius: float = 0.01, sun_samples: int = 8, point_lights: int = 0, debug_view: int = 0):
        self.width, self.height = width, height
        self.ray_count, self.radius, self.bias = ray_count, radius, bias
        self.strength, self.temporal_alpha, self.disocclusion = strength, temporal_alpha, disocclusion
        self.sun_radius, self.sun_samples, self.point_lights = sun_radius, sun_samples, point_lights
        self.debug_view = debug_view
        self.reset()

    def reset(self):                               # FrameManager.cpp:28,57-58
        self.last_view_proj = np.eye(4, dtype=np.float32)
        self.uniform_frame_index = 0

    def update(self, eye, target, fov_deg: float = 45.0, near: float = 0.1, far: float = 1000.0) -> np.ndarray:
        """FrameManager::updateUniformBuffer: returns the 120-float UBO for this frame and advances state."""
        view = look_at_rh(eye, target)
        proj = perspective_rh_zo_yflip(fov_deg, self.width / float(self.height), near, far)   # :452
        ubo = np.zeros(UBO_FLOATS, dtype=np.float32)
        ubo[0:16] = col_major(np.eye(4, dtype=np.float32))
        ubo[UBO_VIEW:UBO_VIEW + 16] = col_major(view)
        ubo[UBO_PROJ:UBO_PROJ + 16] = col_major(proj)
        ubo[UBO_PREV_VIEW_PROJ:UBO_PREV_VIEW_PROJ + 16] = col_major(self.last_view_proj)          # :453
        ubo[UBO_DIR_LIGHT_DIR:UBO_DIR_LIGHT_DIR + 4] = (-0.3, -0.8, -0.5, 1.0)                   # :470
        ubo[UBO_DIR_LIGHT_COLOR:UBO_DIR_LIGHT_COLOR + 4] = (0.92 * 3, 0.82 * 3, 0.72 * 3, 1.0)   # :472-473
        ubo[UBO_DIR_LIGHT_PARAMS:UBO_DIR_LIGHT_PARAMS + 4] = (self.sun_radius, float(self.sun_samples), 0, 0)  # :475
        ubo[UBO_LIGHT_POSITIONS:UBO_LIGHT_POSITIONS + 12] = (2, 4, 2, 1, -2, 2, -2, 1, 0, 3, 2, 1)  # :477-483
        ubo[UBO_LIGHT_COLORS:UBO_LIGHT_COLORS + 12] = (3, 2, 1, 1, 2, 6, 0, 1, 5, 10, 4, 1)
        ubo[UBO_CAM_POS:UBO_CAM_POS + 4] = (eye[0], eye[1], eye[2], 1.0)                          # :456
        ubo[UBO_PARAMS:UBO_PARAMS + 4] = (0.6, 2.2, 0.03, float(self.point_lights))              # :486-490
        ubo[UBO_IBL_PARAMS:UBO_IBL_PARAMS + 4] = (1.0, 1.0, 1.0, float(self.debug_view))         # :495-499
        ubo[UBO_RTAO_PARAMS0:UBO_RTAO_PARAMS0 + 4] = (1.0, float(self.ray_count), self.radius, self.bias)  # :501-505
        ubo[UBO_RTAO_PARAMS1:UBO_RTAO_PARAMS1 + 4] = (self.strength, self.temporal_alpha, self.disocclusion,
                                                      float(self.uniform_frame_index))           # :506-510
        # currViewProj = proj * view in float32 like glm (column-major product, float accumulate)
        self.last_view_proj = (proj.astype(np.float32) @ view.astype(np.float32)).astype(np.float32)  # :458-459
        self.uniform_frame_index += 1                                                             # :463
        return ubo

    @staticmethod
    def sun_dir_to_light(ubo: np.ndarray) -> np.ndarray:
        """L = -safeNormalize(directionalLightDir.xyz) (pbr.frag:379-380)."""
        d = ubo[UBO_DIR_LIGHT_DIR:UBO_DIR_LIGHT_DIR + 3].astype(np.float32)
        n = np.float32(np.sqrt(np.float32((d * d).sum())))
        return (-(d / n)).astype(np.float32)


def camera_path(scene: SceneData, frames: int, step: float = 0.05):
    """A straight dolly along the first suggested camera, `step` metres per frame (1-10 cm/frame)."""
    eye, tgt = (np.asarray(v, dtype=np.float64) for v in scene.cameras[0])
    fwd = tgt - eye
    fwd /= np.linalg.norm(fwd)
    return [((eye + fwd * step * i).astype(np.float32), (tgt + fwd * step * i).astype(np.float32)) for i in range(frames)]
