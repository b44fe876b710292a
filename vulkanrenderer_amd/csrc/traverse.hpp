// traverse.hpp -- ray queries over the wide LBVH (replaces rayQueryEXT + the driver's TLAS/BLAS walk).
//
// Data layout in HBM (16-byte aligned, read as dwordx4 quarters):
//   wide node (64 B, up to 4 children, boxes quantised to 8 bits per plane in the frame of their union --
//   traversal is bound by L1/TA bytes per visit, not by arithmetic, so 64 B beats 128 B of float planes):
//       q0 origin.xyz | scale_z                  (scale_a = a power of two, stored as a float: no decode in the node step)
//       q1 qminx[4] qminy[4] qminz[4] qmaxx[4]   (one byte per child, plane = fma(q, scale, origin))
//       q2 qmaxy[4] qmaxz[4] scale_x scale_y
//       q3 child[4]
//       child >= 0            : wide node index
//       child <  0            : leaf: bit 31 set, bits 30..1 = first triangle, bit 0 = count - 1 (leaves hold 1 or 2
//                               triangles).  The streaming kernels turn it into a queue item with ONE add:
//                               item = ray << (32 - RB) | first << 1 | (count - 1)
//       unused slot           : child = 0x7fffffff, planes stored inverted (qmin = 255, qmax = 0): fails every slab test
//   tri (48 B, Morton order so a leaf's triangles are contiguous):
//       q0 = v0.xyz (mesh | class<<28) | q1 = e1.xyz prim | q2 = e2.xyz flat_id
//       class: 0 = always an occluder for AO, 1 = never (BLEND), 2 = alpha-test (MASK)
// Traversal comes in two forms.  The generic traverse<>() below keeps one ray per lane ("while-while": lanes walk
// internal nodes until each holds a leaf, then the wave tests leaves together; near-to-far order from a 5-exchange
// network on packed (tnear bits | slot) keys; deferred children on a short LDS stack [depth][lane] that spills to
// per-lane scratch beyond VKR_LDS_STACK entries, entries carrying tnear so stale ones are culled at pop time).  It
// serves the G-buffer producer and the fallback kernels.  The hot kernels (k_trace_q, k_shadow_q, k_shadow_sun_pkt)
// use wave-wide work queues instead; their shared pieces are at the end of this file.
#pragma once
#include "device_math.hpp"

namespace vkr {

constexpr int VKR_LDS_STACK = 12;    // entries per lane kept in LDS (2 dwords each)
constexpr int VKR_MAX_STACK = 96;    // hard cap; the host checks 3*levels+1 against it after every build
constexpr int VKR_TRACE_BLOCK = 256;
constexpr int VKR_GRAPH_MAX_DEPTH = 64;   // scene-graph depth accepted by vkr_hierarchy_upload
#ifndef VKR_LEAF_MAX_N
#define VKR_LEAF_MAX_N 2
#endif
constexpr int VKR_LEAF_MAX = VKR_LEAF_MAX_N;      // triangles per leaf; measured 1/2/3/4/6: 2 is best for K1, 1-2 for the sun packets
static_assert(VKR_LEAF_MAX >= 1 && VKR_LEAF_MAX <= 2, "the leaf reference has one bit for the triangle count");
constexpr uint32_t VKR_CHILD_NONE = 0x7fffffffu;
// leaf reference (see the layout above)
__host__ __device__ inline uint32_t leaf_ref_make(uint32_t first, uint32_t count) { return 0x80000000u | (first << 1) | (count - 1u); }
__host__ __device__ inline uint32_t leaf_ref_first(uint32_t ref) { return (ref & 0x7fffffffu) >> 1; }
__host__ __device__ inline uint32_t leaf_ref_count(uint32_t ref) { return (ref & 1u) + 1u; }
#ifndef VKR_LEAF_VOTE
#define VKR_LEAF_VOTE 1   // K1 runs a triangle step when n_leaf * VKR_LEAF_VOTE >= n_node * VKR_LEAF_VOTE_DEN
#endif
#ifndef VKR_REFILL_BATCH
#define VKR_REFILL_BATCH 8   // K1 services finished/idle lanes once this many have piled up (or none has work)
#endif
#ifndef VKR_LEAF_VOTE_DEN
#define VKR_LEAF_VOTE_DEN 1
#endif

struct AlphaTexDev { uint32_t offset, w, h, pad; };

struct SceneDev {
    const float4* nodes;
    const float4* tris;
    uint32_t tri_count;
    uint32_t node_count;
    // MASK path (rtao_trace_half.comp:141-175)
    const uint2* lut;           // InstanceLUTEntry {materialID, indexBufferOffset}
    const uint32_t* indices;    // merged, global vertex ids
    const float2* uvs;          // merged
    const float4* material_params;
    const AlphaTexDev* alpha_tex;
    const uint8_t* alpha_data;
    const float4* tri_alpha;    // 3 x float4 per triangle, same order as tris: {uv0, uv1}, {uv2, cutoff, -}, {texture offset, w, h, -}; written for MASK triangles only
    // G-buffer producer (depth_prepass.vert:47-48): vertex normals and the instance a triangle came from
    const float* normals;            // merged xyz
    const uint32_t* mesh_has_normals;
    const uint32_t* inst_first_tri;  // exclusive prefix of triangles per instance (+1 entry)
    const uint32_t* tri_inst;        // per triangle (sorted order): its instance
#ifdef VKR_PKT_STATS
    unsigned long long* pkt_stats;   // development counters of traverse_packet: packets, node visits, triangle tests
#endif
    const float* inst_xf;            // 12 floats per instance
    uint32_t instance_count;
};

struct HitDev { float t; uint32_t tri; float u, v; };

enum { Q_AO = 0, Q_SHADOW = 1, Q_PRIMARY = 2, Q_PRIMARY_MS = 3 };

struct TraceCounters { uint32_t nodes, tris, wave_node_iters, wave_tri_iters; };

// baseColor alpha: LOD 0, bilinear, REPEAT (rtao_trace_half.comp:170)
VKR_DI float sample_alpha_tex(const SceneDev& sc, AlphaTexDev t, V2 uv)
{
    if (t.w == 0 || t.h == 0) return 1.0f;
    float x = uv.x * (float)t.w - 0.5f;
    float y = uv.y * (float)t.h - 0.5f;
    if (!(__builtin_fabsf(x) < 1.0e9f) || !(__builtin_fabsf(y) < 1.0e9f)) return 1.0f;
    float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    float ax = x - fx, ay = y - fy;
    int W = (int)t.w, H = (int)t.h;
    int ix0 = (int)fx % W; if (ix0 < 0) ix0 += W;
    int iy0 = (int)fy % H; if (iy0 < 0) iy0 += H;
    int ix1 = ix0 + 1; if (ix1 == W) ix1 = 0;
    int iy1 = iy0 + 1; if (iy1 == H) iy1 = 0;
    const uint8_t* a = sc.alpha_data + t.offset;
    const float s = 1.0f / 255.0f;
    float t00 = (float)a[(size_t)iy0 * W + ix0] * s, t10 = (float)a[(size_t)iy0 * W + ix1] * s;
    float t01 = (float)a[(size_t)iy1 * W + ix0] * s, t11 = (float)a[(size_t)iy1 * W + ix1] * s;
    float top = t00 + ax * (t10 - t00);
    float bot = t01 + ax * (t11 - t01);
    return top + ay * (bot - top);
}

VKR_DI float sample_alpha(const SceneDev& sc, uint32_t mat, V2 uv) { return sample_alpha_tex(sc, sc.alpha_tex[mat], uv); }

// MASK candidate: barycentric UV -> alpha >= cutoff (rtao_trace_half.comp:163-173)
VKR_DI bool alpha_test(const SceneDev& sc, uint32_t mesh, uint32_t prim, float bu, float bv)
{
    uint2 lut = sc.lut[mesh];
    float cutoff = sc.material_params[lut.x].x;
    float w0 = (1.0f - bu) - bv;
    uint32_t i0 = sc.indices[lut.y + prim * 3u + 0u];
    uint32_t i1 = sc.indices[lut.y + prim * 3u + 1u];
    uint32_t i2 = sc.indices[lut.y + prim * 3u + 2u];
    float2 a = sc.uvs[i0], b = sc.uvs[i1], c = sc.uvs[i2];
    V2 uv = V2{(a.x * w0 + b.x * bu) + c.x * bv, (a.y * w0 + b.y * bu) + c.y * bv};
    return sample_alpha(sc, lut.x, uv) >= cutoff;
}

// The same test from the triangle's alpha record (bvh_build.hip k_alpha_records): the three UVs, the material's cutoff and
// its texture descriptor sit next to the triangle instead of behind lut -> indices -> uvs and lut -> material tables, so the
// chain of dependent loads is {triangle, record} -> texels.  The tables only change in vkr_scene_upload, which a build follows.
VKR_DI bool alpha_test_rec(const SceneDev& sc, float4 r0, float4 r1, float4 r2, float bu, float bv)
{
    float w0 = (1.0f - bu) - bv;
    V2 uv = V2{(r0.x * w0 + r0.z * bu) + r1.x * bv, (r0.y * w0 + r0.w * bu) + r1.y * bv};
    return sample_alpha_tex(sc, AlphaTexDev{f2u(r2.x), f2u(r2.y), f2u(r2.z), 0u}, uv) >= r1.z;
}

// One decoded wide node: 24 planes + 4 child refs.
struct NodeDec {
    float mnx[4], mny[4], mnz[4], mxx[4], mxy[4], mxz[4];
    uint32_t ref[4];
};

// Wave priority in the queue kernels (k_trace_q, k_shadow_sun_pkt, k_shadow_q): raised (s_setprio) while a step pops its
// items and issues their loads, and again from the end of its arithmetic through its appends (K1: also through the alpha
// steps, the visibility sums, the work-counter round trip and the pixel-record loads); 0 during the slab / triangle
// arithmetic and ray generation.  A wave that is about to put loads in flight goes ahead of waves that only compute, so the
// CU keeps more of them in flight: -3 % on K1 (C3).  -DVKR_WAVE_PRIO=0 turns it off.
#ifndef VKR_WAVE_PRIO
#define VKR_WAVE_PRIO 1
#endif
#if VKR_WAVE_PRIO
#define VKR_PRIO_HI __builtin_amdgcn_s_setprio(VKR_WAVE_PRIO);
#define VKR_PRIO_LO __builtin_amdgcn_s_setprio(0);
#else
#define VKR_PRIO_HI
#define VKR_PRIO_LO
#endif

// v_max3 / v_min3 / v_max / v_min as they are: __builtin_fmaxf makes hipcc quiet its operands first (v_max x, x), eight
// extra instructions per node step.  The slab values are never signalling NaNs.
VKR_DI float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
VKR_DI float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
VKR_DI float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
VKR_DI float vmin2(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

typedef float f2v __attribute__((ext_vector_type(2)));
VKR_DI f2v pk_fma(f2v a, float b, float c) { return __builtin_elementwise_fma(a, f2v{b, b}, f2v{c, c}); }   // v_pk_fma_f32

VKR_DI float ubyte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }   // v_cvt_f32_ubyteK

VKR_DI void decode_node(const float4* np, NodeDec& n)
{
    const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
    const float sx = q2.z, sy = q2.w, sz = q0.w;
    const uint32_t wminx = f2u(q1.x), wminy = f2u(q1.y), wminz = f2u(q1.z), wmaxx = f2u(q1.w), wmaxy = f2u(q2.x),
                   wmaxz = f2u(q2.y);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        n.mnx[k] = fma_(ubyte_f(wminx, k), sx, q0.x); n.mxx[k] = fma_(ubyte_f(wmaxx, k), sx, q0.x);
        n.mny[k] = fma_(ubyte_f(wminy, k), sy, q0.y); n.mxy[k] = fma_(ubyte_f(wmaxy, k), sy, q0.y);
        n.mnz[k] = fma_(ubyte_f(wminz, k), sz, q0.z); n.mxz[k] = fma_(ubyte_f(wmaxz, k), sz, q0.z);
    }
    n.ref[0] = f2u(q3.x); n.ref[1] = f2u(q3.y); n.ref[2] = f2u(q3.z); n.ref[3] = f2u(q3.w);
}

#ifndef VKR_PKT_SCALAR_LOADS
#define VKR_PKT_SCALAR_LOADS 1   // traverse_packet fetches its (wave-uniform) nodes and triangles with scalar loads
#endif
typedef float vkr_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) vkr_f4* vkr_cf4p;   // constant address space: uniform addresses become s_load
VKR_DI float4 ldk(vkr_cf4p p, int i) { const vkr_f4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
VKR_DI void decode_node_k(vkr_cf4p np, NodeDec& n)                   // decode_node() on a node fetched through it
{
    const float4 q0 = ldk(np, 0), q1 = ldk(np, 1), q2 = ldk(np, 2), q3 = ldk(np, 3);
    const float sx = q2.z, sy = q2.w, sz = q0.w;
    const uint32_t wminx = f2u(q1.x), wminy = f2u(q1.y), wminz = f2u(q1.z), wmaxx = f2u(q1.w), wmaxy = f2u(q2.x),
                   wmaxz = f2u(q2.y);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        n.mnx[k] = fma_(ubyte_f(wminx, k), sx, q0.x); n.mxx[k] = fma_(ubyte_f(wmaxx, k), sx, q0.x);
        n.mny[k] = fma_(ubyte_f(wminy, k), sy, q0.y); n.mxy[k] = fma_(ubyte_f(wmaxy, k), sy, q0.y);
        n.mnz[k] = fma_(ubyte_f(wminz, k), sz, q0.z); n.mxz[k] = fma_(ubyte_f(wmaxz, k), sz, q0.z);
    }
    n.ref[0] = f2u(q3.x); n.ref[1] = f2u(q3.y); n.ref[2] = f2u(q3.z); n.ref[3] = f2u(q3.w);
}

// Slab test of one child box against [tmin, tbest]; returns the packed sort key: tnear bits with the child
// slot in the two low bits, or 0xffffffff on a miss.  Padded so rounding can only add visits.
VKR_DI uint32_t box_key(float minx, float miny, float minz, float maxx, float maxy, float maxz, V3 o, V3 id,
                        float tmin, float tbest_pad, uint32_t slot)
{
    float t0 = (minx - o.x) * id.x, t1 = (maxx - o.x) * id.x;
    float tn = __builtin_fminf(t0, t1), tf = __builtin_fmaxf(t0, t1);
    t0 = (miny - o.y) * id.y; t1 = (maxy - o.y) * id.y;
    tn = __builtin_fmaxf(tn, __builtin_fminf(t0, t1)); tf = __builtin_fminf(tf, __builtin_fmaxf(t0, t1));
    t0 = (minz - o.z) * id.z; t1 = (maxz - o.z) * id.z;
    tn = __builtin_fmaxf(tn, __builtin_fminf(t0, t1)); tf = __builtin_fminf(tf, __builtin_fmaxf(t0, t1));
    tn = __builtin_fmaxf(tn, tmin);
    tf = __builtin_fminf(tf * 1.000001f, tbest_pad);
    // tn >= tmin >= 0: its bit pattern orders like an unsigned integer; clearing two bits only lowers it
    uint32_t key = (f2u(tn * 0.999999f) & ~3u) | slot;
    return (tn * 0.999999f <= tf) ? key : 0xffffffffu;
}

VKR_DI void cswap(uint32_t& a, uint32_t& b)
{
    uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
}

// MODE Q_AO      : closest committed hit, candidates filtered by class / alpha test (queryAoClosestHit)
// MODE Q_SHADOW  : any hit, every triangle occludes (gl_RayFlagsOpaqueEXT|TerminateOnFirstHit, pbr.frag:199-215)
// MODE Q_PRIMARY : as Q_AO, ties in t resolved towards the lower flat triangle id (G-buffer producer)
// MODE Q_PRIMARY_MS : Q_PRIMARY for one MSAA sample: the MASK discard of a candidate is decided with the barycentrics of
//                  the pixel-centre ray (o, dc) on the triangle's plane -- the fragment shader runs once per pixel
// `stack` points at this lane's column of the block's LDS stack: entry k = {stack[(2k)*B], stack[(2k+1)*B]}.
template <int MODE, bool STATS, int B = VKR_TRACE_BLOCK>
__device__ __forceinline__ bool traverse(const SceneDev& sc, V3 o, V3 d, float tmin, float tmax, HitDev* hit,
                                         uint32_t* stack, TraceCounters* cnt, V3 dc = V3{0.0f, 0.0f, 1.0f})
{
    float tbest = tmax;
    uint32_t best_flat = 0xffffffffu, best_tri = 0xffffffffu;
    float bu = 0.0f, bv = 0.0f;
    bool found = false;
    if (sc.tri_count == 0) return false;

    const V3 id = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    uint2 spill[VKR_MAX_STACK - VKR_LDS_STACK];
    int sp = 0;
    int cur = 0;  // root
    auto push = [&](uint32_t key, uint32_t ref) {
        if (sp < VKR_LDS_STACK) { stack[(2 * sp) * B] = key; stack[(2 * sp + 1) * B] = ref; }
        else spill[sp - VKR_LDS_STACK] = make_uint2(key, ref);
        ++sp;
    };
    // pop the next entry still in front of the closest hit; false when the stack is exhausted
    auto pop = [&]() -> bool {
        while (sp > 0) {
            --sp;
            uint32_t key, ref;
            if (sp < VKR_LDS_STACK) { key = stack[(2 * sp) * B]; ref = stack[(2 * sp + 1) * B]; }
            else { uint2 e = spill[sp - VKR_LDS_STACK]; key = e.x; ref = e.y; }
            if (u2f(key & ~3u) <= tbest * 1.000001f) { cur = (int)ref; return true; }
        }
        return false;
    };

    bool alive = true;
    while (alive) {
        // ---- internal nodes: descend until this lane holds a leaf (or runs out of work)
        while (cur >= 0) {
            NodeDec nd;
            decode_node(sc.nodes + (size_t)cur * 4, nd);
            if (STATS) { cnt->nodes++; if ((int)(__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63)) cnt->wave_node_iters++; }
            const float tb = tbest * 1.000001f;
            uint32_t k0 = box_key(nd.mnx[0], nd.mny[0], nd.mnz[0], nd.mxx[0], nd.mxy[0], nd.mxz[0], o, id, tmin, tb, 0u);
            uint32_t k1 = box_key(nd.mnx[1], nd.mny[1], nd.mnz[1], nd.mxx[1], nd.mxy[1], nd.mxz[1], o, id, tmin, tb, 1u);
            uint32_t k2 = box_key(nd.mnx[2], nd.mny[2], nd.mnz[2], nd.mxx[2], nd.mxy[2], nd.mxz[2], o, id, tmin, tb, 2u);
            uint32_t k3 = box_key(nd.mnx[3], nd.mny[3], nd.mnz[3], nd.mxx[3], nd.mxy[3], nd.mxz[3], o, id, tmin, tb, 3u);
            if (nd.ref[1] == VKR_CHILD_NONE) k1 = 0xffffffffu;     // used slots are a prefix
            if (nd.ref[2] == VKR_CHILD_NONE) k2 = 0xffffffffu;
            if (nd.ref[3] == VKR_CHILD_NONE) k3 = 0xffffffffu;
            cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);  // ascending
            auto ref_of = [&](uint32_t key) -> uint32_t {
                uint32_t s = key & 3u;
                return s == 0u ? nd.ref[0] : (s == 1u ? nd.ref[1] : (s == 2u ? nd.ref[2] : nd.ref[3]));
            };
            if (k0 == 0xffffffffu) {           // nothing hit
                if (!pop()) { alive = false; break; }
                continue;
            }
            if (k3 != 0xffffffffu) push(k3, ref_of(k3));
            if (k2 != 0xffffffffu) push(k2, ref_of(k2));
            if (k1 != 0xffffffffu) push(k1, ref_of(k1));
            cur = (int)ref_of(k0);
        }
        if (!alive) break;
        // ---- leaf
        {
            uint32_t first = leaf_ref_first((uint32_t)cur), count = leaf_ref_count((uint32_t)cur);
            bool stop = false;
            for (uint32_t k = 0; k < count; ++k) {
                uint32_t ti = first + k;
                const float4* tp = sc.tris + (size_t)ti * 3;
                float4 a = tp[0], b = tp[1], c = tp[2];
                if (STATS) { cnt->tris++; if ((int)(__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63)) cnt->wave_tri_iters++; }
                V3 v0 = V3{a.x, a.y, a.z}, e1 = V3{b.x, b.y, b.z}, e2 = V3{c.x, c.y, c.z};
                V3 pvec = cross3(d, e2);
                float det = dot3(e1, pvec);
                if (det == 0.0f) continue;
                float inv = 1.0f / det;
                V3 tvec = o - v0;
                float u = dot3(tvec, pvec) * inv;
                if (!(u >= 0.0f && u <= 1.0f)) continue;
                V3 qvec = cross3(tvec, e1);
                float v = dot3(d, qvec) * inv;
                if (!(v >= 0.0f && u + v <= 1.0f)) continue;
                float t = dot3(e2, qvec) * inv;
                if (!(t > tmin)) continue;
                uint32_t flat = f2u(c.w);
                if (MODE == Q_PRIMARY || MODE == Q_PRIMARY_MS) {
                    if (!(t < tbest || (t == tbest && flat < best_flat))) continue;
                } else {
                    if (!(t < tbest)) continue;
                }
                if (MODE != Q_SHADOW) {
                    uint32_t mw = f2u(a.w);
                    uint32_t cls = mw >> 28;
                    if (cls == 1u) continue;
                    if (cls == 2u) {
                        float au = u, av = v;
                        if (MODE == Q_PRIMARY_MS) {
                            V3 pc = cross3(dc, e2);
                            float detc = dot3(e1, pc);
                            if (detc == 0.0f) continue;       // plane edge-on to the centre ray: no fragment value exists
                            float invc = 1.0f / detc;
                            au = dot3(tvec, pc) * invc;
                            av = dot3(dc, qvec) * invc;
                        }
                        if (!alpha_test(sc, mw & 0x0fffffffu, f2u(b.w), au, av)) continue;
                    }
                }
                tbest = t; best_flat = flat; best_tri = ti; bu = u; bv = v; found = true;
                if (MODE == Q_SHADOW) { stop = true; break; }
            }
            if (stop) break;
            if (!pop()) break;
        }
    }
    if (hit) { hit->t = tbest; hit->tri = best_tri; hit->u = bu; hit->v = bv; }
    return found;
}

// 1/x for slab tests: approximate reciprocal, magnitude capped so that 0 * inf never appears (a ray parallel to a
// slab then sees +-1e18-scaled distances, which order exactly like the infinite ones)
VKR_DI float safe_rcp(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float a = __builtin_fminf(__builtin_fabsf(r), 1.0e18f);
    return __builtin_copysignf(a, x);
}
// ---------------------------------------------------------------------------------------------
// Packet form of traverse<Q_PRIMARY / Q_PRIMARY_MS>() for coherent rays (the G-buffer producer: the 64 primary rays of an
// 8x8 pixel tile).  One wave walks the tree ONCE for its 64 rays: control flow, the node stack (one word per entry, in LDS,
// `wstack` = this wave's VKR_MAX_STACK words) and every node / triangle address are wave-uniform, each lane tests the
// fetched boxes and triangles against its own ray.  A child is opened when any lane's ray enters its box in front of that
// lane's closest hit; siblings are opened in the order of the entry distance of the first lane that hits them.  Per lane the
// triangle test, the candidate filter and the tie-break are those of traverse<>(), and the closest hit does not depend on the
// visiting order, so every lane's result is bit-identical to the per-lane walk.  `alive` = false: the lane casts no ray (it
// still has to take part: all 64 lanes of the wave must call this).
template <int MODE>
__device__ __forceinline__ bool traverse_packet(const SceneDev& sc, V3 o, V3 d, bool alive, float tmin, float tmax, HitDev* hit,
                                                uint32_t* wstack, V3 dc = V3{0.0f, 0.0f, 1.0f})
{
    static_assert(MODE == Q_PRIMARY || MODE == Q_PRIMARY_MS, "packets serve the primary rays");
    float tbest = alive ? tmax : -1.0f;                       // nothing lies inside (tmin, -1)
    uint32_t best_flat = 0xffffffffu, best_tri = 0xffffffffu;
    float bu = 0.0f, bv = 0.0f;
    bool found = false;
    if (sc.tri_count == 0) return false;
    const V3 id = V3{safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z)};   // capped: the fused slab test must never see 0 * inf
    int sp = 0;
    uint32_t cur = 0;                                          // root
#ifdef VKR_PKT_STATS
    uint32_t st_nodes = 0, st_tris = 0, st_lane_tris = 0;
#endif
    for (;;) {
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
        bool pop = false;
        if ((int)cur >= 0) {
#ifdef VKR_PKT_STATS
            st_nodes++;
#endif
            // Slab tests in the fused form of the streaming kernels (plane distance = fma(byte, scale / d, (origin - o) / d), near / far
            // byte quads picked by the direction sign), with the relative padding of box_key() because these rays are long.  The
            // address is wave-uniform and the tree is read-only while this kernel runs: through the constant address space the 64
            // bytes arrive as scalar loads (scalar cache, no trip through the vector memory path).
#if VKR_PKT_SCALAR_LOADS
            vkr_cf4p np = (vkr_cf4p)(sc.nodes + (size_t)cur * 4);
            const float4 q0 = ldk(np, 0), q1 = ldk(np, 1), q2 = ldk(np, 2), q3 = ldk(np, 3);
#else
            const float4* np = sc.nodes + (size_t)cur * 4;
            const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
#endif
            const float tb = tbest * 1.000001f;
            const float ax = q2.z * id.x, ay = q2.w * id.y, az = q0.w * id.z;
            const float bx = (q0.x - o.x) * id.x, by = (q0.y - o.y) * id.y, bz = (q0.z - o.z) * id.z;
            const uint32_t nx4 = id.x >= 0.0f ? f2u(q1.x) : f2u(q1.w), fx4 = id.x >= 0.0f ? f2u(q1.w) : f2u(q1.x);
            const uint32_t ny4 = id.y >= 0.0f ? f2u(q1.y) : f2u(q2.x), fy4 = id.y >= 0.0f ? f2u(q2.x) : f2u(q1.y);
            const uint32_t nz4 = id.z >= 0.0f ? f2u(q1.z) : f2u(q2.y), fz4 = id.z >= 0.0f ? f2u(q2.y) : f2u(q1.z);
            const uint32_t ref[4] = {f2u(q3.x), f2u(q3.y), f2u(q3.z), f2u(q3.w)};
#define VKR_PLANES(W, A, B, P01, P23)                                                                               \
    const f2v P01 = pk_fma(f2v{ubyte_f(W, 0), ubyte_f(W, 1)}, A, B), P23 = pk_fma(f2v{ubyte_f(W, 2), ubyte_f(W, 3)}, A, B);
            VKR_PLANES(nx4, ax, bx, nx01, nx23) VKR_PLANES(ny4, ay, by, ny01, ny23) VKR_PLANES(nz4, az, bz, nz01, nz23)
            VKR_PLANES(fx4, ax, bx, fx01, fx23) VKR_PLANES(fy4, ay, by, fy01, fy23) VKR_PLANES(fz4, az, bz, fz01, fz23)
#undef VKR_PLANES
            // key: entry distance bits (>= tmin >= 0: they order like unsigned integers) with the slot in the two low bits
#define VKR_SLABK(K, SLOT, NX, NY, NZ, FX, FY, FZ)                                                                  \
    uint32_t K;                                                                                                     \
    {                                                                                                               \
        float tn = vmax3(NX, NY, NZ), tf = vmin3(FX, FY, FZ);                                                       \
        tn = vmax2(tn - __builtin_fabsf(tn) * 1e-6f, tmin);                                                         \
        tf = vmin2(tf + __builtin_fabsf(tf) * 1e-6f, tb);                                                           \
        K = (tn <= tf && ref[SLOT] != VKR_CHILD_NONE) ? ((f2u(tn) & ~3u) | (uint32_t)(SLOT)) : 0xffffffffu;         \
    }
            VKR_SLABK(k0, 0, nx01.x, ny01.x, nz01.x, fx01.x, fy01.x, fz01.x)
            VKR_SLABK(k1, 1, nx01.y, ny01.y, nz01.y, fx01.y, fy01.y, fz01.y)
            VKR_SLABK(k2, 2, nx23.x, ny23.x, nz23.x, fx23.x, fy23.x, fz23.x)
            VKR_SLABK(k3, 3, nx23.y, ny23.y, nz23.y, fx23.y, fy23.y, fz23.y)
#undef VKR_SLABK
            // per child: the key of the first lane that hits it (wave-uniform; 0xffffffff when no lane does)
            auto first_key = [&](uint32_t k) -> uint32_t {
                const unsigned long long m = __ballot(k != 0xffffffffu);
                if (m == 0ull) return 0xffffffffu;
                return (uint32_t)__builtin_amdgcn_readlane((int)k, (int)(__ffsll((long long)m) - 1));
            };
            uint32_t u0 = first_key(k0), u1 = first_key(k1), u2 = first_key(k2), u3 = first_key(k3);
            cswap(u0, u1); cswap(u2, u3); cswap(u0, u2); cswap(u1, u3); cswap(u1, u2);  // ascending
            auto ref_of = [&](uint32_t key) -> uint32_t {
                const uint32_t sl = key & 3u;
                return sl == 0u ? ref[0] : (sl == 1u ? ref[1] : (sl == 2u ? ref[2] : ref[3]));
            };
            if (u0 == 0xffffffffu) pop = true;
            else {
                if (u3 != 0xffffffffu) wstack[sp++] = ref_of(u3);
                if (u2 != 0xffffffffu) wstack[sp++] = ref_of(u2);
                if (u1 != 0xffffffffu) wstack[sp++] = ref_of(u1);
                cur = ref_of(u0);
            }
        } else {
            const uint32_t first = leaf_ref_first(cur), count = leaf_ref_count(cur);
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t ti = first + k;
#ifdef VKR_PKT_STATS
                st_tris++;
#endif
#if VKR_PKT_SCALAR_LOADS
                vkr_cf4p tp = (vkr_cf4p)(sc.tris + (size_t)ti * 3);
                const float4 a = ldk(tp, 0), b = ldk(tp, 1), c = ldk(tp, 2);
#else
                const float4* tp = sc.tris + (size_t)ti * 3;
                const float4 a = tp[0], b = tp[1], c = tp[2];
#endif
                const V3 v0 = V3{a.x, a.y, a.z}, e1 = V3{b.x, b.y, b.z}, e2 = V3{c.x, c.y, c.z};
                const V3 pvec = cross3(d, e2);
                const float det = dot3(e1, pvec);
                if (det == 0.0f) continue;
                const float inv = 1.0f / det;
                const V3 tvec = o - v0;
                const float u = dot3(tvec, pvec) * inv;
                if (!(u >= 0.0f && u <= 1.0f)) continue;
                const V3 qvec = cross3(tvec, e1);
                const float v = dot3(d, qvec) * inv;
                if (!(v >= 0.0f && u + v <= 1.0f)) continue;
                const float t = dot3(e2, qvec) * inv;
                if (!(t > tmin)) continue;
                const uint32_t flat = f2u(c.w);
                if (!(t < tbest || (t == tbest && flat < best_flat))) continue;
                const uint32_t mw = f2u(a.w);
                const uint32_t cls = mw >> 28;
                if (cls == 1u) continue;
                if (cls == 2u) {
                    float au = u, av = v;
                    if (MODE == Q_PRIMARY_MS) {
                        const V3 pc = cross3(dc, e2);
                        const float detc = dot3(e1, pc);
                        if (detc == 0.0f) continue;           // plane edge-on to the centre ray: no fragment value exists
                        const float invc = 1.0f / detc;
                        au = dot3(tvec, pc) * invc;
                        av = dot3(dc, qvec) * invc;
                    }
                    if (!alpha_test(sc, mw & 0x0fffffffu, f2u(b.w), au, av)) continue;
                }
                tbest = t; best_flat = flat; best_tri = ti; bu = u; bv = v; found = true;
            }
            pop = true;
        }
        if (pop) {
            if (sp == 0) break;
            cur = wstack[--sp];
        }
    }
#ifdef VKR_PKT_STATS
    if ((threadIdx.x & 63u) == 0u && sc.pkt_stats) { atomicAdd(&sc.pkt_stats[0], 1ull); atomicAdd(&sc.pkt_stats[1], (unsigned long long)st_nodes); atomicAdd(&sc.pkt_stats[2], (unsigned long long)st_tris); }
    (void)st_lane_tris;
#endif
    if (hit) { hit->t = tbest; hit->tri = best_tri; hit->u = bu; hit->v = bv; }
    return found;
}

// ---------------------------------------------------------------------------------------------
// Shared pieces of the streaming (wave-wide queue) kernels: k_trace_q (rtao_kernels.hip), k_shadow_q
// (shadow_kernels.hip).
// ---------------------------------------------------------------------------------------------
#ifndef KQ_NQ
#define KQ_NQ 1024         // node queue capacity in LDS: 4-byte items, ray << 22 | wide node index
#endif
#ifndef KQ_LQ
#define KQ_LQ 320          // leaf queue capacity in LDS: a node step runs with < 64 leaves queued and adds <= 256
#endif
constexpr int KQ_AQ = 128;               // alpha-test queue: drained whenever it holds a full wave of candidates
constexpr uint32_t KQ_NODE_BITS = 22;    // item payload of the shadow kernels (k_shadow_q, k_shadow_sun_pkt): trees of more than 2^22 wide nodes take
                                         // their one-ray-per-lane forms (K1 sizes its own payload: q_ray_bits, rtao_kernels.hip)
constexpr uint32_t KQ_GUARD = 3u * 32u;   // head-room for strict depth-first growth: 3 deferred children per level, <= 32 levels

VKR_DI uint32_t mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// wave64 inclusive prefix sum (DPP: four shifted adds inside each row of 16, then the two row broadcasts)
VKR_DI uint32_t wave_incl_scan(uint32_t x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return (uint32_t)v;
}

// Moeller-Trumbore exactly as in k_trace / traverse<>() (operation order is part of the numeric contract): no culling,
// t strictly inside (tmin, tbest).
// 1.0f / x, correctly rounded like the IEEE division the compiler would emit (v_div_scale, v_rcp, 4 x v_fma, v_div_fmas,
// v_div_fixup), in three instructions where that is proven: v_rcp_f32 + one fma Newton step gives the same bits for EVERY
// float with biased exponent 1..252 (exhaustive check over all 2^32 inputs on gfx950: tools/calib/rcp_exact.hip,
// profiles/r2_rcp_exact.json); denormals, zeros, infinities, NaNs and |x| >= 2^126 take the division.
VKR_DI float rcp_ieee(float x)
{
#ifdef VKR_RCP_DIVIDE
    return 1.0f / x;
#endif
    const float ax = __builtin_fabsf(x);
    float r = __builtin_amdgcn_rcpf(x);
    r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    if (__builtin_expect(!(ax >= 0x1p-126f && ax < 0x1p126f), 0)) r = 1.0f / x;
    return r;
}

VKR_DI bool tri_hit(V3 ro, V3 rdir, float4 ta, float4 tb4, float4 tc, float tmin, float tbest, float* uo, float* vo,
                    float* to)
{
    V3 v0 = V3{ta.x, ta.y, ta.z}, e1 = V3{tb4.x, tb4.y, tb4.z}, e2 = V3{tc.x, tc.y, tc.z};
    V3 pvec = cross3(rdir, e2);
    float det = dot3(e1, pvec);
    float inv = rcp_ieee(det);
    V3 tvec = ro - v0;
    float u = dot3(tvec, pvec) * inv;
    V3 qvec = cross3(tvec, e1);
    float v = dot3(rdir, qvec) * inv;
    float t = dot3(e2, qvec) * inv;
    *uo = u; *vo = v; *to = t;
    return (det != 0.0f) && (u >= 0.0f && u <= 1.0f) && (v >= 0.0f && u + v <= 1.0f) && (t > tmin && t < tbest);
}

}  // namespace vkr
