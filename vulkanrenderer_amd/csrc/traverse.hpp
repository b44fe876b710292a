// traverse.hpp -- ray queries over the LBVH (replaces rayQueryEXT + the driver's TLAS/BLAS walk).
//
// Data layout in HBM (all 16-byte aligned, one dwordx4 load per quarter):
//   node (64 B)  : q0 = lmin.xyz lmax.x | q1 = lmax.yz rmin.xy | q2 = rmin.z rmax.xyz | q3 = left right hL hR
//                  child >= 0 : internal node index ; child < 0 : leaf, triangle index = ~child
//   tri  (48 B)  : q0 = v0.xyz  (mesh | class<<28) | q1 = e1.xyz prim | q2 = e2.xyz flat_id
//                  class: 0 = always an occluder for AO, 1 = never (BLEND), 2 = alpha-test (MASK)
// Traversal: one ray per lane, near-child-first, short stack in LDS ([depth][lane], conflict-free) that
// spills to per-lane scratch only beyond VKR_LDS_STACK entries.
#pragma once
#include "device_math.hpp"

namespace vkr {

constexpr int VKR_LDS_STACK = 16;   // entries per lane kept in LDS
constexpr int VKR_MAX_STACK = 96;   // hard cap (host checks BVH height against it)
constexpr int VKR_TRACE_BLOCK = 256;

struct AlphaTexDev { uint32_t offset, w, h, pad; };

struct SceneDev {
    const float4* nodes;
    const float4* tris;
    uint32_t tri_count;
    // MASK path (rtao_trace_half.comp:141-175)
    const uint2* lut;           // InstanceLUTEntry {materialID, indexBufferOffset}
    const uint32_t* indices;    // merged, global vertex ids
    const float2* uvs;          // merged
    const float4* material_params;
    const AlphaTexDev* alpha_tex;
    const uint8_t* alpha_data;
};

struct HitDev { float t; uint32_t tri; float u, v; };

enum { Q_AO = 0, Q_SHADOW = 1, Q_PRIMARY = 2 };

struct TraceCounters { uint32_t nodes, tris; };

// baseColor alpha: LOD 0, bilinear, REPEAT (rtao_trace_half.comp:170)
VKR_DI float sample_alpha(const SceneDev& sc, uint32_t mat, V2 uv)
{
    AlphaTexDev t = sc.alpha_tex[mat];
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

// Slab test of one child box against [tmin, tbest]; padded so rounding can only add visits.
VKR_DI bool box_hit(float minx, float miny, float minz, float maxx, float maxy, float maxz, V3 o, V3 id, float tmin,
                    float tbest, float* tnear)
{
    float t0 = (minx - o.x) * id.x, t1 = (maxx - o.x) * id.x;
    float tn = __builtin_fminf(t0, t1), tf = __builtin_fmaxf(t0, t1);
    t0 = (miny - o.y) * id.y; t1 = (maxy - o.y) * id.y;
    tn = __builtin_fmaxf(tn, __builtin_fminf(t0, t1)); tf = __builtin_fminf(tf, __builtin_fmaxf(t0, t1));
    t0 = (minz - o.z) * id.z; t1 = (maxz - o.z) * id.z;
    tn = __builtin_fmaxf(tn, __builtin_fminf(t0, t1)); tf = __builtin_fminf(tf, __builtin_fmaxf(t0, t1));
    tn = __builtin_fmaxf(tn, tmin); tf = __builtin_fminf(tf, tbest);
    *tnear = tn;
    return tn * 0.999999f <= tf * 1.000001f;
}

// MODE Q_AO      : closest committed hit, candidates filtered by class / alpha test (queryAoClosestHit)
// MODE Q_SHADOW  : any hit, every triangle occludes (gl_RayFlagsOpaqueEXT|TerminateOnFirstHit, pbr.frag:199-215)
// MODE Q_PRIMARY : as Q_AO, ties in t resolved towards the lower flat triangle id (G-buffer producer)
// `stack` points at this lane's column of the block's LDS stack: entry k at stack[k * VKR_TRACE_BLOCK].
template <int MODE, bool STATS>
__device__ __forceinline__ bool traverse(const SceneDev& sc, V3 o, V3 d, float tmin, float tmax, HitDev* hit,
                                         uint32_t* stack, TraceCounters* cnt)
{
    float tbest = tmax;
    uint32_t best_flat = 0xffffffffu, best_tri = 0xffffffffu;
    float bu = 0.0f, bv = 0.0f;
    bool found = false;
    if (sc.tri_count == 0) return false;

    V3 id = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    uint32_t spill[VKR_MAX_STACK - VKR_LDS_STACK];
    int sp = 0;
    int node = 0;

    auto test_leaf = [&](int child) -> bool {
        uint32_t ti = (uint32_t)(~child);
        const float4* tp = sc.tris + (size_t)ti * 3;
        float4 a = tp[0], b = tp[1], c = tp[2];
        if (STATS) cnt->tris++;
        V3 v0 = V3{a.x, a.y, a.z}, e1 = V3{b.x, b.y, b.z}, e2 = V3{c.x, c.y, c.z};
        V3 pvec = cross3(d, e2);
        float det = dot3(e1, pvec);
        if (det == 0.0f) return false;
        float inv = 1.0f / det;
        V3 tvec = o - v0;
        float u = dot3(tvec, pvec) * inv;
        if (!(u >= 0.0f && u <= 1.0f)) return false;
        V3 qvec = cross3(tvec, e1);
        float v = dot3(d, qvec) * inv;
        if (!(v >= 0.0f && u + v <= 1.0f)) return false;
        float t = dot3(e2, qvec) * inv;
        if (!(t > tmin)) return false;
        uint32_t flat = f2u(c.w);
        if (MODE == Q_PRIMARY) {
            if (!(t < tbest || (t == tbest && flat < best_flat))) return false;
        } else {
            if (!(t < tbest)) return false;
        }
        if (MODE != Q_SHADOW) {
            uint32_t mw = f2u(a.w);
            uint32_t cls = mw >> 28;
            if (cls == 1u) return false;
            if (cls == 2u && !alpha_test(sc, mw & 0x0fffffffu, f2u(b.w), u, v)) return false;
        }
        tbest = t; best_flat = flat; best_tri = ti; bu = u; bv = v; found = true;
        return true;
    };

    while (true) {
        const float4* np = sc.nodes + (size_t)node * 4;
        float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
        if (STATS) cnt->nodes++;
        int left = (int)f2u(q3.x), right = (int)f2u(q3.y);
        float tnl, tnr;
        bool hl = box_hit(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, id, tmin, tbest, &tnl);
        if (hl && left < 0) {
            bool h = test_leaf(left);
            if (MODE == Q_SHADOW && h) break;
            hl = false;
        }
        bool hr = box_hit(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, id, tmin, tbest, &tnr);
        if (hr && right < 0) {
            bool h = test_leaf(right);
            if (MODE == Q_SHADOW && h) break;
            hr = false;
        }
        if (hl && hr) {
            int nearc = left, farc = right;
            if (tnr < tnl) { nearc = right; farc = left; }
            if (sp < VKR_LDS_STACK) stack[sp * VKR_TRACE_BLOCK] = (uint32_t)farc;
            else spill[sp - VKR_LDS_STACK] = (uint32_t)farc;
            ++sp;
            node = nearc;
        } else if (hl) {
            node = left;
        } else if (hr) {
            node = right;
        } else {
            if (sp == 0) break;
            --sp;
            node = (int)((sp < VKR_LDS_STACK) ? stack[sp * VKR_TRACE_BLOCK] : spill[sp - VKR_LDS_STACK]);
        }
    }
    if (hit) { hit->t = tbest; hit->tri = best_tri; hit->u = bu; hit->v = bv; }
    return found;
}

}  // namespace vkr
