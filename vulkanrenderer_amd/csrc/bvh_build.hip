// bvh_build.hip -- GPU acceleration-structure build / refit.  Replaces the driver's acceleration-structure builder behind
// RayTracingContext::buildBottomLevelASes / buildTopLevelAS / updateTopLevelAS
// (RayTracingContext.cpp:199-345, 165-197).
//
// The reference is two-level (one BLAS per mesh + a TLAS of instances).  Here instances are flattened to
// world-space triangles (instanced meshes are duplicated) and ONE tree is built over them:
//   bounds -> Morton codes (21 or 10 bits/axis) -> LSD radix sort (8-bit digits, LDS ranking)
//   -> binned-SAH splits of every range above 1024 triangles (path bits go in front of the Morton bits, 4 more sort passes)
//   -> Karras 2012 hierarchy -> PLOC rebuild of the subtrees of <= 1024 triangles (LDS) -> bottom-up AABB fit (atomic
//   arrival flags) -> breadth-first collapse into 4-ary nodes with 8-bit quantised child boxes (traverse.hpp layout).
// Refit = re-transform the triangles in the stored order + a level-by-level refit of the 4-ary tree (topology kept).
#include <stdio.h>

#include "internal.hpp"

namespace vkr {

struct BuildIn {
    const float* positions;
    const uint32_t* indices;
    const uint2* lut;
    const uint32_t* mesh_class;
    const uint32_t* inst_mesh;
    const uint32_t* inst_first_tri;
    const float* inst_xf;
    uint32_t instance_count;
    uint32_t tri_count;
};

struct TriW { V3 v0, v1, v2; uint32_t mesh, prim; };

// 3x4 row-major instance transform (RayTracingContext.cpp:28-44), unfused left-to-right.
VKR_DI V3 xform(const float* m, V3 p)
{
    return V3{((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
              ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]};
}

// flat triangle id -> (instance, primitive) -> world-space vertices
VKR_DI TriW emit_tri(const BuildIn& in, uint32_t flat)
{
    uint32_t lo = 0, hi = in.instance_count;  // find inst: first[inst] <= flat < first[inst+1]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (in.inst_first_tri[mid] <= flat) lo = mid; else hi = mid;
    }
    uint32_t inst = lo;
    uint32_t mesh = in.inst_mesh[inst];
    uint32_t prim = flat - in.inst_first_tri[inst];
    uint32_t ioff = in.lut[mesh].y;
    const float* xf = in.inst_xf + (size_t)inst * 12;
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = xf[k];
    TriW t;
    t.mesh = mesh;
    t.prim = prim;
    V3 v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint32_t gi = in.indices[ioff + 3 * prim + k];
        const float* p = in.positions + (size_t)gi * 3;
        v[k] = xform(m, V3{p[0], p[1], p[2]});
    }
    t.v0 = v[0]; t.v1 = v[1]; t.v2 = v[2];
    return t;
}

VKR_DI uint32_t enc_f(float f) { uint32_t u = f2u(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

VKR_DI float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = __builtin_fminf(v, __shfl_xor(v, o, 64));
    return v;
}
VKR_DI float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ void k_bounds_init(BvhBounds* b)
{
    if (threadIdx.x < 3) {
        b->tri_min[threadIdx.x] = 0xffffffffu; b->tri_max[threadIdx.x] = 0u;
        b->cen_min[threadIdx.x] = 0xffffffffu; b->cen_max[threadIdx.x] = 0u;
    }
}

__global__ __launch_bounds__(256) void k_bounds(BuildIn in, BvhBounds* out)
{
    // grid-stride: a bounded number of blocks, so the 12 atomics per block do not queue up on 12 addresses
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (uint32_t flat = blockIdx.x * 256u + threadIdx.x; flat < in.tri_count; flat += gridDim.x * 256u) {
        TriW t = emit_tri(in, flat);
        const float tl[3] = {__builtin_fminf(t.v0.x, __builtin_fminf(t.v1.x, t.v2.x)), __builtin_fminf(t.v0.y, __builtin_fminf(t.v1.y, t.v2.y)),
                             __builtin_fminf(t.v0.z, __builtin_fminf(t.v1.z, t.v2.z))};
        const float th[3] = {__builtin_fmaxf(t.v0.x, __builtin_fmaxf(t.v1.x, t.v2.x)), __builtin_fmaxf(t.v0.y, __builtin_fmaxf(t.v1.y, t.v2.y)),
                             __builtin_fmaxf(t.v0.z, __builtin_fmaxf(t.v1.z, t.v2.z))};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float cc = (tl[k] + th[k]) * 0.5f;
            lo[k] = __builtin_fminf(lo[k], tl[k]); hi[k] = __builtin_fmaxf(hi[k], th[k]);
            clo[k] = __builtin_fminf(clo[k], cc); chi[k] = __builtin_fmaxf(chi[k], cc);
        }
    }
    __shared__ float red[4][12];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float a = wave_min(lo[k]), b = wave_max(hi[k]), c = wave_min(clo[k]), d = wave_max(chi[k]);
        if (lane == 0) { red[wave][k] = a; red[wave][3 + k] = b; red[wave][6 + k] = c; red[wave][9 + k] = d; }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        int k = threadIdx.x;
        bool is_min = (k < 3) || (k >= 6 && k < 9);
        float v = red[0][k];
        for (int w = 1; w < 4; ++w) v = is_min ? __builtin_fminf(v, red[w][k]) : __builtin_fmaxf(v, red[w][k]);
        if (v > -3.0e38f && v < 3.0e38f) {
            uint32_t e = enc_f(v);
            if (k < 3) atomicMin(&out->tri_min[k], e);
            else if (k < 6) atomicMax(&out->tri_max[k - 3], e);
            else if (k < 9) atomicMin(&out->cen_min[k - 6], e);
            else atomicMax(&out->cen_max[k - 9], e);
        }
    }
}

VKR_DI uint64_t expand3(uint64_t x)
{
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void k_morton(BuildIn in, const BvhBounds* bnd, uint32_t bits, int emc, uint64_t* keys,
                                                uint32_t* vals)
{
    uint32_t flat = blockIdx.x * 256u + threadIdx.x;
    if (flat >= in.tri_count) return;
    TriW t = emit_tri(in, flat);
    const float tv[3][3] = {{t.v0.x, t.v1.x, t.v2.x}, {t.v0.y, t.v1.y, t.v2.y}, {t.v0.z, t.v1.z, t.v2.z}};
    float c[3];
    c[0] = (__builtin_fminf(t.v0.x, __builtin_fminf(t.v1.x, t.v2.x)) + __builtin_fmaxf(t.v0.x, __builtin_fmaxf(t.v1.x, t.v2.x))) * 0.5f;
    c[1] = (__builtin_fminf(t.v0.y, __builtin_fminf(t.v1.y, t.v2.y)) + __builtin_fmaxf(t.v0.y, __builtin_fmaxf(t.v1.y, t.v2.y))) * 0.5f;
    c[2] = (__builtin_fminf(t.v0.z, __builtin_fminf(t.v1.z, t.v2.z)) + __builtin_fmaxf(t.v0.z, __builtin_fmaxf(t.v1.z, t.v2.z))) * 0.5f;
    float scale = (float)(1u << bits);
    uint32_t qmax = (1u << bits) - 1u;
    uint64_t q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float lo = dec_f(bnd->cen_min[k]), hi = dec_f(bnd->cen_max[k]);
        float ext = hi - lo;
        float x = (ext > 0.0f) ? ((c[k] - lo) / ext) : 0.0f;
        float s = x * scale;
        uint32_t v = (s > 0.0f) ? (uint32_t)s : 0u;   // NaN -> 0
        q[k] = v > qmax ? qmax : v;
    }
    if (emc) {
        // Extended Morton code (Vinkler, Bittner, Havran 2017): a size component interleaved with x, y, z, so a
        // triangle larger than the current cell leaves the spatial subdivision at that level and is grouped with
        // the other large triangles of the cell instead of inflating the boxes of its small neighbours.
        float srel = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float lo = dec_f(bnd->cen_min[k]), hi = dec_f(bnd->cen_max[k]);
            float ext = hi - lo;
            float tlo = __builtin_fminf(tv[k][0], __builtin_fminf(tv[k][1], tv[k][2]));
            float thi = __builtin_fmaxf(tv[k][0], __builtin_fmaxf(tv[k][1], tv[k][2]));
            float r = (ext > 0.0f) ? (thi - tlo) / ext : 0.0f;
            srel = __builtin_fmaxf(srel, r);
        }
        float sf = srel * scale;
        uint32_t sq = (sf > 0.0f) ? (uint32_t)sf : 0u;
        if (sq > qmax) sq = qmax;
        uint64_t key = 0;
        for (int b = (int)bits - 1; b >= 0; --b) {
            key = (key << 4) | (((q[0] >> b) & 1ull) << 3) | (((q[1] >> b) & 1ull) << 2) | (((q[2] >> b) & 1ull) << 1) |
                  ((uint64_t)(sq >> b) & 1ull);
        }
        keys[flat] = key;
    } else {
        keys[flat] = (expand3(q[0]) << 2) | (expand3(q[1]) << 1) | expand3(q[2]);
    }
    vals[flat] = flat;
}

// ---------------------------------------------------------------- radix sort (LSD, 8-bit digits) --------
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint64_t* keys, uint32_t n, int shift, uint32_t ntiles,
                                                        uint32_t* hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        uint32_t idx = base + i * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of 256 values held one per thread; returns the exclusive prefix, *total = block sum
VKR_DI uint32_t block_excl_scan(uint32_t v, uint32_t* lds4, uint32_t* total)
{
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    __syncthreads();
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t x = lds4[w];
        if (w < wave) wbase += x;
        tot += x;
    }
    *total = tot;
    return wbase + incl - v;
}

// one block per digit: exclusive scan of that digit's per-tile counts, digit total out
__global__ __launch_bounds__(RS_THREADS) void k_rs_scan(uint32_t* hist, uint32_t ntiles, uint32_t* totals)
{
    __shared__ uint32_t lds4[4];
    uint32_t* row = hist + (size_t)blockIdx.x * ntiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += RS_THREADS) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = (i < ntiles) ? row[i] : 0u;
        uint32_t tot;
        uint32_t ex = block_excl_scan(v, lds4, &tot);
        if (i < ntiles) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint64_t* keys_in, const uint32_t* vals_in,
                                                           uint64_t* keys_out, uint32_t* vals_out, uint32_t n,
                                                           int shift, uint32_t ntiles, const uint32_t* hist,
                                                           const uint32_t* totals)
{
    __shared__ uint32_t cnt[RS_ITEMS][4][256];  // per (round, wave, digit): count, then exclusive offset
    __shared__ uint32_t dbase[256];
    __shared__ uint32_t lds4[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_ITEMS * 4 * 256; i += RS_THREADS) (&cnt[0][0][0])[i] = 0;
    {   // exclusive scan of the 256 digit totals -> global start of each digit
        uint32_t tot;
        uint32_t ex = block_excl_scan(totals[threadIdx.x], lds4, &tot);
        dbase[threadIdx.x] = ex + hist[(size_t)threadIdx.x * ntiles + blockIdx.x];
    }
    __syncthreads();

    uint64_t key[RS_ITEMS];
    uint32_t val[RS_ITEMS];
    uint32_t rank[RS_ITEMS];
    const uint32_t base = blockIdx.x * RS_TILE;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        uint32_t idx = base + i * RS_THREADS + threadIdx.x;
        bool valid = idx < n;
        key[i] = valid ? keys_in[idx] : ~0ull;
        val[i] = valid ? vals_in[idx] : 0u;
        uint32_t digit = (uint32_t)(key[i] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            unsigned long long m = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? m : ~m;
        }
        rank[i] = (uint32_t)__popcll(peers & lt_mask);
        if (valid && rank[i] == 0) cnt[i][wave][digit] = (uint32_t)__popcll(peers);
    }
    __syncthreads();
    {   // per digit: exclusive prefix over (round, wave) in tile order
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < RS_ITEMS; ++i)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t c = cnt[i][w][threadIdx.x];
                cnt[i][w][threadIdx.x] = run;
                run += c;
            }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        uint32_t idx = base + i * RS_THREADS + threadIdx.x;
        if (idx < n) {
            uint32_t digit = (uint32_t)(key[i] >> shift) & 255u;
            uint32_t pos = dbase[digit] + cnt[i][wave][digit] + rank[i];
            keys_out[pos] = key[i];
            vals_out[pos] = val[i];
        }
    }
}

// ---------------------------------------------------------------- triangles in Morton order -------------
// BOUNDS (refit): the scene's triangle bounds are reduced here as well, so the refit does not run a second pass over
// the instances' triangles just for them (k_bounds; the centroid bounds are only needed by the Morton codes of a build)
template <bool BOUNDS>
__global__ __launch_bounds__(256) void k_emit_sorted(BuildIn in, const uint32_t* perm, float4* tris, BvhBounds* out)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (i < in.tri_count) {
        uint32_t flat = perm[i];
        TriW t = emit_tri(in, flat);
        V3 e1 = t.v1 - t.v0, e2 = t.v2 - t.v0;
        uint32_t cls = in.mesh_class[t.mesh];
        float4* o = tris + (size_t)i * 3;
        o[0] = make_float4(t.v0.x, t.v0.y, t.v0.z, u2f((t.mesh & 0x0fffffffu) | (cls << 28)));
        o[1] = make_float4(e1.x, e1.y, e1.z, u2f(t.prim));
        o[2] = make_float4(e2.x, e2.y, e2.z, u2f(flat));
        if (BOUNDS) {
            lo[0] = __builtin_fminf(t.v0.x, __builtin_fminf(t.v1.x, t.v2.x)); hi[0] = __builtin_fmaxf(t.v0.x, __builtin_fmaxf(t.v1.x, t.v2.x));
            lo[1] = __builtin_fminf(t.v0.y, __builtin_fminf(t.v1.y, t.v2.y)); hi[1] = __builtin_fmaxf(t.v0.y, __builtin_fmaxf(t.v1.y, t.v2.y));
            lo[2] = __builtin_fminf(t.v0.z, __builtin_fminf(t.v1.z, t.v2.z)); hi[2] = __builtin_fmaxf(t.v0.z, __builtin_fmaxf(t.v1.z, t.v2.z));
        }
    }
    if (BOUNDS) {
        __shared__ float red[4][6];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = wave_min(lo[k]), b = wave_max(hi[k]);
            if (lane == 0) { red[wave][k] = a; red[wave][3 + k] = b; }
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            const int k = threadIdx.x;
            float v = red[0][k];
            for (int w = 1; w < 4; ++w) v = k < 3 ? __builtin_fminf(v, red[w][k]) : __builtin_fmaxf(v, red[w][k]);
            if (v > -3.0e38f && v < 3.0e38f) {
                // 10 k blocks: only the few that improve a bound issue the atomic (a stale read only costs a redundant one)
                const uint32_t e = enc_f(v);
                if (k < 3) { if (e < __hip_atomic_load(&out->tri_min[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&out->tri_min[k], e); }
                else { if (e > __hip_atomic_load(&out->tri_max[k - 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&out->tri_max[k - 3], e); }
            }
        }
    }
}

// Alpha records: for every MASK triangle (class 2) of the final order its three UVs, its material's cutoff and alpha-texture
// descriptor, 48 bytes beside the triangle, so that the alpha test of a candidate (rtao_trace_half.comp:163-173) loads them
// together with the triangle instead of walking lut -> indices -> uvs and lut -> material tables.  Same values, same
// arithmetic afterwards (traverse.hpp alpha_test_rec).
__global__ __launch_bounds__(256) void k_alpha_records(const float4* tris, uint32_t n, const uint2* lut, const uint32_t* indices,
                                                       const float2* uvs, const float4* material_params, const AlphaTexDev* alpha_tex,
                                                       float4* rec)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t mw = f2u(tris[(size_t)i * 3].w);
    if ((mw >> 28) != 2u) return;
    const uint32_t prim = f2u(tris[(size_t)i * 3 + 1].w);
    const uint2 l = lut[mw & 0x0fffffffu];
    const uint32_t i0 = indices[l.y + prim * 3u + 0u], i1 = indices[l.y + prim * 3u + 1u], i2 = indices[l.y + prim * 3u + 2u];
    const float2 a = uvs[i0], b = uvs[i1], c = uvs[i2];
    const AlphaTexDev t = alpha_tex[l.x];
    rec[(size_t)i * 3] = make_float4(a.x, a.y, b.x, b.y);
    rec[(size_t)i * 3 + 1] = make_float4(c.x, c.y, material_params[l.x].x, 0.0f);
    rec[(size_t)i * 3 + 2] = make_float4(u2f(t.offset), u2f(t.w), u2f(t.h), u2f(l.x));
}

#ifndef VKR_TL_MAX
#define VKR_TL_MAX 1024
#endif
#ifndef VKR_TL_R
#define VKR_TL_R 16
#endif
constexpr int TL_MAX = VKR_TL_MAX;     // triangles per treelet = threads per block (64 .. 1024)
constexpr int TL_MIN = 4;
constexpr int TL_R = VKR_TL_R;         // search window (positions either side)
constexpr int TL_WAVES = TL_MAX / 64;

// ---------------------------------------------------------------- binned-SAH top levels ------------------
// The reference asks its driver for ePreferFastTrace (RayTracingContext.cpp:237, :305).  The Karras tree splits by Morton
// bits; this phase replaces its TOP by surface-area-heuristic splits: level by level, every range of more than TL_MAX
// triangles is split at the best of 16 centroid bins along the widest axis of its centroid bounds (measured offline,
// tools/exp/bvh_quality_exp.cpp: one axis is as good as three; -11 % node visits per AO ray on top of the treelets).
// Nothing is moved while the levels run: a triangle only carries the id of its current range and the left/right path that
// led there.  At the end the path becomes the leading bits of the sort key (followed by the Morton bits), four more
// radix passes over the top 32 bits re-sort the array stably, and the unchanged Karras kernel then reproduces the SAH
// splits from the path bits and goes on by Morton bits below them -- where the treelet rebuild takes over.  A range
// that cannot be split (coincident centroids, depth 32) simply keeps its path and is left to the Morton bits.
#ifndef VKR_SAH_LEAF
#define VKR_SAH_LEAF VKR_TL_MAX
#endif
#ifndef VKR_SAH_BINS
#define VKR_SAH_BINS 16
#endif
constexpr int SAH_LEAF = VKR_SAH_LEAF;       // ranges of more triangles than this are split further (<= TL_MAX)
constexpr int SAH_BINS = VKR_SAH_BINS;
constexpr int SAH_DEPTH = 32;                 // path bits
constexpr uint32_t SAH_FINAL = 0xffffffffu;
constexpr int SAH_CHUNK = 1024;               // triangles per k_sah_bin block
constexpr int SAH_WORDS = 13;                // per bin: count, box ~enc(lo.xyz) enc(hi.xyz), centroids ~enc(min.xyz) enc(max.xyz)
constexpr int SAH_BIN_WORDS = SAH_BINS * SAH_WORDS;   // all words grow from 0 (atomicAdd / atomicMax)

struct SahNode {
    uint32_t cmin[3], cmax[3];   // centroid bounds as ~enc_f(min), enc_f(max): both grown with atomicMax from 0
    int32_t axis;                // split axis; -1 = not split
    int32_t ksplit;              // bins <= ksplit go left
    uint32_t child[2];           // id among the next level's ranges, or SAH_FINAL
    uint32_t pad[2];
};

VKR_DI int sah_axis(const SahNode& nd, float& c0, float& inv)
{
    const float l0 = dec_f(~nd.cmin[0]), l1 = dec_f(~nd.cmin[1]), l2 = dec_f(~nd.cmin[2]);
    const float e0 = dec_f(nd.cmax[0]) - l0, e1 = dec_f(nd.cmax[1]) - l1, e2 = dec_f(nd.cmax[2]) - l2;
    float e = e0, l = l0;      // widest axis, ties to the lower one (written as updates: the nested-select form crashes
    int a = 0;                 // hipcc 7.2's instruction selection)
    if (e1 > e) { e = e1; l = l1; a = 1; }
    if (e2 > e) { e = e2; l = l2; a = 2; }
    c0 = l;
    inv = (float)SAH_BINS / e;
    return (e > 0.0f && e < 3.0e38f) ? a : -1;
}
VKR_DI int sah_bin(float c, float c0, float inv)
{
    const float s = (c - c0) * inv;
    return s > 0.0f ? (s < (float)(SAH_BINS - 1) ? (int)s : SAH_BINS - 1) : 0;   // NaN -> 0
}

// triangle boxes in Morton order + per-triangle state; thread 0 seeds level 0 with the whole array as one range
__global__ __launch_bounds__(256) void k_sah_prep(BuildIn in, const uint32_t* perm, float4* box, uint32_t* tnode, uint32_t* path,
                                                  uint8_t* depth, SahNode* nodes0, const BvhBounds* bnd, uint32_t* level_count)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0u) {
        SahNode r;
#pragma unroll
        for (int k = 0; k < 3; ++k) { r.cmin[k] = ~bnd->cen_min[k]; r.cmax[k] = bnd->cen_max[k]; }
        r.axis = -1; r.ksplit = 0; r.child[0] = r.child[1] = SAH_FINAL; r.pad[0] = r.pad[1] = 0u;
        nodes0[0] = r;
        level_count[0] = 1u;
    }
    if (i >= in.tri_count) return;
    const TriW t = emit_tri(in, perm[i]);
    box[2u * i] = make_float4(__builtin_fminf(t.v0.x, __builtin_fminf(t.v1.x, t.v2.x)), __builtin_fminf(t.v0.y, __builtin_fminf(t.v1.y, t.v2.y)),
                              __builtin_fminf(t.v0.z, __builtin_fminf(t.v1.z, t.v2.z)), 0.0f);
    box[2u * i + 1u] = make_float4(__builtin_fmaxf(t.v0.x, __builtin_fmaxf(t.v1.x, t.v2.x)), __builtin_fmaxf(t.v0.y, __builtin_fmaxf(t.v1.y, t.v2.y)),
                                   __builtin_fmaxf(t.v0.z, __builtin_fmaxf(t.v1.z, t.v2.z)), 0.0f);
    tnode[i] = 0u; path[i] = 0u; depth[i] = 0;
}

// A: bin the triangles of every open range.  Every bin word is a count or a maximum (minima are stored complemented), so
// lanes that fall into the same (range, bin) -- in Morton order that is most of a wave -- are first combined in the wave
// and one lane issues the atomics.  Each wave takes 256 consecutive triangles and keeps LDS bins for the range of its
// first one; other ranges go to the global bins directly.
VKR_DI uint32_t wave_umax(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64); v = v > w ? v : w; }
    return v;
}

__global__ __launch_bounds__(256) void k_sah_bin(uint32_t n, const uint32_t* tnode, const float4* box, const SahNode* nodes,
                                                 const uint32_t* level_count, uint32_t* bins)
{
    if (*level_count == 0u) return;
    __shared__ uint32_t s_all[4][SAH_BIN_WORDS];      // one set of bins per wave: the range of the wave's first triangle
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* s_bins = s_all[wave];
    const uint32_t base = blockIdx.x * (uint32_t)SAH_CHUNK + (uint32_t)wave * (uint32_t)(SAH_CHUNK / 4);   // 256 consecutive triangles per wave
    for (int w = lane; w < SAH_BIN_WORDS; w += 64) s_bins[w] = 0u;
    const uint32_t mine = base < n ? tnode[base] : SAH_FINAL;    // wave-uniform
    __syncthreads();
    for (int j = 0; j < SAH_CHUNK / 256; ++j) {
        const uint32_t i = base + (uint32_t)j * 64u + (uint32_t)lane;
        uint32_t nd = i < n ? tnode[i] : SAH_FINAL;
        uint32_t v[SAH_WORDS - 1];
#pragma unroll
        for (int q = 0; q < SAH_WORDS - 1; ++q) v[q] = 0u;
        int k = 0;
        if (nd != SAH_FINAL) {
            const SahNode node = nodes[nd];
            float c0, inv;
            const int a = sah_axis(node, c0, inv);
            if (a < 0) nd = SAH_FINAL;
            else {
                const float4 lo = box[2u * i], hi = box[2u * i + 1u];
                const float cx = (lo.x + hi.x) * 0.5f, cy = (lo.y + hi.y) * 0.5f, cz = (lo.z + hi.z) * 0.5f;
                k = sah_bin(a == 0 ? cx : (a == 1 ? cy : cz), c0, inv);
                v[0] = ~enc_f(lo.x); v[1] = ~enc_f(lo.y); v[2] = ~enc_f(lo.z);
                v[3] = enc_f(hi.x); v[4] = enc_f(hi.y); v[5] = enc_f(hi.z);
                v[6] = ~enc_f(cx); v[7] = ~enc_f(cy); v[8] = ~enc_f(cz);
                v[9] = enc_f(cx); v[10] = enc_f(cy); v[11] = enc_f(cz);
            }
        }
        bool act = nd != SAH_FINAL;
        const uint32_t key = nd * (uint32_t)SAH_BINS + (uint32_t)k;
        for (int round = 0; round < 4; ++round) {                  // up to four distinct (range, bin) pairs are combined
            const unsigned long long mask = __ballot(act);
            if (mask == 0ull) break;
            const int leader = __ffsll((long long)mask) - 1;
            const uint32_t lkey = (uint32_t)__shfl((int)key, leader, 64);
            const bool match = act && key == lkey;
            const uint32_t cnt = (uint32_t)__popcll(__ballot(match));
            uint32_t r[SAH_WORDS - 1];
#pragma unroll
            for (int q = 0; q < SAH_WORDS - 1; ++q) r[q] = wave_umax(match ? v[q] : 0u);
            if (lane == leader) {
                if (nd == mine) {
                    atomicAdd(&s_bins[k * SAH_WORDS], cnt);
#pragma unroll
                    for (int q = 0; q < SAH_WORDS - 1; ++q) atomicMax(&s_bins[k * SAH_WORDS + 1 + q], r[q]);
                } else {
                    uint32_t* b = bins + (size_t)key * SAH_WORDS;
                    atomicAdd(b, cnt);
#pragma unroll
                    for (int q = 0; q < SAH_WORDS - 1; ++q) atomicMax(b + 1 + q, r[q]);
                }
            }
            act = act && !match;
        }
        if (act) {                                                  // a wave that straddles many bins: the rest one by one
            if (nd == mine) {
                atomicAdd(&s_bins[k * SAH_WORDS], 1u);
#pragma unroll
                for (int q = 0; q < SAH_WORDS - 1; ++q) atomicMax(&s_bins[k * SAH_WORDS + 1 + q], v[q]);
            } else {
                uint32_t* b = bins + (size_t)key * SAH_WORDS;
                atomicAdd(b, 1u);
#pragma unroll
                for (int q = 0; q < SAH_WORDS - 1; ++q) atomicMax(b + 1 + q, v[q]);
            }
        }
    }
    __syncthreads();
    if (mine != SAH_FINAL) {
        for (int w = lane; w < SAH_BIN_WORDS; w += 64) {
            const uint32_t v = s_bins[w];
            if (v) {
                uint32_t* dst = bins + (size_t)mine * SAH_BIN_WORDS + w;
                if (w % SAH_WORDS == 0) atomicAdd(dst, v); else atomicMax(dst, v);
            }
        }
    }
}

VKR_DI float sah_half_area(const float lo[3], const float hi[3])
{
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return ex * ey + ey * ez + ez * ex;
}

// B: one thread per open range picks the split and opens the children that are still too large; their centroid bounds
// come from the bins
__global__ __launch_bounds__(64) void k_sah_split(SahNode* cur, SahNode* next, const uint32_t* bins, uint32_t* level_count, int level)
{
    const uint32_t id = blockIdx.x * 64u + threadIdx.x;
    if (id >= level_count[0]) return;
    const SahNode nd = cur[id];
    float c0, inv;
    const int a = sah_axis(nd, c0, inv);
    int best_k = -1;
    uint32_t best_lc = 0u, total = 0u;
    const uint32_t* b = bins + (size_t)id * SAH_BIN_WORDS;
    if (a >= 0) {
        float ra[SAH_BINS];
        uint32_t rc[SAH_BINS];
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        uint32_t cnt = 0u;
#pragma unroll
        for (int k = SAH_BINS - 1; k >= 1; --k) {
            const uint32_t ck = b[k * SAH_WORDS];
            if (ck) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    lo[q] = __builtin_fminf(lo[q], dec_f(~b[k * SAH_WORDS + 1 + q]));
                    hi[q] = __builtin_fmaxf(hi[q], dec_f(b[k * SAH_WORDS + 4 + q]));
                }
            }
            cnt += ck;
            rc[k] = cnt;
            ra[k] = cnt ? sah_half_area(lo, hi) : 0.0f;
        }
        total = cnt + b[0];
        float llo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, lhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        uint32_t lc = 0u;
        float best = 3.0e38f;
#pragma unroll
        for (int k = 0; k < SAH_BINS - 1; ++k) {
            const uint32_t ck = b[k * SAH_WORDS];
            if (ck) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    llo[q] = __builtin_fminf(llo[q], dec_f(~b[k * SAH_WORDS + 1 + q]));
                    lhi[q] = __builtin_fmaxf(lhi[q], dec_f(b[k * SAH_WORDS + 4 + q]));
                }
            }
            lc += ck;
            if (lc == 0u || rc[k + 1] == 0u) continue;
            const float cost = sah_half_area(llo, lhi) * (float)lc + ra[k + 1] * (float)rc[k + 1];
            if (cost < best) { best = cost; best_k = k; best_lc = lc; }
        }
    }
    if (best_k < 0) { cur[id].axis = -1; return; }
    const uint32_t cnt2[2] = {best_lc, total - best_lc};
    uint32_t child[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        child[s] = SAH_FINAL;
        if (cnt2[s] > (uint32_t)SAH_LEAF && level + 1 < SAH_DEPTH) {
            const uint32_t cid = atomicAdd(&level_count[1], 1u);
            SahNode z;
#pragma unroll
            for (int q = 0; q < 3; ++q) { z.cmin[q] = 0u; z.cmax[q] = 0u; }
            const int k0 = s == 0 ? 0 : best_k + 1, k1 = s == 0 ? best_k : SAH_BINS - 1;
            for (int k = k0; k <= k1; ++k) {
                if (b[k * SAH_WORDS] == 0u) continue;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const uint32_t mn = b[k * SAH_WORDS + 7 + q], mx = b[k * SAH_WORDS + 10 + q];
                    z.cmin[q] = z.cmin[q] > mn ? z.cmin[q] : mn;
                    z.cmax[q] = z.cmax[q] > mx ? z.cmax[q] : mx;
                }
            }
            z.axis = -1; z.ksplit = 0; z.child[0] = z.child[1] = SAH_FINAL; z.pad[0] = z.pad[1] = 0u;
            next[cid] = z;
            child[s] = cid;
        }
    }
    cur[id].axis = a;
    cur[id].ksplit = best_k;
    cur[id].child[0] = child[0];
    cur[id].child[1] = child[1];
}

// C: every triangle of a split range takes its side: one more path bit and the child's id
__global__ __launch_bounds__(256) void k_sah_assign(uint32_t n, uint32_t* tnode, const float4* box, uint32_t* path, uint8_t* depth,
                                                    const SahNode* cur, const uint32_t* level_count)
{
    if (*level_count == 0u) return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t nd = tnode[i];
    if (nd == SAH_FINAL) return;
    const SahNode node = cur[nd];
    uint32_t child = SAH_FINAL;
    if (node.axis >= 0) {
        float c0, inv;
        const int a = sah_axis(node, c0, inv);
        const float4 lo = box[2u * i], hi = box[2u * i + 1u];
        const float c = a == 0 ? (lo.x + hi.x) * 0.5f : (a == 1 ? (lo.y + hi.y) * 0.5f : (lo.z + hi.z) * 0.5f);
        const int k = sah_bin(c, c0, inv);
        const uint32_t side = k > node.ksplit ? 1u : 0u;
        path[i] = (path[i] << 1) | side;
        depth[i] = (uint8_t)(depth[i] + 1);
        child = side ? node.child[1] : node.child[0];
    }
    tnode[i] = child;
}

// the path in front of the (left-justified) Morton bits
__global__ __launch_bounds__(256) void k_sah_rekey(uint32_t n, uint64_t* keys, const uint32_t* path, const uint8_t* depth, int key_bits)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t m = keys[i] << (64 - key_bits);
    const uint32_t d = depth[i];
    keys[i] = d ? (((uint64_t)path[i] << (64u - d)) | (m >> d)) : m;
}

// ---------------------------------------------------------------- Karras 2012 hierarchy ------------------
VKR_DI int delta(const uint64_t* keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}

// binary child encoding: >= 0 internal node, < 0 leaf ~index.  bin_range = [first, last] sorted-triangle range.
__global__ __launch_bounds__(256) void k_hierarchy(const uint64_t* keys, int n, int2* bin_child, uint2* bin_range,
                                                   uint32_t* leaf_parent, uint32_t* node_parent)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + (d < 0 ? -1 : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    int left, right;
    if (lo == gamma) { left = ~gamma; leaf_parent[gamma] = (uint32_t)i; }
    else { left = gamma; node_parent[gamma] = (uint32_t)i; }
    if (hi == gamma + 1) { right = ~(gamma + 1); leaf_parent[gamma + 1] = (uint32_t)i; }
    else { right = gamma + 1; node_parent[gamma + 1] = (uint32_t)i; }
    bin_child[i] = make_int2(left, right);
    bin_range[i] = make_uint2((uint32_t)lo, (uint32_t)hi);
    if (i == 0) node_parent[0] = 0xffffffffu;
}

// ---------------------------------------------------------------- treelet rebuild (range-local PLOC) ------
// The Karras tree splits by Morton bits, not by cost.  Every maximal subtree with <= TL_MAX triangles ("treelet") is
// rebuilt here by locally-ordered agglomerative clustering (PLOC, Meister & Bittner 2018, run block-locally in LDS):
// clusters start as the treelet's triangles in Morton order; each round every cluster picks the neighbour within TL_R
// positions whose union box has the smallest area, mutual pairs merge, the cluster list is compacted in order.
// Measured offline (tools/exp/bvh_quality_exp.cpp): -12 % node visits and -12 % triangle tests per AO ray.
// The treelet keeps its triangle range [a, b], its root keeps its index (the parent's link stays valid) and the new
// internal nodes reuse the range's own Karras indices ({a..b} minus the end the root does not sit on); triangles are
// permuted inside the range into the DFS order of the new subtree, so every subtree still owns a contiguous range.
// Runs between k_emit_sorted and k_fit: boxes and heights come from the fit, as before.


__global__ __launch_bounds__(256) void k_treelet_roots(const uint2* bin_range, const uint32_t* node_parent, int n_internal,
                                                       uint32_t* list)   // [0] = count, [1] = work counter, [2..] = roots
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_internal) return;
    const uint2 r = bin_range[i];
    const uint32_t cnt = r.y - r.x + 1u;
    if (cnt > (uint32_t)TL_MAX || cnt < (uint32_t)TL_MIN) return;
    const uint32_t p = node_parent[i];
    if (p != 0xffffffffu) {
        const uint2 pr = bin_range[p];
        if (pr.y - pr.x + 1u <= (uint32_t)TL_MAX) return;      // not maximal
    }
    list[2u + atomicAdd(&list[0], 1u)] = (uint32_t)i;
}

__global__ __launch_bounds__(TL_MAX) void k_treelet_ploc(uint32_t* list, int2* bin_child, uint2* bin_range, uint32_t* leaf_parent,
                                                      uint32_t* node_parent, float4* tris, uint32_t* perm)
{
    __shared__ float s_lo[2][3][TL_MAX], s_hi[2][3][TL_MAX];
    __shared__ short s_ref[2][TL_MAX];           // cluster -> node code: leaf t as -(t+1), created node k as k
    __shared__ unsigned short s_cnt[2][TL_MAX];  // triangles in the cluster
    __shared__ short s_nn[TL_MAX];
    __shared__ short s_nodeL[TL_MAX], s_nodeR[TL_MAX], s_nodeParent[TL_MAX];
    __shared__ unsigned short s_nodeCnt[TL_MAX], s_nodeLeft[TL_MAX];
    __shared__ unsigned char s_nodeSide[TL_MAX], s_leafSide[TL_MAX];
    __shared__ short s_leafParent[TL_MAX];
    __shared__ unsigned short s_pos[TL_MAX];
    __shared__ uint32_t s_wk[TL_WAVES], s_wc[TL_WAVES];
    __shared__ uint32_t s_item;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    for (;;) {
        if (t == 0) s_item = atomicAdd(&list[1], 1u);
        __syncthreads();
        const uint32_t item = s_item;
        __syncthreads();
        if (item >= list[0]) break;                              // uniform: every thread reads the same two values
        const int r = (int)list[2u + item];
        const uint2 rg = bin_range[r];
        const int a = (int)rg.x, m = (int)(rg.y - rg.x) + 1;     // TL_MIN <= m <= TL_MAX
        // ---- clusters = triangles in Morton order
        float4 ta = make_float4(0, 0, 0, 0), tb = ta, tc = ta;
        uint32_t my_perm = 0;
        if (t < m) {
            const float4* tp = tris + (size_t)(a + t) * 3;
            ta = tp[0]; tb = tp[1]; tc = tp[2];
            my_perm = perm[a + t];
            const float v0[3] = {ta.x, ta.y, ta.z}, v1[3] = {ta.x + tb.x, ta.y + tb.y, ta.z + tb.z}, v2[3] = {ta.x + tc.x, ta.y + tc.y, ta.z + tc.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s_lo[0][k][t] = __builtin_fminf(v0[k], __builtin_fminf(v1[k], v2[k]));
                s_hi[0][k][t] = __builtin_fmaxf(v0[k], __builtin_fmaxf(v1[k], v2[k]));
            }
            s_ref[0][t] = (short)(-(t + 1));
            s_cnt[0][t] = 1;
        }
        __syncthreads();
        int ncl = m, created = 0, cur = 0;
        while (ncl > 1) {
            // ---- nearest neighbour inside the window (smallest union area; ties: smallest index)
            if (t < ncl) {
                const float lx = s_lo[cur][0][t], ly = s_lo[cur][1][t], lz = s_lo[cur][2][t];
                const float hx = s_hi[cur][0][t], hy = s_hi[cur][1][t], hz = s_hi[cur][2][t];
                float best = 3.0e38f;
                int bj = -1;
                const int j0 = t - TL_R < 0 ? 0 : t - TL_R, j1 = t + TL_R > ncl - 1 ? ncl - 1 : t + TL_R;
                for (int j = j0; j <= j1; ++j) {
                    if (j == t) continue;
                    const float dx = __builtin_fmaxf(hx, s_hi[cur][0][j]) - __builtin_fminf(lx, s_lo[cur][0][j]);
                    const float dy = __builtin_fmaxf(hy, s_hi[cur][1][j]) - __builtin_fminf(ly, s_lo[cur][1][j]);
                    const float dz = __builtin_fmaxf(hz, s_hi[cur][2][j]) - __builtin_fminf(lz, s_lo[cur][2][j]);
                    const float area = dx * dy + dy * dz + dz * dx;
                    // ties (coincident or regularly spaced geometry): the closer position wins, then the pair partner
                    // t^1, then the smaller index -- identical boxes pair up (0,1)(2,3).. and the subtree stays
                    // balanced instead of degenerating into a chain
                    bool better = area < best;
                    if (area == best && bj >= 0) {
                        const int dj = j > t ? j - t : t - j, db = bj > t ? bj - t : t - bj;
                        better = dj < db || (dj == db && j == (t ^ 1));
                    }
                    if (better) { best = area; bj = j; }
                }
                s_nn[t] = (short)bj;
            }
            __syncthreads();
            // ---- mutual pairs merge (the lower index creates the node), survivors are compacted in order
            bool creator = false, keep = false;
            int j = -1;
            if (t < ncl) {
                j = s_nn[t];
                const bool mutual = (j >= 0) && ((int)s_nn[j] == t);
                creator = mutual && t < j;
                keep = !(mutual && t > j);
            }
            const unsigned long long mk = __ballot(keep), mc = __ballot(creator);
            if (lane == 0) { s_wk[wave] = (uint32_t)__popcll(mk); s_wc[wave] = (uint32_t)__popcll(mc); }
            __syncthreads();
            uint32_t kbase = 0, cbase = 0, ktot = 0, ctot = 0;
#pragma unroll
            for (int w = 0; w < TL_WAVES; ++w) {
                if (w < wave) { kbase += s_wk[w]; cbase += s_wc[w]; }
                ktot += s_wk[w]; ctot += s_wc[w];
            }
            const int nxt = cur ^ 1;
            if (keep) {
                const int pos = (int)(kbase + mbcnt64(mk));
                if (creator) {
                    const int k = created + (int)(cbase + mbcnt64(mc));
                    const short rl = s_ref[cur][t], rr = s_ref[cur][j];
                    s_nodeL[k] = rl; s_nodeR[k] = rr;
                    s_nodeLeft[k] = s_cnt[cur][t];
                    s_nodeCnt[k] = (unsigned short)(s_cnt[cur][t] + s_cnt[cur][j]);
                    if (rl < 0) { s_leafParent[-rl - 1] = (short)k; s_leafSide[-rl - 1] = 0; }
                    else { s_nodeParent[rl] = (short)k; s_nodeSide[rl] = 0; }
                    if (rr < 0) { s_leafParent[-rr - 1] = (short)k; s_leafSide[-rr - 1] = 1; }
                    else { s_nodeParent[rr] = (short)k; s_nodeSide[rr] = 1; }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        s_lo[nxt][c][pos] = __builtin_fminf(s_lo[cur][c][t], s_lo[cur][c][j]);
                        s_hi[nxt][c][pos] = __builtin_fmaxf(s_hi[cur][c][t], s_hi[cur][c][j]);
                    }
                    s_ref[nxt][pos] = (short)k;
                    s_cnt[nxt][pos] = s_nodeCnt[k];
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { s_lo[nxt][c][pos] = s_lo[cur][c][t]; s_hi[nxt][c][pos] = s_hi[cur][c][t]; }
                    s_ref[nxt][pos] = s_ref[cur][t];
                    s_cnt[nxt][pos] = s_cnt[cur][t];
                }
            }
            __syncthreads();
            ncl = (int)ktot; created += (int)ctot; cur = nxt;
            if (ctot == 0u) break;      // cannot happen (the globally closest pair is always mutual); guards the loop
        }
        const int root_local = m - 2;
        const bool ok = (ncl == 1) && (created == m - 1);        // uniform
        if (ok) {
            if (t == 0) { s_nodeParent[root_local] = -1; s_nodeSide[root_local] = 0; }
            __syncthreads();
            // ---- DFS position of every leaf and first leaf of every node: left-sibling counts on the way to the root
            if (t < m) {
                uint32_t off = 0;
                int p = s_leafParent[t], side = s_leafSide[t];
                for (int guard = 0; guard < TL_MAX; ++guard) {
                    if (side) off += s_nodeLeft[p];
                    if (p == root_local) break;
                    side = s_nodeSide[p];
                    p = s_nodeParent[p];
                }
                s_pos[t] = (unsigned short)off;
            }
            __syncthreads();
            // internal indices of the range: {a .. a+m-1} minus the end the root does not sit on; root keeps r
            const int pa = a + 1;                                 // first non-root index, in both cases:
            // r == a     : pool = a .. a+m-2, root = a      -> others a+1 .. a+m-2
            // r == a+m-1 : pool = a+1 .. a+m-1, root = a+m-1 -> others a+1 .. a+m-2
            auto gidx = [&](int k) { return k == root_local ? r : pa + k; };
            auto code = [&](short ref) { return ref < 0 ? ~(a + (int)s_pos[-ref - 1]) : gidx(ref); };
            if (t < m - 1) {
                const int k = t;
                uint32_t off = 0;
                int c = k;
                for (int guard = 0; guard < TL_MAX && c != root_local; ++guard) {
                    const int p = s_nodeParent[c];
                    if (s_nodeSide[c]) off += s_nodeLeft[p];
                    c = p;
                }
                const int g = gidx(k);
                bin_child[g] = make_int2(code(s_nodeL[k]), code(s_nodeR[k]));
                bin_range[g] = make_uint2((uint32_t)a + off, (uint32_t)a + off + (uint32_t)s_nodeCnt[k] - 1u);
                if (k != root_local) node_parent[g] = (uint32_t)gidx(s_nodeParent[k]);
            }
            if (t < m) {
                const int np = a + (int)s_pos[t];
                leaf_parent[np] = (uint32_t)gidx(s_leafParent[t]);
                float4* tp = tris + (size_t)np * 3;              // every thread read its triangle before the first barrier
                tp[0] = ta; tp[1] = tb; tp[2] = tc;
                perm[np] = my_perm;
            }
        }
        __syncthreads();
    }
}

// padded AABB of sorted triangle j (padding keeps the slab test conservative w.r.t. the triangle test)
VKR_DI float pad_abs_of(const BvhBounds* bnd)
{
    float e = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) e += dec_f(bnd->tri_max[k]) - dec_f(bnd->tri_min[k]);
    return 4.0e-7f * e + 1.0e-6f;   // covers the rounding of the folded slab form at every level of the tree
}

VKR_DI void tri_box(const float4* tris, uint32_t j, float pad_abs, float lo[3], float hi[3])
{
    const float4* tp = tris + (size_t)j * 3;
    float4 a = tp[0], b = tp[1], c = tp[2];
    float v0[3] = {a.x, a.y, a.z};
    float v1[3] = {a.x + b.x, a.y + b.y, a.z + b.z};
    float v2[3] = {a.x + c.x, a.y + c.y, a.z + c.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float l = __builtin_fminf(v0[k], __builtin_fminf(v1[k], v2[k]));
        float h = __builtin_fmaxf(v0[k], __builtin_fmaxf(v1[k], v2[k]));
        float pad = 1e-5f * __builtin_fmaxf(__builtin_fabsf(l), __builtin_fabsf(h)) + pad_abs;
        lo[k] = l - pad;
        hi[k] = h + pad;
    }
}

// ---------------------------------------------------------------- bottom-up fit (binary tree) ------------
// One thread per leaf walks towards the root.  At each node the first arrival deposits its box in the node's
// left/right hand-off slot and stops; the second arrival reads the sibling's deposit and continues with the
// union, which it also records as node_box[p] for the wide-node pass.
// Inter-workgroup hand-off (cdna_hip_programming.md Guideline 16, R1 + "value its add returned" form): every
// deposited byte is stored write-through at agent scope (8-byte relaxed agent atomic stores = sc1), the storing
// wave drains them (s_waitcnt vmcnt(0)) before its returning atomic add on the arrival counter, and the consumer's
// loads of the deposit are agent-scope (sc1) loads issued after its own add returned 1.  No L2 write-back fence.
// flags[] must be zero on entry.
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

VKR_DI u64 pack2(float a, float b) { return (u64)f2u(a) | ((u64)f2u(b) << 32); }

__global__ __launch_bounds__(256) void k_fit(const float4* tris, int n, const int2* bin_child, u64* deposit,
                                             float4* node_box, const uint32_t* leaf_parent,
                                             const uint32_t* node_parent, uint32_t* flags, uint32_t* height_out,
                                             const BvhBounds* bnd)
{
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float lo[3], hi[3];
    tri_box(tris, (uint32_t)j, pad_abs_of(bnd), lo, hi);
    int child = ~j;
    uint32_t p = leaf_parent[j];
    uint32_t h = 0;  // height of the subtree rooted at `child`, in internal nodes
    while (true) {
        const bool is_left = (bin_child[p].x == child);
        u64* mine = deposit + (size_t)p * 8 + (is_left ? 0 : 4);
        u64* sib = deposit + (size_t)p * 8 + (is_left ? 4 : 0);
        __hip_atomic_store(mine + 0, pack2(lo[0], lo[1]), RLX_AGENT);
        __hip_atomic_store(mine + 1, pack2(lo[2], u2f(h)), RLX_AGENT);
        __hip_atomic_store(mine + 2, pack2(hi[0], hi[1]), RLX_AGENT);
        __hip_atomic_store(mine + 3, pack2(hi[2], 0.0f), RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t old = __hip_atomic_fetch_add(&flags[p], 1u, RLX_AGENT);
        if (old == 0u) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler-only: keep the loads below the add
        u64 s0 = __hip_atomic_load(sib + 0, RLX_AGENT), s1 = __hip_atomic_load(sib + 1, RLX_AGENT);
        u64 s2 = __hip_atomic_load(sib + 2, RLX_AGENT), s3 = __hip_atomic_load(sib + 3, RLX_AGENT);
        lo[0] = __builtin_fminf(lo[0], u2f((uint32_t)s0)); lo[1] = __builtin_fminf(lo[1], u2f((uint32_t)(s0 >> 32)));
        lo[2] = __builtin_fminf(lo[2], u2f((uint32_t)s1));
        hi[0] = __builtin_fmaxf(hi[0], u2f((uint32_t)s2)); hi[1] = __builtin_fmaxf(hi[1], u2f((uint32_t)(s2 >> 32)));
        hi[2] = __builtin_fmaxf(hi[2], u2f((uint32_t)s3));
        uint32_t hs = (uint32_t)(s1 >> 32);
        h = (h > hs ? h : hs) + 1u;
        node_box[(size_t)p * 2 + 0] = make_float4(lo[0], lo[1], lo[2], u2f(h));
        node_box[(size_t)p * 2 + 1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
        if (p == 0u) { *height_out = h; return; }
        child = (int)p;
        p = node_parent[p];
    }
}

// ---------------------------------------------------------------- wide (4-ary) nodes ---------------------
// Collapse of the binary LBVH, breadth-first: wide node w stands for binary node wide_bin[w]; its children are
// found by opening, up to twice, the largest-area child that is still worth a node of its own (more than
// VKR_LEAF_MAX triangles).  Subtrees with <= VKR_LEAF_MAX triangles become one leaf (their triangles are
// contiguous in Morton order).  Wide nodes are numbered in BFS order, so each level is the work list of the
// next launch and the top of the tree is contiguous in memory.
struct WideCtl {
    uint32_t count;          // wide nodes allocated so far
    uint32_t level_begin;    // current level = [level_begin, level_end)
    uint32_t level_end;
    uint32_t levels;         // non-empty levels processed
    uint32_t level_start[72]; // level l = wide nodes [level_start[l], level_start[l + 1]) (the level-by-level refit's work lists)
};

VKR_DI float half_area(const float4& mn, const float4& mx)
{
    float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
    return dx * dy + dy * dz + dz * dx;
}

VKR_DI void cand_box(int c, const float4* node_box, const float4* tris, float pad_abs, float lo[3], float hi[3])
{
    if (c >= 0) {
        float4 mn = node_box[(size_t)c * 2], mx = node_box[(size_t)c * 2 + 1];
        lo[0] = mn.x; lo[1] = mn.y; lo[2] = mn.z; hi[0] = mx.x; hi[1] = mx.y; hi[2] = mx.z;
    } else {
        tri_box(tris, (uint32_t)(~c), pad_abs, lo, hi);
    }
}

// Writes q0..q2 of wide node `node` (traverse.hpp layout): the children's boxes quantised to 8 bits per plane in
// the frame of their union.  origin = union min; per axis scale = 2^e with 254 * scale >= extent; child plane =
// fma(q, scale, origin), the SAME expression the traversal evaluates, so the rounding checks below make the
// decoded box contain the exact one.
VKR_DI void quantise_wide(float4* node, const float lo[4][3], const float hi[4][3], int ncand, const float ulo[3], const float uhi[3]);

VKR_DI void write_wide_boxes(float4* node, const int cand[4], int ncand, const float4* node_box, const float4* tris,
                             float pad_abs)
{
    float lo[4][3], hi[4][3];
    float ulo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, uhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < ncand) {
            cand_box(cand[k], node_box, tris, pad_abs, lo[k], hi[k]);
#pragma unroll
            for (int a = 0; a < 3; ++a) { ulo[a] = __builtin_fminf(ulo[a], lo[k][a]); uhi[a] = __builtin_fmaxf(uhi[a], hi[k][a]); }
        }
    }
    quantise_wide(node, lo, hi, ncand, ulo, uhi);
}

VKR_DI void quantise_wide(float4* node, const float lo[4][3], const float hi[4][3], int ncand, const float ulo[3], const float uhi[3])
{
    float scale[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float ext = (uhi[a] - ulo[a]) * (1.0f / 254.0f);
        // smallest power of two >= ext (>= 2^-100 so the exponent byte is never 0 and 1/scale stays finite)
        uint32_t e = (f2u(ext) >> 23) & 0xffu;
        if ((f2u(ext) & 0x7fffffu) != 0u) e += 1u;
        if (e < 27u) e = 27u;
        if (e > 254u) e = 254u;
        scale[a] = u2f(e << 23);
    }
    uint32_t q[6] = {0, 0, 0, 0, 0, 0};   // qminx, qminy, qminz, qmaxx, qmaxy, qmaxz : one byte per child
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            uint32_t ql = 255u, qh = 0u;   // unused slot: inverted, misses every slab test
            if (k < ncand) {
                float inv = 1.0f / scale[a];
                float fl = __builtin_floorf((lo[k][a] - ulo[a]) * inv);
                float fh = __builtin_ceilf((hi[k][a] - ulo[a]) * inv);
                int il = fl < 0.0f ? 0 : (fl > 255.0f ? 255 : (int)fl);
                int ih = fh < 0.0f ? 0 : (fh > 255.0f ? 255 : (int)fh);
                while (il > 0 && fma_((float)il, scale[a], ulo[a]) > lo[k][a]) --il;
                while (ih < 255 && fma_((float)ih, scale[a], ulo[a]) < hi[k][a]) ++ih;
                ql = (uint32_t)il;
                qh = (uint32_t)ih;
            }
            q[a] |= ql << (8 * k);
            q[3 + a] |= qh << (8 * k);
        }
    }
    node[0] = make_float4(ulo[0], ulo[1], ulo[2], scale[2]);
    node[1] = make_float4(u2f(q[0]), u2f(q[1]), u2f(q[2]), u2f(q[3]));
    node[2] = make_float4(u2f(q[4]), u2f(q[5]), scale[0], scale[1]);
}

#ifndef VKR_WIDE_SLOT_ORDER
#define VKR_WIDE_SLOT_ORDER 1   // 1: child slots in Morton order; 0: the expanded child's halves at its slot and at the end
#endif
__global__ __launch_bounds__(256) void k_wide_level(WideCtl* ctl, const int2* bin_child, const uint2* bin_range,
                                                    const float4* node_box, const float4* tris, float4* nodes,
                                                    uint4* wide_bref, uint32_t* wide_bin, const BvhBounds* bnd)
{
    // Child nodes of one level are numbered in the order of their parents (block-wide exclusive scan, one atomic per block;
    // blocks reach it roughly in launch order), and the children of a node in Morton order: nodes that are neighbours in
    // space are neighbours in memory on every level, so the node fetches of a wave of coherent rays fall into few
    // 128-byte lines (the L1 return path charges a wave-level load by the distinct lines it touches, tools/calib/l1_rate2).
    __shared__ uint32_t s_scan[256];
    __shared__ uint32_t s_base;
    const uint32_t begin = ctl->level_begin, end = ctl->level_end;
    if (begin + blockIdx.x * 256u >= end) return;             // block-uniform: the launch is sized for the widest level
    const uint32_t w = begin + blockIdx.x * 256u + threadIdx.x;
    const bool live = w < end;
    int cand[4] = {0, 0, 0, 0};
    int ncand = 0;
    auto tri_count = [&](int c) -> uint32_t {
        if (c < 0) return 1u;
        uint2 r = bin_range[c];
        return r.y - r.x + 1u;
    };
    if (live) {
        int b = (int)wide_bin[w];
        int2 c0 = bin_child[b];
        cand[0] = c0.x; cand[1] = c0.y;
        ncand = 2;
        for (int round = 0; round < 2; ++round) {
            int pick = -1;
            float best = -1.0f;
            for (int k = 0; k < ncand; ++k) {
                int c = cand[k];
                if (c < 0 || tri_count(c) <= (uint32_t)VKR_LEAF_MAX) continue;
#ifdef VKR_COLLAPSE_BY_COUNT
                float a = (float)tri_count(c);
#else
                float a = half_area(node_box[(size_t)c * 2], node_box[(size_t)c * 2 + 1]);
#endif
                if (a > best) { best = a; pick = k; }
            }
            if (pick < 0) break;
            int2 c = bin_child[cand[pick]];
#if VKR_WIDE_SLOT_ORDER
            for (int k = ncand; k > pick + 1; --k) cand[k] = cand[k - 1];     // keep the slots in Morton (left to right) order
            cand[pick] = c.x;
            cand[pick + 1] = c.y;
            ++ncand;
#else
            cand[pick] = c.x;                                                 // round-1 order: the split child's right half goes last
            cand[ncand++] = c.y;
#endif
        }
    }
    uint32_t nint = 0;
    for (int k = 0; k < ncand; ++k) nint += (cand[k] >= 0 && tri_count(cand[k]) > (uint32_t)VKR_LEAF_MAX) ? 1u : 0u;
    s_scan[threadIdx.x] = nint;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {
        uint32_t v = threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0u;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    if (threadIdx.x == 255u) s_base = atomicAdd(&ctl->count, s_scan[255]);
    __syncthreads();
    if (!live) return;
    uint32_t next = s_base + s_scan[threadIdx.x] - nint;
    float4* node = nodes + (size_t)w * 4;
    write_wide_boxes(node, cand, ncand, node_box, tris, pad_abs_of(bnd));
    uint32_t ref[4], bref[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ref[k] = VKR_CHILD_NONE;
        bref[k] = VKR_CHILD_NONE;
        if (k < ncand) {
            int c = cand[k];
            bref[k] = (uint32_t)c;
            uint32_t cnt = tri_count(c);
            if (c >= 0 && cnt > (uint32_t)VKR_LEAF_MAX) {
                uint32_t idx = next++;
                wide_bin[idx] = (uint32_t)c;
                ref[k] = idx;
            } else {
                uint32_t first = (c >= 0) ? bin_range[c].x : (uint32_t)(~c);
                ref[k] = leaf_ref_make(first, cnt);
            }
        }
    }
    node[3] = make_float4(u2f(ref[0]), u2f(ref[1]), u2f(ref[2]), u2f(ref[3]));
    wide_bref[w] = make_uint4(bref[0], bref[1], bref[2], bref[3]);
}

__global__ void k_wide_init(WideCtl* ctl, uint32_t* wide_bin)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ctl->count = 1; ctl->level_begin = 0; ctl->level_end = 1; ctl->levels = 0;
        ctl->level_start[0] = 0;
        wide_bin[0] = 0;
    }
}

__global__ void k_wide_next(WideCtl* ctl)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (ctl->level_end > ctl->level_begin) {
            if (ctl->levels < 71u) { ctl->level_start[ctl->levels] = ctl->level_begin; ctl->level_start[ctl->levels + 1u] = ctl->level_end; }
            ctl->levels += 1;
        }
        ctl->level_begin = ctl->level_end;
        ctl->level_end = ctl->count;
    }
}

// refit: same topology, new boxes
__global__ __launch_bounds__(256) void k_wide_refit(uint32_t count, const float4* node_box, const float4* tris,
                                                    float4* nodes, const uint4* wide_bref, const BvhBounds* bnd)
{
    uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= count) return;
    uint4 br = wide_bref[w];
    int cand[4] = {(int)br.x, (int)br.y, (int)br.z, (int)br.w};
    int ncand = 0;
    if (br.x != VKR_CHILD_NONE) ncand = 1;
    if (br.y != VKR_CHILD_NONE) ncand = 2;
    if (br.z != VKR_CHILD_NONE) ncand = 3;
    if (br.w != VKR_CHILD_NONE) ncand = 4;
    write_wide_boxes(nodes + (size_t)w * 4, cand, ncand, node_box, tris, pad_abs_of(bnd));
}

// Refit on the wide tree itself, one launch per level from the deepest up (levels are contiguous index ranges): a leaf
// child's box is the union of its triangles' padded boxes, an internal child's box is the one its own launch stored.
// min / max are exact, so these are the boxes the binary fit produces (the same sets of triangles) and the nodes come
// out bit-identical to k_fit + k_wide_refit -- without the 39-deep chain of atomic hand-offs of the binary fit.
VKR_DI void wide_refit_node(uint32_t w, float4* nodes, float4* wide_box, const float4* tris, const BvhBounds* bnd)
{
    float4* node = nodes + (size_t)w * 4;
    const float4 rf = node[3];
    const uint32_t ref[4] = {f2u(rf.x), f2u(rf.y), f2u(rf.z), f2u(rf.w)};
    const float pad_abs = pad_abs_of(bnd);
    float lo[4][3], hi[4][3];
    float ulo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, uhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    int ncand = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (ref[k] == VKR_CHILD_NONE) continue;     // used slots are a prefix
        ncand = k + 1;
        if ((int)ref[k] < 0) {
            const uint32_t first = leaf_ref_first(ref[k]), cnt = leaf_ref_count(ref[k]);
            tri_box(tris, first, pad_abs, lo[k], hi[k]);
            for (uint32_t j = 1; j < cnt; ++j) {
                float l2[3], h2[3];
                tri_box(tris, first + j, pad_abs, l2, h2);
#pragma unroll
                for (int a = 0; a < 3; ++a) { lo[k][a] = __builtin_fminf(lo[k][a], l2[a]); hi[k][a] = __builtin_fmaxf(hi[k][a], h2[a]); }
            }
        } else {
            const float4 mn = wide_box[(size_t)ref[k] * 2], mx = wide_box[(size_t)ref[k] * 2 + 1];
            lo[k][0] = mn.x; lo[k][1] = mn.y; lo[k][2] = mn.z; hi[k][0] = mx.x; hi[k][1] = mx.y; hi[k][2] = mx.z;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { ulo[a] = __builtin_fminf(ulo[a], lo[k][a]); uhi[a] = __builtin_fmaxf(uhi[a], hi[k][a]); }
    }
    quantise_wide(node, lo, hi, ncand, ulo, uhi);
    wide_box[(size_t)w * 2] = make_float4(ulo[0], ulo[1], ulo[2], 0.0f);
    wide_box[(size_t)w * 2 + 1] = make_float4(uhi[0], uhi[1], uhi[2], 0.0f);
}

__global__ __launch_bounds__(256) void k_wide_refit_level(uint32_t begin, uint32_t end, float4* nodes, float4* wide_box,
                                                          const float4* tris, const BvhBounds* bnd)
{
    const uint32_t w = begin + blockIdx.x * 256u + threadIdx.x;
    if (w < end) wide_refit_node(w, nodes, wide_box, tris, bnd);
}

// n == 1: a root with one single-triangle leaf
__global__ void k_single(const float4* tris, float4* nodes, uint4* wide_bref, uint32_t* height_out, WideCtl* ctl,
                         const BvhBounds* bnd)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int cand[4] = {~0, 0, 0, 0};
    write_wide_boxes(nodes, cand, 1, nullptr, tris, pad_abs_of(bnd));
    nodes[3] = make_float4(u2f(leaf_ref_make(0u, 1u)), u2f(VKR_CHILD_NONE), u2f(VKR_CHILD_NONE), u2f(VKR_CHILD_NONE));
    wide_bref[0] = make_uint4(~0u, VKR_CHILD_NONE, VKR_CHILD_NONE, VKR_CHILD_NONE);
    *height_out = 1u;
    ctl->count = 1; ctl->levels = 1; ctl->level_begin = 1; ctl->level_end = 1;
    ctl->level_start[0] = 0; ctl->level_start[1] = 1;
}

// ---------------------------------------------------------------- scene graph -> instance transforms ----
// Renderer::rebuildRayTracingInstances (Renderer.cpp:503-527) without the CPU walk: one thread per node climbs to
// its root, then multiplies parentWorld * local back down in the reference's order (GLM scalar mat4 product, every
// product and sum rounded -- identical bits to the DFS), and writes toVkTransformMatrix (RayTracingContext.cpp:28-44)
// for the node's instances.  O(nodes x depth); depth is checked against VKR_GRAPH_MAX_DEPTH on upload.
struct ModelMat { float m[16]; };
VKR_DI void glm_mul_m4(const float* A, const float* B, float* R)
{
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float acc = A[0 * 4 + r] * B[j * 4 + 0];
            acc = acc + A[1 * 4 + r] * B[j * 4 + 1];
            acc = acc + A[2 * 4 + r] * B[j * 4 + 2];
            acc = acc + A[3 * 4 + r] * B[j * 4 + 3];
            R[j * 4 + r] = acc;
        }
}

__global__ __launch_bounds__(64) void k_instances_from_nodes(const vkr_ctx::NodeDev* nodes, uint32_t node_count,
                                                             ModelMat model, float* inst_xf)
{
    const uint32_t n = blockIdx.x * 64u + threadIdx.x;
    if (n >= node_count) return;
    if (nodes[n].instance_count == 0u) return;         // inner nodes without meshes contribute through their children
    int chain[VKR_GRAPH_MAX_DEPTH];
    int depth = 0;
    for (int a = (int)n; a >= 0 && depth < VKR_GRAPH_MAX_DEPTH; a = nodes[a].parent) chain[depth++] = a;
    float world[16], next[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) world[k] = model.m[k];
    for (int d = depth - 1; d >= 0; --d) {
        glm_mul_m4(world, nodes[chain[d]].local, next);
#pragma unroll
        for (int k = 0; k < 16; ++k) world[k] = next[k];
    }
    for (uint32_t k = 0; k < nodes[n].instance_count; ++k) {
        float* o = inst_xf + (size_t)(nodes[n].first_instance + k) * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) o[r * 4 + c] = world[c * 4 + r];
    }
}

hipError_t instances_from_nodes_launch(vkr_ctx* c, const float model[16], hipStream_t st)
{
    if (c->graph_nodes == 0) return hipSuccess;
    ModelMat m;
    for (int k = 0; k < 16; ++k) m.m[k] = model[k];
    k_instances_from_nodes<<<(c->graph_nodes + 63u) / 64u, 64, 0, st>>>(c->d_graph.p, c->graph_nodes, m, c->d_inst_xf.p);
    return hipGetLastError();
}

#define VKR_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

// Development check (VKR_VALIDATE_TREE=1): the binary tree is well formed before anything walks it -- every internal
// node's children tile its range in order, leaves sit at their own position, parent links agree, node 0 is the root.
static hipError_t validate_binary_tree(vkr_ctx* c, hipStream_t st)
{
    const uint32_t n = c->tri_count;
    if (n < 2) return hipSuccess;
    VKR_TRY(hipStreamSynchronize(st));
    std::vector<int2> child(n - 1);
    std::vector<uint2> range(n - 1);
    std::vector<uint32_t> lpar(n), npar(n - 1), perm(n);
    VKR_TRY(hipMemcpy(child.data(), c->d_bin_child.p, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost));
    VKR_TRY(hipMemcpy(range.data(), c->d_bin_range.p, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost));
    VKR_TRY(hipMemcpy(lpar.data(), c->d_leaf_parent.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    VKR_TRY(hipMemcpy(npar.data(), c->d_node_parent.p, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost));
    VKR_TRY(hipMemcpy(perm.data(), c->d_perm.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    auto bad = [&](const char* what, uint32_t i) {
        c->err = std::string("binary tree check failed: ") + what + " at node " + std::to_string(i);
        fprintf(stderr, "[vkr] %s\n", c->err.c_str());
        return hipErrorUnknown;
    };
    if (range[0].x != 0u || range[0].y != n - 1u || npar[0] != 0xffffffffu) return bad("root range/parent", 0);
    std::vector<uint8_t> seen(n - 1, 0);
    std::vector<uint32_t> stack{0u};
    size_t visited = 0;
    while (!stack.empty()) {
        const uint32_t i = stack.back(); stack.pop_back();
        if (i >= n - 1 || seen[i]) return bad("node index / revisit", i);
        seen[i] = 1; ++visited;
        const uint32_t lo = range[i].x, hi = range[i].y;
        if (lo >= hi || hi >= n) return bad("range", i);
        uint32_t at = lo;
        const int ch[2] = {child[i].x, child[i].y};
        for (int k = 0; k < 2; ++k) {
            if (ch[k] < 0) {
                const uint32_t leaf = (uint32_t)~ch[k];
                if (leaf != at) return bad("leaf position", i);
                if (lpar[leaf] != i) return bad("leaf parent", i);
                at += 1;
            } else {
                const uint32_t j = (uint32_t)ch[k];
                if (j >= n - 1) return bad("child index", i);
                if (range[j].x != at || range[j].y > hi) return bad("child range", i);
                if (npar[j] != i) return bad("node parent", i);
                at = range[j].y + 1;
                stack.push_back(j);
            }
        }
        if (at != hi + 1) return bad("children do not tile the range", i);
    }
    if (visited != (size_t)n - 1) return bad("unreachable nodes", (uint32_t)visited);
    std::vector<uint8_t> hit(n, 0);
    for (uint32_t i = 0; i < n; ++i) { if (perm[i] >= n || hit[perm[i]]) return bad("perm is not a permutation", i); hit[perm[i]] = 1; }
    return hipSuccess;
}

// Instance of every triangle of the final order (inst_first_tri[i] <= flat id < inst_first_tri[i + 1]): looked up once per
// build here instead of once per pixel per frame by the G-buffer producer (a refit keeps order and membership).
__global__ __launch_bounds__(256) void k_tri_instance(const float4* tris, uint32_t n, const uint32_t* inst_first_tri, uint32_t instance_count,
                                                      uint32_t* tri_inst)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t flat = f2u(tris[(size_t)i * 3 + 2].w);
    uint32_t lo = 0, hi = instance_count;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (inst_first_tri[mid] <= flat) lo = mid; else hi = mid;
    }
    tri_inst[i] = lo;
}

// Stage 1 of a build (everything up to the binary fit); the host reads back the binary height, then stage 2
// collapses to wide nodes.  refit_only: stage 1 re-transforms + refits, stage 2 regathers wide boxes.
hipError_t bvh_build_launch(vkr_ctx* c, bool refit_only, hipStream_t st)
{
    const uint32_t n = c->tri_count;
    if (n == 0) return hipSuccess;
    BuildIn in;
    in.positions = c->d_positions.p; in.indices = c->d_indices.p; in.lut = c->d_lut.p;
    in.mesh_class = c->d_mesh_class.p; in.inst_mesh = c->d_inst_mesh.p; in.inst_first_tri = c->d_inst_first_tri.p;
    in.inst_xf = c->d_inst_xf.p; in.instance_count = c->instance_count; in.tri_count = n;
    const uint32_t nb = (n + 255u) / 256u;

    k_bounds_init<<<1, 64, 0, st>>>(c->d_bounds.p);   // also on refit: transforms move the scene, and the conservative
    if (!refit_only)                                  // padding scales with its extent (refit: reduced in k_emit_sorted)
        k_bounds<<<nb < 1024u ? nb : 1024u, 256, 0, st>>>(in, c->d_bounds.p);
    if (!refit_only) {
        const int emc = c->morton_emc ? 1 : 0;
        const uint32_t mbits = emc ? (c->morton_bits > 15u ? 15u : c->morton_bits) : c->morton_bits;
        k_morton<<<nb, 256, 0, st>>>(in, c->d_bounds.p, mbits, emc, c->d_keys[0].p, c->d_vals[0].p);
        const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
        const int key_bits = (emc ? 4 : 3) * (int)mbits;
        int cur = 0;
        for (int shift = 0; shift < key_bits; shift += 8) {
            k_rs_hist<<<ntiles, RS_THREADS, 0, st>>>(c->d_keys[cur].p, n, shift, ntiles, c->d_hist.p);
            k_rs_scan<<<256, RS_THREADS, 0, st>>>(c->d_hist.p, ntiles, c->d_digit_totals.p);
            k_rs_scatter<<<ntiles, RS_THREADS, 0, st>>>(c->d_keys[cur].p, c->d_vals[cur].p, c->d_keys[cur ^ 1].p,
                                                        c->d_vals[cur ^ 1].p, n, shift, ntiles, c->d_hist.p,
                                                        c->d_digit_totals.p);
            cur ^= 1;
        }
        if (c->sah_top && n > (uint32_t)SAH_LEAF) {
            const size_t cap = (size_t)n / SAH_LEAF + 2;                       // open ranges per level: each holds > SAH_LEAF triangles
            const size_t w_depth = ((size_t)n + 3) / 4, w_nodes = cap * (sizeof(SahNode) / 4);
            const size_t words = 2 * (size_t)n + w_depth + 64 + 2 * w_nodes + cap * SAH_BIN_WORDS;
            if (c->d_sah.n < words) VKR_TRY(c->d_sah.alloc(words));
            uint32_t* tnode = c->d_sah.p;
            uint32_t* path = tnode + n;
            uint8_t* depth = (uint8_t*)(path + n);
            uint32_t* level_count = path + n + w_depth;                        // [SAH_DEPTH + 1] used
            SahNode* nodes[2] = {(SahNode*)(level_count + 64), (SahNode*)(level_count + 64 + w_nodes)};
            uint32_t* bins = level_count + 64 + 2 * w_nodes;
            float4* box = (float4*)c->d_deposit.p;                             // 2 float4 per triangle; the fit's slots are free until k_fit
            VKR_TRY(hipMemsetAsync(level_count, 0, 64 * 4, st));
            k_sah_prep<<<nb, 256, 0, st>>>(in, c->d_vals[cur].p, box, tnode, path, depth, nodes[0], c->d_bounds.p, level_count);
            for (int level = 0; level < SAH_DEPTH; ++level) {
                VKR_TRY(hipMemsetAsync(bins, 0, cap * SAH_BIN_WORDS * 4, st));
                k_sah_bin<<<(n + SAH_CHUNK - 1) / SAH_CHUNK, 256, 0, st>>>(n, tnode, box, nodes[level & 1], level_count + level, bins);
                k_sah_split<<<(uint32_t)((cap + 63) / 64), 64, 0, st>>>(nodes[level & 1], nodes[(level + 1) & 1], bins, level_count + level, level);
                k_sah_assign<<<nb, 256, 0, st>>>(n, tnode, box, path, depth, nodes[level & 1], level_count + level);
                // every fourth level the host looks at the number of ranges the next level opens and stops launching once there
                // is none (the scene decides: 16 levels for the bistro-class one, not SAH_DEPTH = 32)
                if ((level & 3) == 3 && level + 1 < SAH_DEPTH) {
                    uint32_t open_next = 1u;
                    VKR_TRY(hipMemcpyAsync(&open_next, level_count + level + 1, 4, hipMemcpyDeviceToHost, st));
                    VKR_TRY(hipStreamSynchronize(st));
                    if (open_next == 0u) break;
                }
            }
            k_sah_rekey<<<nb, 256, 0, st>>>(n, c->d_keys[cur].p, path, depth, key_bits);
            for (int shift = 32; shift < 64; shift += 8) {                     // stable: the Morton order survives inside equal paths
                k_rs_hist<<<ntiles, RS_THREADS, 0, st>>>(c->d_keys[cur].p, n, shift, ntiles, c->d_hist.p);
                k_rs_scan<<<256, RS_THREADS, 0, st>>>(c->d_hist.p, ntiles, c->d_digit_totals.p);
                k_rs_scatter<<<ntiles, RS_THREADS, 0, st>>>(c->d_keys[cur].p, c->d_vals[cur].p, c->d_keys[cur ^ 1].p,
                                                            c->d_vals[cur ^ 1].p, n, shift, ntiles, c->d_hist.p,
                                                            c->d_digit_totals.p);
                cur ^= 1;
            }
        }
        VKR_TRY(hipMemcpyAsync(c->d_perm.p, c->d_vals[cur].p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        if (n > 1)
            k_hierarchy<<<(n - 1 + 255u) / 256u, 256, 0, st>>>(c->d_keys[cur].p, (int)n, c->d_bin_child.p,
                                                             c->d_bin_range.p, c->d_leaf_parent.p, c->d_node_parent.p);
    }
    if (refit_only) k_emit_sorted<true><<<nb, 256, 0, st>>>(in, c->d_perm.p, c->d_tris.p, c->d_bounds.p);
    else k_emit_sorted<false><<<nb, 256, 0, st>>>(in, c->d_perm.p, c->d_tris.p, c->d_bounds.p);
    if (!refit_only && n >= (uint32_t)TL_MIN && c->treelets) {
        if (c->d_treelet.n < (size_t)n + 2) VKR_TRY(c->d_treelet.alloc((size_t)n + 2));
        VKR_TRY(hipMemsetAsync(c->d_treelet.p, 0, 8, st));
        k_treelet_roots<<<(n - 1 + 255u) / 256u, 256, 0, st>>>(c->d_bin_range.p, c->d_node_parent.p, (int)n - 1, c->d_treelet.p);
        const uint32_t blocks = (uint32_t)(c->num_cus > 0 ? c->num_cus : 256) * (TL_MAX <= 256 ? 6u : (TL_MAX <= 512 ? 3u : 2u));
        k_treelet_ploc<<<blocks, TL_MAX, 0, st>>>(c->d_treelet.p, c->d_bin_child.p, c->d_bin_range.p, c->d_leaf_parent.p,
                                               c->d_node_parent.p, c->d_tris.p, c->d_perm.p);
    }
    if (!refit_only)   // the triangle order is final here; a refit keeps it
        k_alpha_records<<<nb, 256, 0, st>>>(c->d_tris.p, n, c->d_lut.p, c->d_indices.p, c->d_uvs.p, c->d_material_params.p,
                                              c->d_alpha_tex.p, c->d_tri_alpha.p);
    if (!refit_only)   // ... and so is the instance every triangle came from (the G-buffer producer's normal matrix)
        k_tri_instance<<<nb, 256, 0, st>>>(c->d_tris.p, n, c->d_inst_first_tri.p, c->instance_count, c->d_tri_inst.p);
    if (!refit_only && c->validate_tree) VKR_TRY(validate_binary_tree(c, st));
    if (n == 1) {
        k_single<<<1, 64, 0, st>>>(c->d_tris.p, c->d_nodes.p, c->d_wide_bref.p, c->d_height.p, (WideCtl*)c->d_wide_ctl.p,
                                   c->d_bounds.p);
        return hipGetLastError();
    }
    if (refit_only && c->refit_wide && c->wide_level_start.size() == (size_t)c->wide_levels + 1u && c->wide_levels > 0u) {
        // wide boxes live in d_node_box (the binary boxes there are only needed by the collapse of a build)
        // one launch per level (batching the small levels into single-block launches with barriers was slower: 0.68 ms)
        for (uint32_t l = c->wide_levels; l-- > 0u;) {
            const uint32_t b = c->wide_level_start[l], e = c->wide_level_start[l + 1u];
            if (e > b)
                k_wide_refit_level<<<(e - b + 255u) / 256u, 256, 0, st>>>(b, e, c->d_nodes.p, c->d_node_box.p, c->d_tris.p, c->d_bounds.p);
        }
        return hipGetLastError();
    }
    VKR_TRY(hipMemsetAsync(c->d_flags.p, 0, (size_t)(n - 1) * 4, st));
    k_fit<<<nb, 256, 0, st>>>(c->d_tris.p, (int)n, c->d_bin_child.p, (u64*)c->d_deposit.p, c->d_node_box.p,
                              c->d_leaf_parent.p, c->d_node_parent.p, c->d_flags.p, c->d_height.p, c->d_bounds.p);
    if (refit_only) {
        k_wide_refit<<<(c->wide_count + 255u) / 256u, 256, 0, st>>>(c->wide_count, c->d_node_box.p, c->d_tris.p,
                                                                    c->d_nodes.p, c->d_wide_bref.p, c->d_bounds.p);
    }
    return hipGetLastError();
}

// Stage 2: BFS collapse; `bin_height` bounds the number of levels.
hipError_t bvh_wide_launch(vkr_ctx* c, uint32_t bin_height, hipStream_t st)
{
    const uint32_t n = c->tri_count;
    if (n <= 1) return hipSuccess;
    WideCtl* ctl = (WideCtl*)c->d_wide_ctl.p;
    k_wide_init<<<1, 64, 0, st>>>(ctl, c->d_wide_bin.p);
    uint64_t cap = 1;
    for (uint32_t level = 0; level < bin_height + 1; ++level) {
        uint64_t width = cap < (uint64_t)n ? cap : (uint64_t)n;   // a level never holds more than 4^level nodes
        k_wide_level<<<(uint32_t)((width + 255u) / 256u), 256, 0, st>>>(ctl, c->d_bin_child.p, c->d_bin_range.p,
                                                                        c->d_node_box.p, c->d_tris.p, c->d_nodes.p,
                                                                        c->d_wide_bref.p, c->d_wide_bin.p, c->d_bounds.p);
        k_wide_next<<<1, 64, 0, st>>>(ctl);
        if (cap < (uint64_t)n) cap *= 4;
        // the binary height only bounds the number of levels (39 for the bistro-class scene, whose wide tree has 18): every
        // eighth level the host reads the control block and stops once the next level is empty
        if ((level & 7u) == 7u && level + 1u < bin_height + 1u) {
            uint32_t h[4] = {0, 0, 1, 0};                              // count, level_begin, level_end, levels
            VKR_TRY(hipMemcpyAsync(h, c->d_wide_ctl.p, 16, hipMemcpyDeviceToHost, st));
            VKR_TRY(hipStreamSynchronize(st));
            if (h[1] == h[2]) break;
        }
    }
    return hipGetLastError();
}

}  // namespace vkr
