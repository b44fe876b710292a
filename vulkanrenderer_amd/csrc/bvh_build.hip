// bvh_build.hip -- GPU LBVH build / refit.  Replaces the driver's acceleration-structure builder behind
// RayTracingContext::buildBottomLevelASes / buildTopLevelAS / updateTopLevelAS
// (RayTracingContext.cpp:199-345, 165-197).
//
// The reference is two-level (one BLAS per mesh + a TLAS of instances).  Here instances are flattened to
// world-space triangles (instanced meshes are duplicated) and ONE binary LBVH is built over them:
//   bounds -> Morton codes (21 or 10 bits/axis) -> LSD radix sort (8-bit digits, LDS ranking)
//   -> Karras 2012 hierarchy -> bottom-up AABB fit (atomic arrival flags).
// Refit = re-transform the triangles in the stored Morton order + the same bottom-up fit.
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
    uint32_t flat = blockIdx.x * 256u + threadIdx.x;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (flat < in.tri_count) {
        TriW t = emit_tri(in, flat);
        lo[0] = __builtin_fminf(t.v0.x, __builtin_fminf(t.v1.x, t.v2.x)); hi[0] = __builtin_fmaxf(t.v0.x, __builtin_fmaxf(t.v1.x, t.v2.x));
        lo[1] = __builtin_fminf(t.v0.y, __builtin_fminf(t.v1.y, t.v2.y)); hi[1] = __builtin_fmaxf(t.v0.y, __builtin_fmaxf(t.v1.y, t.v2.y));
        lo[2] = __builtin_fminf(t.v0.z, __builtin_fminf(t.v1.z, t.v2.z)); hi[2] = __builtin_fmaxf(t.v0.z, __builtin_fmaxf(t.v1.z, t.v2.z));
#pragma unroll
        for (int k = 0; k < 3; ++k) { clo[k] = (lo[k] + hi[k]) * 0.5f; chi[k] = clo[k]; }
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

__global__ __launch_bounds__(256) void k_morton(BuildIn in, const BvhBounds* bnd, uint32_t bits, uint64_t* keys,
                                                uint32_t* vals)
{
    uint32_t flat = blockIdx.x * 256u + threadIdx.x;
    if (flat >= in.tri_count) return;
    TriW t = emit_tri(in, flat);
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
    keys[flat] = (expand3(q[0]) << 2) | (expand3(q[1]) << 1) | expand3(q[2]);
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
__global__ __launch_bounds__(256) void k_emit_sorted(BuildIn in, const uint32_t* perm, float4* tris)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= in.tri_count) return;
    uint32_t flat = perm[i];
    TriW t = emit_tri(in, flat);
    V3 e1 = t.v1 - t.v0, e2 = t.v2 - t.v0;
    uint32_t cls = in.mesh_class[t.mesh];
    float4* o = tris + (size_t)i * 3;
    o[0] = make_float4(t.v0.x, t.v0.y, t.v0.z, u2f((t.mesh & 0x0fffffffu) | (cls << 28)));
    o[1] = make_float4(e1.x, e1.y, e1.z, u2f(t.prim));
    o[2] = make_float4(e2.x, e2.y, e2.z, u2f(flat));
}

// ---------------------------------------------------------------- Karras 2012 hierarchy ------------------
VKR_DI int delta(const uint64_t* keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}

__global__ __launch_bounds__(256) void k_hierarchy(const uint64_t* keys, int n, float* nodes, uint32_t* leaf_parent,
                                                   uint32_t* node_parent)
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
    float* nd = nodes + (size_t)i * 16;
    nd[12] = u2f((uint32_t)left);
    nd[13] = u2f((uint32_t)right);
    if (i == 0) node_parent[0] = 0xffffffffu;
}

// ---------------------------------------------------------------- bottom-up fit --------------------------
// One thread per leaf walks towards the root; at each node the first arrival stores its child box and stops,
// the second arrival (which sees the sibling's box after an agent-scope fence) continues with the union.
// flags[] must be zero on entry.
__global__ __launch_bounds__(256) void k_fit(const float4* tris, int n, float* nodes, const uint32_t* leaf_parent,
                                             const uint32_t* node_parent, uint32_t* flags, uint32_t* height_out)
{
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float4* tp = tris + (size_t)j * 3;
    float4 a = tp[0], b = tp[1], c = tp[2];
    float lo[3], hi[3];
    {
        float v0[3] = {a.x, a.y, a.z};
        float v1[3] = {a.x + b.x, a.y + b.y, a.z + b.z};
        float v2[3] = {a.x + c.x, a.y + c.y, a.z + c.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float l = __builtin_fminf(v0[k], __builtin_fminf(v1[k], v2[k]));
            float h = __builtin_fmaxf(v0[k], __builtin_fmaxf(v1[k], v2[k]));
            float pad = 1e-5f * __builtin_fmaxf(__builtin_fabsf(l), __builtin_fabsf(h)) + 1e-6f;
            lo[k] = l - pad;
            hi[k] = h + pad;
        }
    }
    uint32_t child = (uint32_t)(~j);
    uint32_t p = leaf_parent[j];
    uint32_t h = 0;  // height of the subtree rooted at `child`, in internal nodes
    while (true) {
        float* nd = nodes + (size_t)p * 16;
        bool is_left = (f2u(nd[12]) == child);
        float* slot = nd + (is_left ? 0 : 6);
#pragma unroll
        for (int k = 0; k < 3; ++k) { slot[k] = lo[k]; slot[3 + k] = hi[k]; }
        nd[is_left ? 14 : 15] = u2f(h);
        __threadfence();
        uint32_t old = atomicAdd(&flags[p], 1u);
        if (old == 0u) return;
        __threadfence();
        const volatile float* sib = nd + (is_left ? 6 : 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = __builtin_fminf(lo[k], sib[k]);
            hi[k] = __builtin_fmaxf(hi[k], sib[3 + k]);
        }
        uint32_t hs = f2u(((const volatile float*)nd)[is_left ? 15 : 14]);
        h = (h > hs ? h : hs) + 1u;
        if (p == 0u) { *height_out = h; return; }
        child = p;
        p = node_parent[p];
    }
}

// n == 1: a root whose two children are the same leaf
__global__ void k_single(const float4* tris, float* nodes, uint32_t* height_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float4 a = tris[0], b = tris[1], c = tris[2];
    float v0[3] = {a.x, a.y, a.z}, v1[3] = {a.x + b.x, a.y + b.y, a.z + b.z}, v2[3] = {a.x + c.x, a.y + c.y, a.z + c.z};
    for (int k = 0; k < 3; ++k) {
        float l = __builtin_fminf(v0[k], __builtin_fminf(v1[k], v2[k]));
        float h = __builtin_fmaxf(v0[k], __builtin_fmaxf(v1[k], v2[k]));
        float pad = 1e-5f * __builtin_fmaxf(__builtin_fabsf(l), __builtin_fabsf(h)) + 1e-6f;
        nodes[k] = l - pad; nodes[3 + k] = h + pad; nodes[6 + k] = l - pad; nodes[9 + k] = h + pad;
    }
    nodes[12] = u2f(~0u); nodes[13] = u2f(~0u); nodes[14] = 0.0f; nodes[15] = 0.0f;
    *height_out = 1u;
}

#define VKR_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t bvh_build_launch(vkr_ctx* c, bool refit_only, hipStream_t st)
{
    const uint32_t n = c->tri_count;
    if (n == 0) return hipSuccess;
    BuildIn in;
    in.positions = c->d_positions.p; in.indices = c->d_indices.p; in.lut = c->d_lut.p;
    in.mesh_class = c->d_mesh_class.p; in.inst_mesh = c->d_inst_mesh.p; in.inst_first_tri = c->d_inst_first_tri.p;
    in.inst_xf = c->d_inst_xf.p; in.instance_count = c->instance_count; in.tri_count = n;
    const uint32_t nb = (n + 255u) / 256u;
    float* nodes = (float*)c->d_nodes.p;

    if (!refit_only) {
        k_bounds_init<<<1, 64, 0, st>>>(c->d_bounds.p);
        k_bounds<<<nb, 256, 0, st>>>(in, c->d_bounds.p);
        k_morton<<<nb, 256, 0, st>>>(in, c->d_bounds.p, c->morton_bits, c->d_keys[0].p, c->d_vals[0].p);
        const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
        const int key_bits = 3 * (int)c->morton_bits;
        int cur = 0;
        for (int shift = 0; shift < key_bits; shift += 8) {
            k_rs_hist<<<ntiles, RS_THREADS, 0, st>>>(c->d_keys[cur].p, n, shift, ntiles, c->d_hist.p);
            k_rs_scan<<<256, RS_THREADS, 0, st>>>(c->d_hist.p, ntiles, c->d_digit_totals.p);
            k_rs_scatter<<<ntiles, RS_THREADS, 0, st>>>(c->d_keys[cur].p, c->d_vals[cur].p, c->d_keys[cur ^ 1].p,
                                                        c->d_vals[cur ^ 1].p, n, shift, ntiles, c->d_hist.p,
                                                        c->d_digit_totals.p);
            cur ^= 1;
        }
        VKR_TRY(hipMemcpyAsync(c->d_perm.p, c->d_vals[cur].p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        if (cur != 0)
            VKR_TRY(hipMemcpyAsync(c->d_keys[0].p, c->d_keys[1].p, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
        if (n > 1) {
            k_hierarchy<<<(n - 1 + 255u) / 256u, 256, 0, st>>>(c->d_keys[0].p, (int)n, nodes, c->d_leaf_parent.p,
                                                             c->d_node_parent.p);
        }
    }
    k_emit_sorted<<<nb, 256, 0, st>>>(in, c->d_perm.p, c->d_tris.p);
    if (n == 1) {
        k_single<<<1, 64, 0, st>>>(c->d_tris.p, nodes, c->d_height.p);
    } else {
        VKR_TRY(hipMemsetAsync(c->d_flags.p, 0, (size_t)(n - 1) * 4, st));
        k_fit<<<nb, 256, 0, st>>>(c->d_tris.p, (int)n, nodes, c->d_leaf_parent.p, c->d_node_parent.p, c->d_flags.p,
                                  c->d_height.p);
    }
    return hipGetLastError();
}

}  // namespace vkr
