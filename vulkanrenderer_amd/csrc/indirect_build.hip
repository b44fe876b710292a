// indirect_build.hip -- SURVEY 8(f) row 4: FrameManager::prepareSharedOpaqueIndirect (FrameManager.cpp:118-175) on
// the GPU: node world matrices, per-slot validity (+ optional frustum test), stream compaction into
// DrawIndexedIndirectCommand / per-draw model matrices, bucket spans.  HBM-bound integer/copy work: no MFMA, no LDS
// beyond the block scans.
//
// Why plain compaction reproduces the reference's std::map of buckets: buildSharedDrawSlots (:108-115) leaves the
// slots sorted by (doubleSided, matIndex, meshIndex), the bucket key is (doubleSided, matIndex) and std::map walks
// its keys in exactly that order, so the concatenation of the buckets is the slot order itself.  A surviving slot's
// command therefore lands at index drawId, the rank of the slot among the survivors.
#include "internal.hpp"

namespace vkr {

constexpr int IB_TILE = 2048;      // slots per scan tile (256 threads x 8)

struct DrawSlotDev { uint32_t node, mesh, mat, ds; };
struct MeshInfoDev { uint32_t vertex_offset, first_index, index_count; };
struct DrawCmdDev { uint32_t index_count, instance_count, first_index; int32_t vertex_offset; uint32_t first_instance; };
struct SpanDev { uint32_t ds, mat, first, count; };
struct Planes { float p[24]; };
struct Mat16 { float m[16]; };

VKR_DI void glm_mul_m4_ib(const float* A, const float* B, float* R)   // GLM scalar mat4 product, see bvh_build.hip
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

// sharedNodeWorldMatrices[n] = parentWorld * local (:121-126).  One thread per node climbs to its root and multiplies
// back down in the reference's association; depth is bounded by VKR_GRAPH_MAX_DEPTH (checked on upload).
__global__ __launch_bounds__(256) void k_ib_world(const int32_t* parent, const float4* local, uint32_t n_nodes, Mat16 model,
                                                  float4* world)
{
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= n_nodes) return;
    int chain[VKR_GRAPH_MAX_DEPTH];
    int depth = 0;
    for (int a = (int)n; a >= 0 && depth < VKR_GRAPH_MAX_DEPTH; a = parent[a]) chain[depth++] = a;
    float w[16], nx[16], l[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = model.m[k];
    for (int d = depth - 1; d >= 0; --d) {
        const float4* lp = local + (size_t)chain[d] * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) { float4 v = lp[c]; l[4 * c] = v.x; l[4 * c + 1] = v.y; l[4 * c + 2] = v.z; l[4 * c + 3] = v.w; }
        glm_mul_m4_ib(w, l, nx);
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = nx[k];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) world[(size_t)n * 4 + c] = make_float4(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
}

// The same product one hierarchy level per launch (levels known from the upload): parents' matrices are final when a
// level starts, so each node is one 64 B read of its parent's matrix, one of its local matrix and one 64 B write.
// Four lanes per node, one result column each: loads and stores are 16 B per lane and contiguous per node.
__global__ __launch_bounds__(256) void k_ib_world_level(const int32_t* parent, const float4* local, const uint32_t* order,
                                                        uint32_t first, uint32_t count, Mat16 model, float4* world)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if ((t >> 2) >= count) return;
    const uint32_t n = order[first + (t >> 2)], j = t & 3u;
    const int32_t p = parent[n];
    float4 a0, a1, a2, a3;
    if (p >= 0) {
        const float4* pw = world + (size_t)p * 4;
        a0 = pw[0]; a1 = pw[1]; a2 = pw[2]; a3 = pw[3];
    } else {
        a0 = make_float4(model.m[0], model.m[1], model.m[2], model.m[3]);
        a1 = make_float4(model.m[4], model.m[5], model.m[6], model.m[7]);
        a2 = make_float4(model.m[8], model.m[9], model.m[10], model.m[11]);
        a3 = make_float4(model.m[12], model.m[13], model.m[14], model.m[15]);
    }
    const float4 b = local[(size_t)n * 4 + j];
    float4 r;   // column j of A * B, GLM's order: ((A0*b.x + A1*b.y) + A2*b.z) + A3*b.w
    r.x = ((a0.x * b.x + a1.x * b.y) + a2.x * b.z) + a3.x * b.w;
    r.y = ((a0.y * b.x + a1.y * b.y) + a2.y * b.z) + a3.y * b.w;
    r.z = ((a0.z * b.x + a1.z * b.y) + a2.z * b.z) + a3.z * b.w;
    r.w = ((a0.w * b.x + a1.w * b.y) + a2.w * b.z) + a3.w * b.w;
    world[(size_t)n * 4 + j] = r;
}

// BoundingBox::Transform (BoundingBox.h:14-39) + Frustum::Intersects (Frustum.h:18-44)
VKR_DI bool box_in_frustum(const float4* wm, const float* b, const Planes& pl)
{
    const float4 c0 = wm[0], c1 = wm[1], c2 = wm[2], c3 = wm[3];
    const float m0[3] = {c0.x, c0.y, c0.z}, m1[3] = {c1.x, c1.y, c1.z}, m2[3] = {c2.x, c2.y, c2.z}, m3[3] = {c3.x, c3.y, c3.z};
    float nmin[3] = {1e30f, 1e30f, 1e30f}, nmax[3] = {-1e30f, -1e30f, -1e30f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float cx = (i & 1) ? b[3] : b[0], cy = (i & 2) ? b[4] : b[1], cz = (i & 4) ? b[5] : b[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float p = (m0[r] * cx + m1[r] * cy) + (m2[r] * cz + m3[r] * 1.0f);   // glm mat4 * vec4
            nmin[r] = p < nmin[r] ? p : nmin[r];
            nmax[r] = nmax[r] < p ? p : nmax[r];
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float* q = pl.p + 4 * k;
        const float vx = q[0] >= 0.0f ? nmax[0] : nmin[0], vy = q[1] >= 0.0f ? nmax[1] : nmin[1], vz = q[2] >= 0.0f ? nmax[2] : nmin[2];
        if (((q[0] * vx + q[1] * vy) + q[2] * vz) + q[3] < 0.0f) return false;
    }
    return true;
}

VKR_DI bool slot_survives(const DrawSlotDev& s, uint32_t n_nodes, uint32_t n_meshes, const float4* world, const float* bounds,
                          bool cull, const Planes& pl)
{
    if (s.node >= n_nodes || s.mesh >= n_meshes) return false;                    // :140
    if (cull && !box_in_frustum(world + (size_t)s.node * 4, bounds + 6 * (size_t)s.mesh, pl)) return false;
    return true;
}

// pass 1: survivors per tile
__global__ __launch_bounds__(256) void k_ib_count(const DrawSlotDev* slots, uint32_t n_slots, uint32_t n_nodes, uint32_t n_meshes,
                                                  const float4* world, const float* bounds, int cull, Planes pl,
                                                  uint32_t* tile_count, uint8_t* flags)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (int k = 0; k < IB_TILE / 256; ++k) {
        const uint32_t i = blockIdx.x * IB_TILE + k * 256 + threadIdx.x;
        bool ok = false;
        if (i < n_slots) {
            ok = slot_survives(slots[i], n_nodes, n_meshes, world, bounds, cull != 0, pl);
            flags[i] = ok ? 1 : 0;
        }
        mine += (uint32_t)__popcll(__ballot(ok)) * ((threadIdx.x & 63) == 0 ? 1u : 0u);
    }
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = s_cnt;
}

// pass 2: exclusive scan of the tile counts (one block; tiles <= 2^24 / 2048 = 8192)
__global__ __launch_bounds__(1024) void k_ib_scan_tiles(uint32_t* tile_count, uint32_t n_tiles, uint32_t* total)
{
    __shared__ uint32_t s_part[1024];
    const uint32_t per = (n_tiles + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * per, e = (b + per < n_tiles) ? b + per : n_tiles;
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; ++i) sum += tile_count[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (threadIdx.x >= (uint32_t)off) ? s_part[threadIdx.x - off] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
    for (uint32_t i = b; i < e; ++i) { uint32_t c = tile_count[i]; tile_count[i] = run; run += c; }
    if (threadIdx.x == 1023) *total = s_part[1023];
}

// pass 3: rank the survivors inside each tile and emit (:141-151)
__global__ __launch_bounds__(256) void k_ib_emit(const DrawSlotDev* slots, uint32_t n_slots, const uint8_t* flags,
                                                 const uint32_t* tile_base, const MeshInfoDev* infos, const float4* world,
                                                 uint32_t max_draws, float4* draw_data, DrawCmdDev* commands)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_run;
    if (threadIdx.x == 0) s_run = tile_base[blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < IB_TILE / 256; ++k) {
        const uint32_t i = blockIdx.x * IB_TILE + k * 256 + threadIdx.x;
        const bool ok = (i < n_slots) && flags[i];
        const unsigned long long m = __ballot(ok);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t base = s_run;
        for (int w = 0; w < wave; ++w) base += s_wave[w];
        const uint32_t id = base + mbcnt64(m);
        if (ok && id < max_draws) {                                               // :139 "break" = every later id is >= too
            const DrawSlotDev s = slots[i];
            const float4* wm = world + (size_t)s.node * 4;
            float4* dd = draw_data + (size_t)id * 4;
            dd[0] = wm[0]; dd[1] = wm[1]; dd[2] = wm[2]; dd[3] = wm[3];           // :141
            const MeshInfoDev info = infos[s.mesh];
            DrawCmdDev c;
            c.index_count = info.index_count; c.instance_count = 1; c.first_index = info.first_index;
            c.vertex_offset = (int32_t)info.vertex_offset; c.first_instance = id; // :143-148
            commands[id] = c;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
}

// bucket spans (:154-172).  The buckets are the runs of equal (doubleSided, matIndex) in the sorted slot list, known
// since the upload: seg_begin[s] = first slot of run s, seg_begin[n_segs] = n_slots.  The span of run s is
// [rank(seg_begin[s]), rank(seg_begin[s+1])) clipped to the draw count, with rank(x) = survivors before slot x --
// no atomics, no sorting.  One wave per boundary: rank = exclusive tile base + the tile's flag bytes before x.
__global__ __launch_bounds__(256) void k_ib_seg_rank(const uint32_t* seg_begin, uint32_t n_bounds, uint32_t n_slots,
                                                     const uint32_t* tile_base, const uint8_t* flags, const uint32_t* total,
                                                     uint32_t* seg_rank)
{
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= n_bounds) return;                                   // whole waves leave together
    const uint32_t x = seg_begin[b];
    uint32_t rank;
    if (x >= n_slots) rank = *total;
    else {
        const uint32_t t0 = (x / (uint32_t)IB_TILE) * (uint32_t)IB_TILE;
        // the tile's flag bytes (0/1) as 32-bit words: 8 independent loads per lane, bytes at or past x masked off
        const uint32_t* fw = (const uint32_t*)(flags + t0);      // t0 is a multiple of 2048 and flags is hipMalloc-aligned
        const uint32_t nb = x - t0;                              // bytes that count
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)IB_TILE / 256u; ++k) {
            const uint32_t w = k * 64u + (uint32_t)lane, byte0 = 4u * w;
            if (byte0 < nb) {
                const uint32_t keep = nb - byte0;                // 1..: bytes of this word that count
                uint32_t v = fw[w];
                if (keep < 4u) v &= (1u << (8u * keep)) - 1u;
                cnt += (uint32_t)__popc(v);
            }
        }
        cnt = wave_incl_scan(cnt);
        rank = tile_base[x / (uint32_t)IB_TILE] + (uint32_t)__builtin_amdgcn_readlane((int)cnt, 63);
    }
    if (lane == 0) seg_rank[b] = rank;
}

// one block: spans of the non-empty runs, in run order (:169-171 pushes only drawCount > 0)
__global__ __launch_bounds__(1024) void k_ib_span_emit(const uint32_t* seg_begin, const DrawSlotDev* slots, uint32_t n_segs,
                                                       const uint32_t* seg_rank, const uint32_t* total, uint32_t max_draws,
                                                       uint32_t* span_count, SpanDev* spans, uint32_t span_capacity)
{
    __shared__ uint32_t s_scan[1024];
    __shared__ uint32_t s_carry;
    const uint32_t n = *total < max_draws ? *total : max_draws;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_segs; base += 1024u) {
        const uint32_t sgm = base + threadIdx.x;
        uint32_t first = 0, count = 0;
        if (sgm < n_segs) {
            const uint32_t r0 = seg_rank[sgm], r1 = seg_rank[sgm + 1];
            first = r0 < n ? r0 : n;
            count = (r1 < n ? r1 : n) - first;
        }
        const uint32_t flag = count ? 1u : 0u;
        s_scan[threadIdx.x] = flag;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t v = (threadIdx.x >= (uint32_t)off) ? s_scan[threadIdx.x - off] : 0u;
            __syncthreads();
            s_scan[threadIdx.x] += v;
            __syncthreads();
        }
        const uint32_t pos = s_carry + s_scan[threadIdx.x] - flag;
        if (flag && pos < span_capacity) {
            const DrawSlotDev sl = slots[seg_begin[sgm]];
            SpanDev sp;
            sp.ds = sl.ds ? 1u : 0u; sp.mat = sl.mat; sp.first = first; sp.count = count;
            spans[pos] = sp;
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry += s_scan[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *span_count = s_carry;
}

#define IB_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t indirect_prepare_launch(vkr_ctx* c, const float model[16], const float* planes24, float4* draw_data,
                                   void* commands, hipStream_t st)
{
    const uint32_t ns = c->ib_slot_count, nn = c->ib_node_count;
    Mat16 m;
    for (int k = 0; k < 16; ++k) m.m[k] = model[k];
    Planes pl;
    for (int k = 0; k < 24; ++k) pl.p[k] = planes24 ? planes24[k] : 0.0f;
    IB_TRY(hipMemsetAsync(c->d_ib_counters.p, 0, 8, st));      // [0] total survivors, [1] span count
    if (nn && !c->ib_level_first.empty()) {
        for (size_t l = 0; l + 1 < c->ib_level_first.size(); ++l) {
            const uint32_t first = c->ib_level_first[l], cnt = c->ib_level_first[l + 1] - first;
            if (cnt) k_ib_world_level<<<(cnt * 4u + 255u) / 256u, 256, 0, st>>>(c->d_ib_parent.p, c->d_ib_local.p, c->d_ib_order.p,
                                                                              first, cnt, m, c->d_ib_world.p);
        }
    } else if (nn) {
        k_ib_world<<<(nn + 255u) / 256u, 256, 0, st>>>(c->d_ib_parent.p, c->d_ib_local.p, nn, m, c->d_ib_world.p);
    }
    if (ns == 0) return hipGetLastError();
    const uint32_t tiles = (ns + IB_TILE - 1) / IB_TILE;
    const int cull = (planes24 && c->d_ib_bounds.p) ? 1 : 0;
    k_ib_count<<<tiles, 256, 0, st>>>((const DrawSlotDev*)c->d_ib_slots.p, ns, nn, c->ib_mesh_count, c->d_ib_world.p,
                                      c->d_ib_bounds.p, cull, pl, c->d_ib_tile.p, c->d_ib_flags.p);
    k_ib_scan_tiles<<<1, 1024, 0, st>>>(c->d_ib_tile.p, tiles, c->d_ib_counters.p);
    k_ib_emit<<<tiles, 256, 0, st>>>((const DrawSlotDev*)c->d_ib_slots.p, ns, c->d_ib_flags.p, c->d_ib_tile.p,
                                     (const MeshInfoDev*)c->d_ib_infos.p, c->d_ib_world.p, c->ib_max_draws, draw_data,
                                     (DrawCmdDev*)commands);
    const uint32_t nseg = c->ib_seg_count;
    k_ib_seg_rank<<<(nseg + 1u + 3u) / 4u, 256, 0, st>>>(c->d_ib_seg_begin.p, nseg + 1u, ns, c->d_ib_tile.p, c->d_ib_flags.p,
                                                        c->d_ib_counters.p, c->d_ib_seg_rank.p);
    k_ib_span_emit<<<1, 1024, 0, st>>>(c->d_ib_seg_begin.p, (const DrawSlotDev*)c->d_ib_slots.p, nseg, c->d_ib_seg_rank.p,
                                       c->d_ib_counters.p, c->ib_max_draws, c->d_ib_counters.p + 1, (SpanDev*)c->d_ib_spans.p,
                                       c->ib_span_capacity);
    return hipGetLastError();
}

}  // namespace vkr
