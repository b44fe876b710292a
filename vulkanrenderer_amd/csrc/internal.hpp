// internal.hpp -- host-side context and launcher declarations shared by the translation units of libvkr_rt.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/vkr_rt.h"
#include "traverse.hpp"

namespace vkr {

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        release();
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        else p = nullptr;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// Per-frame constants of the RTAO kernels: PBRUniformBufferObject fields + host-hoisted inverses
// (rtao_trace_half.comp:245-246 computes inverse(proj)/inverse(view) per thread).
struct FrameParams {
    float invProj[16];
    float invView[16];
    float view[16];
    float prevViewProj[16];
    float camPos[3];
    uint32_t debugMode;
    int rayCount;
    float baseRadius;   // max(rtaoParams0.z, 0.01)
    float baseBias;     // max(rtaoParams0.w, 0.0005)
    float frameIndex;   // rtaoParams1.w
    float temporalAlphaRaw;  // rtaoParams1.y
    float disocclusionRaw;   // rtaoParams1.z
    float radiusRaw;         // rtaoParams0.z (A-Trous uses max(.,0.01) itself)
    int W, H;
    int stripeMul;      // K1 tile rows are 8-row stripes y0/8 + k*stripeMul (1 = contiguous rows; N = every N-th stripe)
};

// inverse of the ordered-uint float encoding used for atomicMin/atomicMax on bounds
__host__ __device__ inline float dec_f(uint32_t u)
{
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, b);
}

struct BvhBounds {  // ordered-uint encoded floats, updated with atomicMin/atomicMax
    uint32_t tri_min[3], tri_max[3], cen_min[3], cen_max[3];
};

}  // namespace vkr

struct vkr_ctx {
    int device = 0;
    uint32_t W = 0, H = 0;
    std::string err;

    // ---- scene (FrameManager::createReflectionBuffers + RayTracingContext inputs)
    bool has_scene = false;
    uint32_t mesh_count = 0, instance_count = 0, tri_count = 0;
    std::vector<uint32_t> h_inst_mesh;       // mesh index per instance (refit validation)
    std::vector<uint32_t> h_mesh_prims;      // primitives per mesh
    vkr::DevBuf<float> d_positions;          // merged xyz
    vkr::DevBuf<uint32_t> d_indices;         // merged, global vertex ids (FrameManager.cpp:1296-1304)
    vkr::DevBuf<float2> d_uvs;               // merged (FrameManager.cpp:1287-1294)
    vkr::DevBuf<float> d_normals;            // merged xyz vertex normals (G-buffer producer only)
    vkr::DevBuf<uint32_t> d_mesh_has_normals;
    vkr::DevBuf<uint2> d_lut;                // InstanceLUTEntry (FrameManager.cpp:1276-1285)
    vkr::DevBuf<uint32_t> d_mesh_class;      // per mesh AO class (0 commit, 1 skip, 2 alpha test)
    vkr::DevBuf<float4> d_material_params;   // 256 x vec4 (FrameManager.cpp:1354-1371)
    vkr::DevBuf<vkr::AlphaTexDev> d_alpha_tex;
    vkr::DevBuf<uint8_t> d_alpha_data;
    vkr::DevBuf<uint32_t> d_inst_mesh;
    vkr::DevBuf<uint32_t> d_inst_first_tri;  // exclusive prefix of primitives per instance (+1 entry)
    vkr::DevBuf<float> d_inst_xf;            // 12 floats per instance

    // ---- scene graph (Renderer::rebuildRayTracingInstances on the device)
    struct NodeDev { int32_t parent; uint32_t first_instance, instance_count, pad; float local[16]; };
    vkr::DevBuf<NodeDev> d_graph;
    uint32_t graph_nodes = 0;
    bool has_cached_model = false;
    float cached_model[16] = {0};

    // ---- GPU-driven indirect build (FrameManager::buildSharedDrawSlots / prepareSharedOpaqueIndirect)
    bool has_indirect = false;
    uint32_t ib_node_count = 0, ib_slot_count = 0, ib_mesh_count = 0, ib_max_draws = 1, ib_span_capacity = 1;
    vkr::DevBuf<int32_t> d_ib_parent;
    vkr::DevBuf<float4> d_ib_local, d_ib_world;   // 4 x float4 per linear node
    vkr::DevBuf<uint4> d_ib_slots;                // sorted SharedOpaqueDrawSlot
    vkr::DevBuf<uint32_t> d_ib_infos;             // MeshDrawInfo, 3 words per mesh
    vkr::DevBuf<float> d_ib_bounds;               // optional mesh AABBs, 6 floats per mesh
    vkr::DevBuf<uint8_t> d_ib_flags;
    vkr::DevBuf<uint32_t> d_ib_tile, d_ib_counters;
    vkr::DevBuf<uint32_t> d_ib_seg_begin, d_ib_seg_rank;   // runs of equal (doubleSided, matIndex): first slot (+ n_slots), rank
    uint32_t ib_seg_count = 0;
    vkr::DevBuf<uint4> d_ib_spans;
    std::vector<uint4> h_ib_slots;
    vkr::DevBuf<uint32_t> d_ib_order;             // nodes grouped by hierarchy level (level-per-launch world matrices)
    std::vector<uint32_t> ib_level_first;         // level l = order[first[l] .. first[l+1]); empty = chain-walk kernel

    // ---- IBL precompute scratch: irradiance sin/cos tables (512 float2) + GGX half-vector table (1024 float4)
    vkr::DevBuf<float> d_ibl_tables;
    bool ibl_quads = true;             // VKR_IBL_QUADS=0: sample the cube image directly (the path cubes above 2048^2 take)
    vkr::DevBuf<float4> d_ibl_quads;   // footprint table of the environment cube (4 float4 per bilinear footprint), grow-only

    // ---- BVH
    bool has_bvh = false;
    uint32_t morton_bits = 21;
    uint32_t bvh_height = 0;
    uint32_t wide_count = 0, wide_levels = 0;
    std::vector<uint32_t> wide_level_start;   // level l = wide nodes [start[l], start[l+1]); filled by vkr_bvh_build
    bool refit_wide = true;    // refit level by level on the wide tree (VKR_REFIT=binary: binary fit + regather)
    bool bounds_dirty = false;
    bool morton_emc = false;   // extended Morton codes (size component), see k_morton
    bool treelets = true;      // range-local PLOC rebuild of the small subtrees (k_treelet_ploc)
    bool validate_tree = false; // VKR_VALIDATE_TREE=1: host-side check of the binary tree before the fit walks it
    vkr::DevBuf<uint32_t> d_treelet;   // [0] root count, [1] work counter, [2..] treelet roots
    bool sah_top = true;       // binned-SAH splits for ranges above the treelet size (k_sah_*), else Morton splits all the way
    vkr::DevBuf<uint32_t> d_sah;       // per-triangle range id / path / depth, per-level range records and bins
    vkr::DevBuf<float4> d_nodes;             // 4 x float4 per wide node (traverse.hpp)
    vkr::DevBuf<uint4> d_wide_bref;          // per wide node: binary refs of its children (refit only)
    vkr::DevBuf<int2> d_bin_child;           // binary LBVH (build/refit only)
    vkr::DevBuf<uint2> d_bin_range;
    vkr::DevBuf<unsigned long long> d_deposit;  // 8 x u64 per binary node: fit hand-off slots
    vkr::DevBuf<float4> d_node_box;          // 2 x float4 per binary node: min.xyz|height, max.xyz
    vkr::DevBuf<uint32_t> d_wide_ctl, d_wide_bin;
    vkr::DevBuf<float4> d_tris;              // 3 x float4 per leaf, Morton order
    vkr::DevBuf<float4> d_tri_alpha;         // 3 x float4 per triangle (MASK ones written): UVs, cutoff and texture descriptor for the alpha test
    vkr::DevBuf<uint32_t> d_perm;            // sorted position -> flat triangle id
    vkr::DevBuf<uint32_t> d_tri_inst;        // sorted position -> instance the triangle came from (G-buffer producer)
    vkr::DevBuf<uint32_t> d_leaf_parent, d_node_parent, d_flags;
    vkr::DevBuf<uint64_t> d_keys[2];
    vkr::DevBuf<uint32_t> d_vals[2];
    vkr::DevBuf<uint32_t> d_hist, d_digit_totals;
    vkr::DevBuf<vkr::BvhBounds> d_bounds;
    vkr::DevBuf<uint32_t> d_height;
    float bounds_min[3] = {0, 0, 0}, bounds_max[3] = {0, 0, 0};
    void* build_stream = nullptr;

    // ---- RTAO images (FrameManager::createRtaoComputeTextures)
    vkr::DevBuf<uint16_t> d_hist_img[2], d_ping[2];
    vkr::DevBuf<float4> d_raybuf;            // K1 ray slices: per wave (2 float4/ray + retry list), grow-only
    vkr::DevBuf<uint32_t> d_tile_counter;    // K1 persistent waves pull items from it (8 counters, 64 B apart)
    vkr::DevBuf<float4> d_shq_raybuf;        // k_shadow_q: its own ray slices and chunk counter, so a shadow launch on a
    vkr::DevBuf<uint32_t> d_shq_counter;     // second stream can overlap K1 (point lights always; sun under VKR_SUN_MODE=ray)
    vkr::DevBuf<uint8_t> d_sun_frame;        // K4 packet form: the frame's sample directions (SunFrame)
    int sun_reserve_waves = 4;               // K4 packet form: wave slots per CU left free (VKR_SUN_RESERVE_WAVES)
    vkr::DevBuf<uint32_t> d_sun_counter;     // K4 packet form: chunk counter of its persistent waves (separate from K1's)
    bool sun_packets = true;                 // K4 form: one packet per pixel (k_shadow_sun_pkt) or one item per ray (k_shadow_q)
    int atrous_mode = 1;                     // K2 form: 1 LDS-tiled (default), 0 per-pixel gathers (VKR_ATROUS=gather), 2 tiled + v_exp_f32 (VKR_ATROUS=tol, outside the bit-exact contract)
    bool gbuffer_packets = true;             // G-buffer producer: one packet walk per 8x8 tile (default) or one walk per lane (VKR_GBUFFER_MODE=lane)
    bool trace_streaming = true;             // K1 form: wave-wide queues (k_trace_q) or one ray per lane (k_trace)
    int num_cus = 0;
    int k1_per_cu[2][3] = {{0, 0, 0}, {0, 0, 0}};   // resident k_trace_q workgroups per CU, [STATS][RB - 6]; 0 = not asked yet
    int sun_per_cu = 0, shq_per_cu[2] = {0, 0};     // the same for k_shadow_sun_pkt and k_shadow_q<0/1>
    int q_ray_bits_override = 0;                    // VKR_Q_RAY_BITS (tests)
    int k1_extra_lds = 0;                           // VKR_K1_EXTRA_LDS (development)
    vkr::DevBuf<float4> d_pixctx;            // K1_SPLIT: 3 float4 per pixel from k_pix_setup (rtao_kernels.hip)
    vkr::DevBuf<float2> d_vis;               // K1_SPLIT: summed visibility + hit bits per pixel, k_trace_q -> k_pix_resolve
    vkr::DevBuf<float4> d_tile_bounds;       // K1_CUT: per 8x8 tile of a launch, the bound of its ray segments (k_pix_setup -> k_tile_cut)
    vkr::DevBuf<uint4> d_cuts;               // K1_CUT: per 8x8 tile of a launch, header + entries of its entry cut (k_tile_cut)
    uint32_t cut_l2_begin = 0, cut_l2_count = 0;   // third level of the wide tree, when k_tile_cut may start there (vkr_bvh_build)
    uint32_t cut_max = 16;                   // entries per cut at most (VKR_CUT_MAX, <= K1_CUT_MAX)
    float cut_small = 1.0f;                  // a node joins the cut once its half area is <= this x the half area of the tile's ray bound (VKR_CUT_SMALL)
    bool cut_fused = true;                   // the cuts are made by k_pix_setup's own waves (VKR_CUT_FUSED=0: by a k_tile_cut launch after it)
    bool sun_tiles = true;                   // sun packets: chunks are 8x8 tiles when the arrays are rows of the image (VKR_SUN_TILES=0: runs along a row)
    vkr::DevBuf<float4> d_nz;                // decoded normal.xyz + clamped depth: operand plane of the gather form of A-Trous (VKR_ATROUS=gather) only
    const uint16_t* at_depth = nullptr;      // the frame's A-Trous operands as the caller handed them over (rtao_prep_launch) ...
    const uint16_t* at_normal = nullptr;
    int at_y0 = 0, at_y1 = 0;                // ... the rows they are needed on, and whether d_nz holds them yet
    bool at_nz_valid = false;
    uint32_t frame_parity = 0;               // FrameManager currentFrame % 2
    uint32_t last_curr = 0;

    // ---- multi-GPU collective behind the boundary (comm.hip): borrowed RCCL communicator or the host's own all-gather
    void* comm_nccl = nullptr;
    vkr_allgather_fn comm_fn = nullptr;
    void* comm_user = nullptr;
    uint32_t comm_rank = 0, comm_n = 0;          // comm_n == 0: nothing attached
    vkr::DevBuf<uint16_t> d_comm_send, d_comm_recv;

    // ---- counters
    bool stats_enabled = false;
    vkr::DevBuf<unsigned long long> d_stats; // rays, nodes, tris
};

namespace vkr {

inline SceneDev scene_dev(const vkr_ctx* c)
{
    SceneDev s;
    s.nodes = c->d_nodes.p;
    s.tris = c->d_tris.p;
    s.tri_count = c->has_bvh ? c->tri_count : 0;
    s.node_count = c->has_bvh ? c->wide_count : 0;
    s.lut = c->d_lut.p;
    s.indices = c->d_indices.p;
    s.uvs = c->d_uvs.p;
    s.material_params = c->d_material_params.p;
    s.alpha_tex = c->d_alpha_tex.p;
    s.alpha_data = c->d_alpha_data.p;
    s.tri_alpha = c->d_tri_alpha.p;
    s.tri_inst = c->d_tri_inst.p;
#ifdef VKR_PKT_STATS
    s.pkt_stats = c->d_stats.p;
#endif
    s.normals = c->d_normals.p;
    s.mesh_has_normals = c->d_mesh_has_normals.p;
    s.inst_first_tri = c->d_inst_first_tri.p;
    s.inst_xf = c->d_inst_xf.p;
    s.instance_count = c->instance_count;
    return s;
}

// bvh_build.hip
hipError_t bvh_build_launch(vkr_ctx* c, bool refit_only, hipStream_t st);
hipError_t bvh_wide_launch(vkr_ctx* c, uint32_t bin_height, hipStream_t st);
hipError_t instances_from_nodes_launch(vkr_ctx* c, const float model[16], hipStream_t st);
// indirect_build.hip
hipError_t indirect_prepare_launch(vkr_ctx* c, const float model[16], const float* planes24, float4* draw_data,
                                   void* commands, hipStream_t st);
// rtao_kernels.hip
hipError_t rtao_trace_launch(vkr_ctx* c, const FrameParams& fp, const uint16_t* depth, const uint16_t* hist_prev,
                             uint16_t* hist_curr, uint32_t* hit_mask, int y0, int y1, hipStream_t st);
hipError_t rtao_prep_launch(vkr_ctx* c, const uint16_t* depth, const uint16_t* normal, int y0, int y1, hipStream_t st);
hipError_t rtao_atrous_launch(vkr_ctx* c, float radius_raw, int step, const uint16_t* src, uint16_t* dst,
                              uint16_t* dst2, int y0, int y1, hipStream_t st);
hipError_t rtao_upsample_launch(vkr_ctx* c, uint32_t debug_mode, const uint16_t* src, const uint16_t* normal,
                                uint16_t* dst, int y0, int y1, hipStream_t st);
hipError_t fill_r16f_launch(uint16_t* img, size_t count, float value, hipStream_t st);
// shadow_kernels.hip
hipError_t shadow_sun_launch(vkr_ctx* c, const float* wp, const float* ng, const uint8_t* valid, uint32_t count,
                             const float L[3], float radius, float samples, float* out, hipStream_t st);
hipError_t noalloc_scratch_init(int device);
hipError_t shadow_prepare(vkr_ctx* c);
hipError_t rtao_prepare(vkr_ctx* c, hipStream_t st);
hipError_t shadow_sun_noalloc_launch(vkr_ctx* c, const float* wp, const float* ng, const uint8_t* valid, uint32_t count,
                                     const float L[3], float radius, float samples, float* out, hipStream_t st);
hipError_t shadow_point_launch(vkr_ctx* c, const float* wp, const float* ng, const uint8_t* valid, uint32_t count,
                               const float P[3], float* out, hipStream_t st);
hipError_t gbuffer_launch(vkr_ctx* c, const FrameParams& fp, uint16_t* depth, uint16_t* normal, float* wp, float* ng,
                          uint8_t* valid, hipStream_t st);
hipError_t gbuffer_msaa_launch(vkr_ctx* c, const FrameParams& fp, int samples, uint16_t* depth, uint16_t* normal, float* wp,
                               float* ng, uint8_t* valid, hipStream_t st);

}  // namespace vkr
