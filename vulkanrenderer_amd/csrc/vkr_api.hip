// vkr_api.hip -- the C ABI of include/vkr_rt.h: context, scene upload, BVH build/refit, the RtaoComputePass
// dispatch sequence and the shadow/G-buffer entry points.  Host code only; kernels live in the other files.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "internal.hpp"

using namespace vkr;

namespace {

std::string g_create_error;

int fail(vkr_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_OR_FAIL(c, call)                                                                        \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail((c), e_ == hipErrorOutOfMemory ? VKR_ERR_OOM : VKR_ERR_HIP, "%s: %s", #call, \
                        hipGetErrorString(e_));                                                     \
    } while (0)

// inverse(mat4) hoisted from the shader (rtao_trace_half.comp:245-246): cofactors in double, one rounding.
void inverse_m4(const float* mf, float* out)
{
    double m[16], inv[16];
    for (int i = 0; i < 16; ++i) m[i] = (double)mf[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    double id = 1.0 / det;
    for (int i = 0; i < 16; ++i) out[i] = (float)(inv[i] * id);
}

inline float gmax_h(float a, float b) { return (a < b) ? b : a; }

FrameParams make_frame_params(const vkr_ctx* c, const float* ubo)
{
    FrameParams fp;
    inverse_m4(ubo + VKR_UBO_PROJ, fp.invProj);
    inverse_m4(ubo + VKR_UBO_VIEW, fp.invView);
    memcpy(fp.view, ubo + VKR_UBO_VIEW, 64);
    memcpy(fp.prevViewProj, ubo + VKR_UBO_PREV_VIEW_PROJ, 64);
    memcpy(fp.camPos, ubo + VKR_UBO_CAM_POS, 12);
    fp.debugMode = (uint32_t)(ubo[VKR_UBO_IBL_PARAMS + 3] + 0.5f);             // rtao_trace_half.comp:250
    int rc = (int)(ubo[VKR_UBO_RTAO_PARAMS0 + 1] + 0.5f);                        // :252
    fp.rayCount = rc < 1 ? 1 : (rc > 16 ? 16 : rc);
    fp.baseRadius = gmax_h(ubo[VKR_UBO_RTAO_PARAMS0 + 2], 0.01f);                // :253
    fp.baseBias = gmax_h(ubo[VKR_UBO_RTAO_PARAMS0 + 3], 0.0005f);                // :254
    fp.frameIndex = ubo[VKR_UBO_RTAO_PARAMS1 + 3];
    fp.temporalAlphaRaw = ubo[VKR_UBO_RTAO_PARAMS1 + 1];
    fp.disocclusionRaw = ubo[VKR_UBO_RTAO_PARAMS1 + 2];
    fp.radiusRaw = ubo[VKR_UBO_RTAO_PARAMS0 + 2];
    fp.W = (int)c->W;
    fp.H = (int)c->H;
    fp.stripeMul = 1;
    return fp;
}

int check_io(vkr_ctx* c, const vkr_rtao_frame_io* io, bool need_normal, bool need_out)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!io) return fail(c, VKR_ERR_INVALID_ARG, "io is NULL");
    if (!io->linear_depth_r16f) return fail(c, VKR_ERR_INVALID_ARG, "linear_depth_r16f is NULL");
    if (need_normal && !io->normal_rgba16f) return fail(c, VKR_ERR_INVALID_ARG, "normal_rgba16f is NULL");
    if (need_out && !io->rtao_full_r16f) return fail(c, VKR_ERR_INVALID_ARG, "rtao_full_r16f is NULL");
    return VKR_OK;
}

int set_device(vkr_ctx* c) { HIP_OR_FAIL(c, hipSetDevice(c->device)); return VKR_OK; }

}  // namespace

extern "C" {

int vkr_create(const vkr_create_desc* desc, vkr_ctx** out)
{
    if (!desc || !out) return fail(nullptr, VKR_ERR_INVALID_ARG, "vkr_create: NULL argument");
    *out = nullptr;
    if (desc->width == 0 || desc->height == 0 || desc->width > 32768u || desc->height > 32768u)
        return fail(nullptr, VKR_ERR_INVALID_ARG, "vkr_create: bad extent %ux%u", desc->width, desc->height);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, VKR_ERR_HIP, "vkr_create: no HIP device (%s)", hipGetErrorString(e));
    if (desc->device < 0 || desc->device >= ndev)
        return fail(nullptr, VKR_ERR_INVALID_ARG, "vkr_create: device %d out of range (%d devices)", desc->device, ndev);
    vkr_ctx* c = new vkr_ctx();
    c->device = desc->device;
    c->W = desc->width;
    // experiment switches (development only; defaults are the shipped configuration)
    if (const char* e = getenv("VKR_MORTON_EMC")) c->morton_emc = (e[0] == '1');
    if (const char* e = getenv("VKR_TREELETS")) c->treelets = (e[0] != '0');            // "0" = plain Karras tree
    if (const char* e = getenv("VKR_SAH_TOP")) c->sah_top = (e[0] != '0');              // "0" = Morton splits above the treelets
    if (const char* e = getenv("VKR_IBL_QUADS")) c->ibl_quads = (e[0] != '0');
    if (const char* e = getenv("VKR_REFIT")) c->refit_wide = (e[0] != 'b');             // "binary" = k_fit + k_wide_refit
    if (const char* e = getenv("VKR_SUN_RESERVE_WAVES")) { int v = atoi(e); c->sun_reserve_waves = v < 0 ? 0 : (v > 16 ? 16 : v); }
    if (const char* e = getenv("VKR_VALIDATE_TREE")) c->validate_tree = (e[0] == '1');
    if (const char* e = getenv("VKR_Q_RAY_BITS")) c->q_ray_bits_override = atoi(e);
    if (const char* e = getenv("VKR_K1_EXTRA_LDS")) { int v = atoi(e); c->k1_extra_lds = v < 0 ? 0 : v; }
    if (const char* e = getenv("VKR_CUT_MAX")) { int v = atoi(e); c->cut_max = v < 1 ? 1u : (uint32_t)v; }   // 1 = every ray enters at the root's children
    if (const char* e = getenv("VKR_CUT_SMALL")) { float v = (float)atof(e); if (v > 0.0f) c->cut_small = v; }
    if (const char* e = getenv("VKR_CUT_FUSED")) c->cut_fused = atoi(e) != 0;
    if (const char* e = getenv("VKR_SUN_TILES")) c->sun_tiles = atoi(e) != 0;
    if (const char* e = getenv("VKR_ATROUS")) c->atrous_mode = !strcmp(e, "gather") ? 0 : (!strcmp(e, "tol") ? 2 : 1);
    if (const char* e = getenv("VKR_TRACE_MODE")) c->trace_streaming = (e[0] != 'l');   // "lane" = one ray per lane
    if (const char* e = getenv("VKR_GBUFFER_MODE")) c->gbuffer_packets = (e[0] != 'l');  // "lane" = one walk per lane
    if (const char* e = getenv("VKR_SUN_MODE")) c->sun_packets = (e[0] != 'r');         // "ray" = one queue item per ray
    (void)hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, c->device);
    c->H = desc->height;
    auto bail = [&](int code, const char* what, hipError_t err) {
        fail(nullptr, code, "vkr_create: %s: %s", what, hipGetErrorString(err));
        vkr_destroy(c);
        return code;
    };
    if ((e = hipSetDevice(c->device)) != hipSuccess) return bail(VKR_ERR_HIP, "hipSetDevice", e);
    // where this GPU holds the fixed scratch array of vkr_shadow_sun_noalloc (hipGetSymbolAddress, once per device)
    if ((e = noalloc_scratch_init(c->device)) != hipSuccess) return bail(VKR_ERR_HIP, "scratch symbol address", e);
    size_t px = (size_t)c->W * c->H;
    for (int i = 0; i < 2; ++i) {
        if ((e = c->d_hist_img[i].alloc(px)) != hipSuccess) return bail(VKR_ERR_OOM, "history image", e);
        if ((e = c->d_ping[i].alloc(px)) != hipSuccess) return bail(VKR_ERR_OOM, "ping-pong image", e);
    }
    if ((e = c->d_nz.alloc(px)) != hipSuccess) return bail(VKR_ERR_OOM, "A-Trous operand", e);
    if ((e = c->d_stats.alloc(24)) != hipSuccess) return bail(VKR_ERR_OOM, "counters", e);
    if ((e = hipMemset(c->d_stats.p, 0, 192)) != hipSuccess) return bail(VKR_ERR_HIP, "hipMemset", e);
    int rc = vkr_rtao_reset(c, nullptr);
    if (rc != VKR_OK) { g_create_error = c->err; vkr_destroy(c); return rc; }
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return bail(VKR_ERR_HIP, "hipStreamSynchronize", e);
    *out = c;
    return VKR_OK;
}

void vkr_destroy(vkr_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    c->d_positions.release(); c->d_indices.release(); c->d_uvs.release(); c->d_lut.release();
    c->d_mesh_class.release(); c->d_material_params.release(); c->d_alpha_tex.release(); c->d_alpha_data.release();
    c->d_inst_mesh.release(); c->d_inst_first_tri.release(); c->d_inst_xf.release();
    c->d_nodes.release(); c->d_tris.release(); c->d_tri_alpha.release(); c->d_perm.release(); c->d_tri_inst.release(); c->d_leaf_parent.release();
    c->d_node_parent.release(); c->d_flags.release();
    c->d_bin_child.release(); c->d_bin_range.release(); c->d_deposit.release(); c->d_node_box.release();
    c->d_wide_ctl.release(); c->d_wide_bin.release(); c->d_wide_bref.release();
    for (int i = 0; i < 2; ++i) { c->d_keys[i].release(); c->d_vals[i].release(); c->d_hist_img[i].release(); c->d_ping[i].release(); }
    c->d_hist.release(); c->d_digit_totals.release(); c->d_bounds.release(); c->d_height.release();
    c->d_nz.release(); c->d_stats.release(); c->d_raybuf.release(); c->d_pixctx.release(); c->d_vis.release(); c->d_cuts.release(); c->d_tile_bounds.release();
    c->d_normals.release(); c->d_mesh_has_normals.release(); c->d_graph.release(); c->d_tile_counter.release();
    c->d_sun_frame.release(); c->d_sun_counter.release(); c->d_shq_raybuf.release(); c->d_shq_counter.release(); c->d_comm_send.release(); c->d_comm_recv.release();
    c->d_ib_parent.release(); c->d_ib_local.release(); c->d_ib_world.release(); c->d_ib_slots.release();
    c->d_ib_infos.release(); c->d_ib_bounds.release(); c->d_ib_flags.release(); c->d_ib_tile.release();
    c->d_ib_counters.release(); c->d_ib_seg_begin.release(); c->d_ib_seg_rank.release(); c->d_ib_spans.release();
    c->d_ib_order.release(); c->d_ibl_tables.release(); c->d_ibl_quads.release(); c->d_treelet.release(); c->d_sah.release();
    delete c;
}

const char* vkr_last_error(const vkr_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int vkr_sync(vkr_ctx* c, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (int rc = set_device(c)) return rc;
    HIP_OR_FAIL(c, hipStreamSynchronize((hipStream_t)stream));
    return VKR_OK;
}

// ------------------------------------------------------------------------------------------------ scene
int vkr_scene_upload(vkr_ctx* c, const vkr_scene_desc* s)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!s) return fail(c, VKR_ERR_INVALID_ARG, "scene is NULL");
    if ((s->mesh_count && !s->meshes) || (s->instance_count && !s->instances))
        return fail(c, VKR_ERR_INVALID_ARG, "scene: NULL mesh/instance array");
    if (int rc = set_device(c)) return rc;
    c->has_scene = false;
    c->has_bvh = false;
    c->graph_nodes = 0;
    c->has_cached_model = false;

    // merged buffers exactly as FrameManager::createReflectionBuffers builds them (FrameManager.cpp:1264-1304)
    std::vector<float> positions;
    std::vector<uint32_t> indices;
    std::vector<float2> uvs;
    std::vector<float> normals;
    std::vector<uint32_t> mesh_has_normals(s->mesh_count);
    std::vector<uint2> lut(s->mesh_count);
    std::vector<uint32_t> mesh_class(s->mesh_count);
    c->h_mesh_prims.assign(s->mesh_count, 0);
    float mp[VKR_MAX_MATERIALS * 4];
    for (uint32_t m = 0; m < VKR_MAX_MATERIALS; ++m) {  // defaults of FrameManager.cpp:1359-1360
        mp[4 * m + 0] = s->material_params ? s->material_params[4 * m + 0] : 0.5f;
        mp[4 * m + 1] = s->material_params ? s->material_params[4 * m + 1] : 0.0f;
        mp[4 * m + 2] = 0.0f;
        mp[4 * m + 3] = 0.0f;
    }
    uint64_t voff = 0, ioff = 0;
    for (uint32_t m = 0; m < s->mesh_count; ++m) {
        const vkr_mesh_desc& md = s->meshes[m];
        if ((md.vertex_count && !md.positions) || (md.index_count && !md.indices))
            return fail(c, VKR_ERR_INVALID_ARG, "mesh %u: NULL positions/indices", m);
        if (md.vertex_count && md.position_stride < 12) return fail(c, VKR_ERR_INVALID_ARG, "mesh %u: position_stride < 12", m);
        if (md.uvs && md.uv_stride < 8) return fail(c, VKR_ERR_INVALID_ARG, "mesh %u: uv_stride < 8", m);
        if (voff + md.vertex_count > 0xffffffffull || ioff + md.index_count > 0xffffffffull)
            return fail(c, VKR_ERR_LIMIT, "scene exceeds 2^32 vertices/indices");
        int mi = md.material_index;
        lut[m].x = (mi >= 0 && mi < (int)VKR_MAX_MATERIALS) ? (uint32_t)mi : 0u;  // FrameManager.cpp:1280-1283
        lut[m].y = (uint32_t)ioff;
        // AO candidate class: the branch chain of rtao_trace_half.comp:142-173 evaluated per mesh.
        // opaque geometry never produces a candidate (auto-commit); otherwise alphaMode decides.
        float mode = mp[4 * lut[m].x + 1];
        mesh_class[m] = md.opaque ? 0u : (mode < 0.5f ? 0u : (mode > 1.5f ? 1u : 2u));
        c->h_mesh_prims[m] = md.index_count / 3u;                                 // RayTracingContext.cpp:230
        for (uint32_t v = 0; v < md.vertex_count; ++v) {
            const float* p = (const float*)((const char*)md.positions + (size_t)v * md.position_stride);
            positions.push_back(p[0]); positions.push_back(p[1]); positions.push_back(p[2]);
            if (md.uvs) {
                const float* t = (const float*)((const char*)md.uvs + (size_t)v * md.uv_stride);
                uvs.push_back(make_float2(t[0], t[1]));
            } else {
                uvs.push_back(make_float2(0.0f, 0.0f));
            }
            if (md.normals) {
                const float* nn = (const float*)((const char*)md.normals + (size_t)v * md.normal_stride);
                normals.push_back(nn[0]); normals.push_back(nn[1]); normals.push_back(nn[2]);
            } else {
                normals.push_back(0.0f); normals.push_back(1.0f); normals.push_back(0.0f);   // Vertex.h default
            }
        }
        mesh_has_normals[m] = md.normals ? 1u : 0u;
        for (uint32_t i = 0; i < md.index_count; ++i) {
            if (md.indices[i] >= md.vertex_count) return fail(c, VKR_ERR_INVALID_ARG, "mesh %u: index %u out of range", m, i);
            indices.push_back((uint32_t)voff + md.indices[i]);                   // FrameManager.cpp:1299-1303
        }
        voff += md.vertex_count;
        ioff += md.index_count;
    }
    // instances (RayTracingContext::writeInstances, RayTracingContext.cpp:46-71)
    std::vector<uint32_t> inst_first(s->instance_count + 1, 0);
    std::vector<float> inst_xf((size_t)s->instance_count * 12);
    c->h_inst_mesh.resize(s->instance_count);
    uint64_t tri_total = 0;
    for (uint32_t i = 0; i < s->instance_count; ++i) {
        uint32_t m = s->instances[i].mesh_index;
        if (m >= s->mesh_count) return fail(c, VKR_ERR_INVALID_ARG, "invalid meshIndex for ray tracing instance %u", i);
        c->h_inst_mesh[i] = m;
        inst_first[i] = (uint32_t)tri_total;
        tri_total += c->h_mesh_prims[m];
        memcpy(&inst_xf[(size_t)i * 12], s->instances[i].transform, 48);
    }
    if (tri_total >= (1ull << 28)) return fail(c, VKR_ERR_LIMIT, "scene exceeds 2^28 world-space triangles");
    if (s->mesh_count >= (1u << 28)) return fail(c, VKR_ERR_LIMIT, "scene exceeds 2^28 meshes");
    if (s->instance_count) inst_first[s->instance_count] = (uint32_t)tri_total;

    // alpha planes
    std::vector<AlphaTexDev> tex(VKR_MAX_MATERIALS, AlphaTexDev{0, 0, 0, 0});
    std::vector<uint8_t> alpha;
    if (s->alpha_textures) {
        for (uint32_t m = 0; m < VKR_MAX_MATERIALS; ++m) {
            const vkr_alpha_texture& t = s->alpha_textures[m];
            if (!t.alpha || t.width == 0 || t.height == 0) continue;
            if (t.width > 16384u || t.height > 16384u) return fail(c, VKR_ERR_LIMIT, "alpha texture %u too large", m);
            tex[m].offset = (uint32_t)alpha.size();
            tex[m].w = t.width;
            tex[m].h = t.height;
            alpha.insert(alpha.end(), t.alpha, t.alpha + (size_t)t.width * t.height);
            if (alpha.size() > 0xf0000000ull) return fail(c, VKR_ERR_LIMIT, "alpha textures exceed 4 GiB");
        }
    }

#define UPLOAD(buf, vec)                                                                                    \
    do {                                                                                                    \
        HIP_OR_FAIL(c, (buf).alloc((vec).size() ? (vec).size() : 1));                                       \
        if ((vec).size())                                                                                   \
            HIP_OR_FAIL(c, hipMemcpy((buf).p, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice)); \
    } while (0)
    UPLOAD(c->d_positions, positions);
    UPLOAD(c->d_indices, indices);
    UPLOAD(c->d_uvs, uvs);
    UPLOAD(c->d_normals, normals);
    UPLOAD(c->d_mesh_has_normals, mesh_has_normals);
    UPLOAD(c->d_lut, lut);
    UPLOAD(c->d_mesh_class, mesh_class);
    UPLOAD(c->d_alpha_tex, tex);
    UPLOAD(c->d_alpha_data, alpha);
    UPLOAD(c->d_inst_mesh, c->h_inst_mesh);
    UPLOAD(c->d_inst_first_tri, inst_first);
    UPLOAD(c->d_inst_xf, inst_xf);
#undef UPLOAD
    HIP_OR_FAIL(c, c->d_material_params.alloc(VKR_MAX_MATERIALS));
    HIP_OR_FAIL(c, hipMemcpy(c->d_material_params.p, mp, sizeof(mp), hipMemcpyHostToDevice));

    c->mesh_count = s->mesh_count;
    c->instance_count = s->instance_count;
    c->tri_count = (uint32_t)tri_total;
    const size_t n = c->tri_count ? c->tri_count : 1;
    HIP_OR_FAIL(c, c->d_nodes.alloc(n * 4));
    HIP_OR_FAIL(c, c->d_wide_bref.alloc(n));
    HIP_OR_FAIL(c, c->d_bin_child.alloc(n));
    HIP_OR_FAIL(c, c->d_bin_range.alloc(n));
    HIP_OR_FAIL(c, c->d_deposit.alloc(n * 8));
    HIP_OR_FAIL(c, c->d_node_box.alloc(n * 2));
    HIP_OR_FAIL(c, c->d_wide_ctl.alloc(4 + 72));
    HIP_OR_FAIL(c, c->d_wide_bin.alloc(n));
    HIP_OR_FAIL(c, c->d_tris.alloc(n * 3));
    HIP_OR_FAIL(c, c->d_tri_alpha.alloc(n * 3));
    HIP_OR_FAIL(c, c->d_perm.alloc(n));
    HIP_OR_FAIL(c, c->d_tri_inst.alloc(n));
    HIP_OR_FAIL(c, c->d_leaf_parent.alloc(n));
    HIP_OR_FAIL(c, c->d_node_parent.alloc(n));
    HIP_OR_FAIL(c, c->d_flags.alloc(n));
    for (int i = 0; i < 2; ++i) {
        HIP_OR_FAIL(c, c->d_keys[i].alloc(n));
        HIP_OR_FAIL(c, c->d_vals[i].alloc(n));
    }
    const size_t ntiles = (n + 2047) / 2048;
    HIP_OR_FAIL(c, c->d_hist.alloc(256 * ntiles));
    HIP_OR_FAIL(c, c->d_digit_totals.alloc(256));
    HIP_OR_FAIL(c, c->d_bounds.alloc(1));
    HIP_OR_FAIL(c, c->d_height.alloc(1));
    c->has_scene = true;
    return VKR_OK;
}

static int read_height_and_bounds(vkr_ctx* c, hipStream_t st)
{
    BvhBounds b;
    uint32_t h = 0;
    HIP_OR_FAIL(c, hipMemcpyAsync(&h, c->d_height.p, 4, hipMemcpyDeviceToHost, st));
    HIP_OR_FAIL(c, hipMemcpyAsync(&b, c->d_bounds.p, sizeof(b), hipMemcpyDeviceToHost, st));
    HIP_OR_FAIL(c, hipStreamSynchronize(st));
    c->bvh_height = h;
    c->bounds_dirty = false;
    for (int k = 0; k < 3; ++k) { c->bounds_min[k] = dec_f(b.tri_min[k]); c->bounds_max[k] = dec_f(b.tri_max[k]); }
    return VKR_OK;
}

int vkr_bvh_build(vkr_ctx* c, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!c->has_scene) return fail(c, VKR_ERR_STATE, "vkr_bvh_build before vkr_scene_upload");
    if (int rc = set_device(c)) return rc;
    hipStream_t st = (hipStream_t)stream;
    c->has_bvh = false;
    c->bvh_height = 0;
    c->wide_count = 0;
    c->wide_levels = 0;
    c->cut_l2_begin = c->cut_l2_count = 0;
    if (c->tri_count == 0) {
        // empty scene (RayTracingContext::init with no instances): a root whose four slots are all unused, in the wide-node
        // layout of traverse.hpp -- origin 0, the three scales 1.0f (q0.w = scale_z, q2.z = scale_x, q2.w = scale_y), planes
        // stored inverted (qmin = 255, qmax = 0) so every slab test fails by itself.  The traversal kernels do not rely on
        // that alone: they never walk a tree of zero triangles (sc.tri_count == 0 is checked before a root item is queued).
        const uint32_t none = (uint32_t)VKR_CHILD_NONE;
        const uint32_t one = 0x3f800000u;
        const uint32_t root[16] = {0u, 0u, 0u, one,
                                   0xffffffffu, 0xffffffffu, 0xffffffffu, 0u,
                                   0u, 0u, one, one, none, none, none, none};
        HIP_OR_FAIL(c, hipMemcpyAsync(c->d_nodes.p, root, sizeof(root), hipMemcpyHostToDevice, st));
        HIP_OR_FAIL(c, hipStreamSynchronize(st));
        c->has_bvh = true;
        return VKR_OK;
    }
    // 21 bits/axis with the treelet rebuild; if the wide tree comes out deeper than the traversal stack allows, first
    // drop the treelets (clustering has no depth bound), then fall back to 10 bits/axis, where the index tie-break of
    // the Karras tree bounds the height by 30 + log2(n) (pathological duplicate centroids).
    // Try 0 additionally puts binned-SAH splits on top (no depth bound either); it is skipped when switched off.
    const uint32_t tries[4] = {21u, 21u, 21u, 10u};
    const bool want_treelets = c->treelets, want_sah = c->sah_top;
    for (int t = 0; t < 4; ++t) {
        if (t == 0 && !want_sah) continue;
        if (t == 2 && !want_treelets) continue;                  // same configuration as try 1
        c->morton_bits = tries[t];
        c->sah_top = want_sah && t == 0;
        c->treelets = want_treelets && t <= 1;
        hipError_t be = bvh_build_launch(c, false, st);
        if (be != hipSuccess) { c->treelets = want_treelets; c->sah_top = want_sah; HIP_OR_FAIL(c, be); }
        // the binary height bounds the number of BFS levels of the collapse; it is one word
        if (int rc = read_height_and_bounds(c, st)) { c->treelets = want_treelets; c->sah_top = want_sah; return rc; }
        hipError_t we = bvh_wide_launch(c, c->bvh_height, st);
        c->treelets = want_treelets;
        c->sah_top = want_sah;
        HIP_OR_FAIL(c, we);
        uint32_t ctl[4 + 72] = {0};
        HIP_OR_FAIL(c, hipMemcpyAsync(ctl, c->d_wide_ctl.p, sizeof(ctl), hipMemcpyDeviceToHost, st));
        HIP_OR_FAIL(c, hipStreamSynchronize(st));
        c->wide_count = ctl[0];
        c->wide_levels = ctl[3];
        c->wide_level_start.clear();
        if (c->wide_levels <= 70u) c->wide_level_start.assign(ctl + 4, ctl + 4 + c->wide_levels + 1u);
        // near-first traversal defers at most 3 children per level
        if (3u * c->wide_levels + 1u <= (uint32_t)VKR_MAX_STACK) {
            // K1's entry cut may start at the third level when the two levels above it hold no leaf (a leaf up there is a cut
            // entry of its own, which only the descent from the root finds): look at the root and its children once
            c->cut_l2_begin = c->cut_l2_count = 0;
            if (c->wide_level_start.size() >= 4u && c->wide_level_start[2] <= 5u) {
                const uint32_t n_top = c->wide_level_start[2];          // nodes of levels 0 and 1
                uint32_t top[5 * 16];
                HIP_OR_FAIL(c, hipMemcpyAsync(top, c->d_nodes.p, (size_t)n_top * 64u, hipMemcpyDeviceToHost, st));
                HIP_OR_FAIL(c, hipStreamSynchronize(st));
                bool leaf_up_there = false;
                for (uint32_t nd = 0; nd < n_top; ++nd)
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t ref = top[nd * 16u + 12u + (uint32_t)k];
                        if (ref != (uint32_t)VKR_CHILD_NONE && (ref & 0x80000000u)) leaf_up_there = true;
                    }
                const uint32_t cnt = c->wide_level_start[3] - c->wide_level_start[2];
                if (!leaf_up_there && cnt >= 1u && cnt <= 16u) { c->cut_l2_begin = c->wide_level_start[2]; c->cut_l2_count = cnt; }
            }
            c->has_bvh = true;
            // occupancy queries and per-frame scratch of the traversal kernels: here, once, so that the frame calls neither ask
            // the runtime anything nor allocate (and can be captured into a HIP graph)
            HIP_OR_FAIL(c, rtao_prepare(c, st));
            HIP_OR_FAIL(c, shadow_prepare(c));
            return VKR_OK;
        }
    }
    return fail(c, VKR_ERR_LIMIT, "BVH depth %u levels exceeds the traversal stack (%d entries)", c->wide_levels,
                VKR_MAX_STACK);
}

int vkr_bvh_refit(vkr_ctx* c, const vkr_instance* inst, uint32_t count, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "vkr_bvh_refit before vkr_bvh_build");
    if (count != c->instance_count)  // RayTracingContext.cpp:51-53
        return fail(c, VKR_ERR_INVALID_ARG, "instance count mismatch: RayTracingContext must be re-initialized");
    if (count && !inst) return fail(c, VKR_ERR_INVALID_ARG, "instances is NULL");
    if (int rc = set_device(c)) return rc;
    c->has_cached_model = false;   // explicit transforms supersede whatever model matrix produced the current ones
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> xf((size_t)count * 12);
    for (uint32_t i = 0; i < count; ++i) {
        if (inst[i].mesh_index != c->h_inst_mesh[i])
            return fail(c, VKR_ERR_INVALID_ARG, "instance %u: mesh index changed; rebuild required", i);
        memcpy(&xf[(size_t)i * 12], inst[i].transform, 48);
    }
    if (count) {
        HIP_OR_FAIL(c, hipMemcpyAsync(c->d_inst_xf.p, xf.data(), xf.size() * 4, hipMemcpyHostToDevice, st));
        HIP_OR_FAIL(c, hipStreamSynchronize(st));  // xf is a stack-lifetime staging vector
    }
    if (c->tri_count == 0) return VKR_OK;
    HIP_OR_FAIL(c, bvh_build_launch(c, true, st));
    c->bounds_dirty = true;   // the refit recomputed the scene bounds on the device; info_get fetches them on demand
    return VKR_OK;
}

int vkr_hierarchy_upload(vkr_ctx* c, const vkr_scene_node* nodes, uint32_t node_count)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!c->has_scene) return fail(c, VKR_ERR_STATE, "vkr_hierarchy_upload before vkr_scene_upload");
    if (node_count && !nodes) return fail(c, VKR_ERR_INVALID_ARG, "nodes is NULL");
    if (int rc = set_device(c)) return rc;
    c->graph_nodes = 0;
    c->has_cached_model = false;
    std::vector<vkr_ctx::NodeDev> h(node_count);
    std::vector<int> depth(node_count, 0);
    uint32_t running = 0;
    for (uint32_t i = 0; i < node_count; ++i) {
        const vkr_scene_node& n = nodes[i];
        if (n.parent >= (int32_t)i || n.parent < -1)
            return fail(c, VKR_ERR_INVALID_ARG, "node %u: parent %d must precede it (DFS pre-order)", i, n.parent);
        depth[i] = n.parent < 0 ? 1 : depth[n.parent] + 1;
        if (depth[i] > VKR_GRAPH_MAX_DEPTH) return fail(c, VKR_ERR_LIMIT, "node %u: hierarchy deeper than %d", i, VKR_GRAPH_MAX_DEPTH);
        // one instance per (node, meshIndex) in visiting order (Renderer.cpp:514-519): ranges tile [0, instance_count)
        if (n.first_instance != running)
            return fail(c, VKR_ERR_INVALID_ARG, "node %u: first_instance %u breaks DFS instance order (expected %u)", i,
                        n.first_instance, running);
        running += n.instance_count;
        h[i].parent = n.parent; h[i].first_instance = n.first_instance; h[i].instance_count = n.instance_count; h[i].pad = 0;
        memcpy(h[i].local, n.local, 64);
    }
    if (running != c->instance_count)   // RayTracingContext.cpp:51-53
        return fail(c, VKR_ERR_INVALID_ARG, "instance count mismatch: hierarchy yields %u instances, scene has %u", running,
                    c->instance_count);
    HIP_OR_FAIL(c, c->d_graph.alloc(node_count ? node_count : 1));
    if (node_count) HIP_OR_FAIL(c, hipMemcpy(c->d_graph.p, h.data(), h.size() * sizeof(h[0]), hipMemcpyHostToDevice));
    c->graph_nodes = node_count;
    return VKR_OK;
}

int vkr_bvh_refit_model(vkr_ctx* c, const float model[16], int* refitted, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (refitted) *refitted = 0;
    if (!model) return fail(c, VKR_ERR_INVALID_ARG, "model_matrix is NULL");
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "vkr_bvh_refit_model before vkr_bvh_build");
    if (c->graph_nodes == 0 && c->instance_count != 0)
        return fail(c, VKR_ERR_STATE, "vkr_bvh_refit_model before vkr_hierarchy_upload");
    // Renderer.cpp:471-474: glm::mat4 operator!= (float compare, so -0 == 0 and NaN != NaN)
    bool same = c->has_cached_model;
    for (int k = 0; k < 16 && same; ++k) same = (c->cached_model[k] == model[k]);
    if (same) return VKR_OK;
    if (int rc = set_device(c)) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIP_OR_FAIL(c, instances_from_nodes_launch(c, model, st));
    if (c->tri_count != 0) {
        HIP_OR_FAIL(c, bvh_build_launch(c, true, st));
        c->bounds_dirty = true;
    }
    memcpy(c->cached_model, model, 64);
    c->has_cached_model = true;
    if (refitted) *refitted = 1;
    return VKR_OK;
}

int vkr_instances_download(vkr_ctx* c, float* out)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!out) return fail(c, VKR_ERR_INVALID_ARG, "out is NULL");
    if (!c->has_scene) return fail(c, VKR_ERR_STATE, "no scene");
    if (int rc = set_device(c)) return rc;
    HIP_OR_FAIL(c, hipDeviceSynchronize());
    if (c->instance_count)
        HIP_OR_FAIL(c, hipMemcpy(out, c->d_inst_xf.p, (size_t)c->instance_count * 48, hipMemcpyDeviceToHost));
    return VKR_OK;
}

// ---------------------------------------------------------------- GPU-driven indirect build

int vkr_indirect_upload(vkr_ctx* c, const vkr_indirect_desc* d)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!d) return fail(c, VKR_ERR_INVALID_ARG, "desc is NULL");
    if ((d->node_count && (!d->node_parent || !d->node_local)) || (d->slot_count && !d->slots) ||
        (d->mesh_count && !d->mesh_infos))
        return fail(c, VKR_ERR_INVALID_ARG, "indirect desc: NULL array with non-zero count");
    if (d->slot_count > (1u << 24)) return fail(c, VKR_ERR_LIMIT, "more than 2^24 draw slots");
    if (d->node_count > (1u << 28)) return fail(c, VKR_ERR_LIMIT, "more than 2^28 linear nodes");
    if (int rc = set_device(c)) return rc;
    c->has_indirect = false;
    std::vector<int> depth(d->node_count, 0);
    for (uint32_t i = 0; i < d->node_count; ++i) {   // FrameManager.cpp:122-126 reads the parent's matrix: it must be earlier
        const int32_t p = d->node_parent[i];
        if (p >= (int32_t)i || p < -1)
            return fail(c, VKR_ERR_INVALID_ARG, "linear node %u: parent %d must precede it", i, p);
        depth[i] = p < 0 ? 1 : depth[p] + 1;
        if (depth[i] > VKR_GRAPH_MAX_DEPTH) return fail(c, VKR_ERR_LIMIT, "linear node %u: hierarchy deeper than %d", i, VKR_GRAPH_MAX_DEPTH);
    }
    // nodes grouped by level (stable: ascending linear index inside a level) for the level-per-launch world-matrix
    // kernel; hierarchies deeper than 16 levels keep the one-launch chain-walk kernel
    int max_depth = 0;
    for (uint32_t i = 0; i < d->node_count; ++i) max_depth = depth[i] > max_depth ? depth[i] : max_depth;
    c->ib_level_first.clear();
    std::vector<uint32_t> order;
    if (max_depth >= 1 && max_depth <= 16) {
        std::vector<uint32_t> first(max_depth + 1, 0);
        for (uint32_t i = 0; i < d->node_count; ++i) ++first[depth[i]];
        for (int l = 1; l <= max_depth; ++l) first[l] += first[l - 1];      // first[l] = end of level l (1-based)
        order.resize(d->node_count);
        std::vector<uint32_t> cursor(first.begin(), first.end() - 1);        // start of level l+1 = end of level l
        for (uint32_t i = 0; i < d->node_count; ++i) order[cursor[depth[i] - 1]++] = i;
        c->ib_level_first = first;
    }
    // the sort of buildSharedDrawSlots (FrameManager.cpp:110-115), once per model like the reference
    c->h_ib_slots.resize(d->slot_count);
    for (uint32_t i = 0; i < d->slot_count; ++i)
        c->h_ib_slots[i] = make_uint4(d->slots[i].node_linear_index, d->slots[i].mesh_index, d->slots[i].mat_index,
                                      d->slots[i].double_sided ? 1u : 0u);
    std::stable_sort(c->h_ib_slots.begin(), c->h_ib_slots.end(), [](const uint4& a, const uint4& b) {
        if (a.w != b.w) return a.w < b.w;
        if (a.z != b.z) return a.z < b.z;
        return a.y < b.y;
    });
    // runs of equal (doubleSided, matIndex) = the buckets of prepareSharedOpaqueIndirect's std::map (:135-136,149)
    std::vector<uint32_t> seg_begin;
    for (uint32_t i = 0; i < d->slot_count; ++i)
        if (i == 0 || c->h_ib_slots[i].w != c->h_ib_slots[i - 1].w || c->h_ib_slots[i].z != c->h_ib_slots[i - 1].z) seg_begin.push_back(i);
    const uint32_t keys = (uint32_t)seg_begin.size();
    seg_begin.push_back(d->slot_count);
    c->ib_seg_count = keys;
    c->ib_node_count = d->node_count; c->ib_slot_count = d->slot_count; c->ib_mesh_count = d->mesh_count;
    c->ib_max_draws = d->max_draws < 1u ? 1u : d->max_draws;      // std::max(1u, MAX_DRAWS), FrameManager.cpp:24
    c->ib_span_capacity = keys ? keys : 1u;
    const uint32_t tiles = (d->slot_count + 2047u) / 2048u;
    HIP_OR_FAIL(c, c->d_ib_parent.alloc(d->node_count ? d->node_count : 1));
    HIP_OR_FAIL(c, c->d_ib_local.alloc((size_t)(d->node_count ? d->node_count : 1) * 4));
    HIP_OR_FAIL(c, c->d_ib_world.alloc((size_t)(d->node_count ? d->node_count : 1) * 4));
    HIP_OR_FAIL(c, c->d_ib_slots.alloc(d->slot_count ? d->slot_count : 1));
    HIP_OR_FAIL(c, c->d_ib_infos.alloc((size_t)(d->mesh_count ? d->mesh_count : 1) * 3));
    HIP_OR_FAIL(c, c->d_ib_flags.alloc((size_t)d->slot_count + 4));   // k_ib_seg_rank reads whole 32-bit words
    HIP_OR_FAIL(c, c->d_ib_tile.alloc(tiles ? tiles : 1));
    HIP_OR_FAIL(c, c->d_ib_counters.alloc(2));
    HIP_OR_FAIL(c, c->d_ib_seg_begin.alloc(seg_begin.size()));
    HIP_OR_FAIL(c, c->d_ib_seg_rank.alloc(seg_begin.size()));
    HIP_OR_FAIL(c, hipMemcpy(c->d_ib_seg_begin.p, seg_begin.data(), seg_begin.size() * 4, hipMemcpyHostToDevice));
    HIP_OR_FAIL(c, c->d_ib_spans.alloc(c->ib_span_capacity));
    HIP_OR_FAIL(c, hipMemset(c->d_ib_counters.p, 0, 8));
    HIP_OR_FAIL(c, c->d_ib_order.alloc(order.empty() ? 1 : order.size()));
    if (!order.empty()) HIP_OR_FAIL(c, hipMemcpy(c->d_ib_order.p, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    if (d->node_count) {
        HIP_OR_FAIL(c, hipMemcpy(c->d_ib_parent.p, d->node_parent, (size_t)d->node_count * 4, hipMemcpyHostToDevice));
        HIP_OR_FAIL(c, hipMemcpy(c->d_ib_local.p, d->node_local, (size_t)d->node_count * 64, hipMemcpyHostToDevice));
    }
    if (d->slot_count) HIP_OR_FAIL(c, hipMemcpy(c->d_ib_slots.p, c->h_ib_slots.data(), (size_t)d->slot_count * 16, hipMemcpyHostToDevice));
    if (d->mesh_count) HIP_OR_FAIL(c, hipMemcpy(c->d_ib_infos.p, d->mesh_infos, (size_t)d->mesh_count * 12, hipMemcpyHostToDevice));
    c->d_ib_bounds.release();
    if (d->mesh_bounds && d->mesh_count) {
        HIP_OR_FAIL(c, c->d_ib_bounds.alloc((size_t)d->mesh_count * 6));
        HIP_OR_FAIL(c, hipMemcpy(c->d_ib_bounds.p, d->mesh_bounds, (size_t)d->mesh_count * 24, hipMemcpyHostToDevice));
    }
    c->has_indirect = true;
    return VKR_OK;
}

int vkr_indirect_slots_download(vkr_ctx* c, vkr_draw_slot* out)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!c->has_indirect) return fail(c, VKR_ERR_STATE, "vkr_indirect_slots_download before vkr_indirect_upload");
    if (!out && c->ib_slot_count) return fail(c, VKR_ERR_INVALID_ARG, "out is NULL");
    for (uint32_t i = 0; i < c->ib_slot_count; ++i) {
        out[i].node_linear_index = c->h_ib_slots[i].x; out[i].mesh_index = c->h_ib_slots[i].y;
        out[i].mat_index = c->h_ib_slots[i].z; out[i].double_sided = c->h_ib_slots[i].w;
    }
    return VKR_OK;
}

int vkr_indirect_prepare(vkr_ctx* c, const float model[16], const float* view_proj, float* draw_data,
                         vkr_draw_indexed_indirect_command* commands, uint32_t capacity, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!c->has_indirect) return fail(c, VKR_ERR_STATE, "vkr_indirect_prepare before vkr_indirect_upload");
    if (!model || !draw_data || !commands) return fail(c, VKR_ERR_INVALID_ARG, "model, draw_data and commands must be non-NULL");
    const uint32_t need = c->ib_slot_count < c->ib_max_draws ? c->ib_slot_count : c->ib_max_draws;
    if (capacity < need)
        return fail(c, VKR_ERR_INVALID_ARG, "capacity %u of draw_data/commands is below min(max_draws, slot_count) = %u", capacity, need);
    if (view_proj && !c->d_ib_bounds.p) return fail(c, VKR_ERR_STATE, "frustum culling asked for but no mesh_bounds were uploaded");
    if (int rc = set_device(c)) return rc;
    float planes[24];
    if (view_proj) {   // Frustum::extractPlanes (Frustum.h:46-60), once per frame on the host like the reference would
        const int row[6] = {0, 0, 1, 1, 2, 2};
        for (int p = 0; p < 6; ++p) {
            float* o = planes + 4 * p;
            for (int k = 0; k < 4; ++k)
                o[k] = (p & 1) ? view_proj[k * 4 + 3] - view_proj[k * 4 + row[p]] : view_proj[k * 4 + 3] + view_proj[k * 4 + row[p]];
            const float len = sqrtf((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]);
            if (len > 1e-6f) { o[0] /= len; o[1] /= len; o[2] /= len; o[3] /= len; }
        }
    }
    HIP_OR_FAIL(c, indirect_prepare_launch(c, model, view_proj ? planes : nullptr, (float4*)draw_data, commands, (hipStream_t)stream));
    return VKR_OK;
}

int vkr_indirect_result(vkr_ctx* c, uint32_t* draw_count, vkr_bucket_span* spans, uint32_t span_capacity, uint32_t* span_count,
                        void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!c->has_indirect) return fail(c, VKR_ERR_STATE, "vkr_indirect_result before vkr_indirect_upload");
    if (int rc = set_device(c)) return rc;
    HIP_OR_FAIL(c, hipStreamSynchronize((hipStream_t)stream));
    uint32_t cnt[2] = {0, 0};
    HIP_OR_FAIL(c, hipMemcpy(cnt, c->d_ib_counters.p, 8, hipMemcpyDeviceToHost));
    const uint32_t draws = cnt[0] < c->ib_max_draws ? cnt[0] : c->ib_max_draws;
    const uint32_t ns = cnt[1] < c->ib_span_capacity ? cnt[1] : c->ib_span_capacity;
    if (draw_count) *draw_count = draws;                      // sharedOpaqueDrawCount, FrameManager.cpp:174
    if (span_count) *span_count = ns;
    if (spans && ns) {
        const uint32_t n = ns < span_capacity ? ns : span_capacity;
        if (n) HIP_OR_FAIL(c, hipMemcpy(spans, c->d_ib_spans.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    }
    return VKR_OK;
}

int vkr_bvh_info_get(vkr_ctx* c, vkr_bvh_info* out)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!out) return fail(c, VKR_ERR_INVALID_ARG, "out is NULL");
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "no BVH");
    if (c->bounds_dirty) {
        if (int rc = set_device(c)) return rc;
        BvhBounds b;
        HIP_OR_FAIL(c, hipDeviceSynchronize());
        HIP_OR_FAIL(c, hipMemcpy(&b, c->d_bounds.p, sizeof(b), hipMemcpyDeviceToHost));
        for (int k = 0; k < 3; ++k) { c->bounds_min[k] = dec_f(b.tri_min[k]); c->bounds_max[k] = dec_f(b.tri_max[k]); }
        c->bounds_dirty = false;
    }
    out->triangle_count = c->tri_count;
    out->node_count = c->wide_count;
    out->height = c->wide_levels;
    out->binary_height = c->bvh_height;
    out->morton_bits = c->morton_bits;
    for (int k = 0; k < 3; ++k) { out->bounds_min[k] = c->bounds_min[k]; out->bounds_max[k] = c->bounds_max[k]; }
    return VKR_OK;
}

int vkr_bvh_download(vkr_ctx* c, float* nodes, float* tris)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "no BVH");
    if (int rc = set_device(c)) return rc;
    HIP_OR_FAIL(c, hipDeviceSynchronize());
    size_t nn = c->wide_count;
    if (nodes && nn) HIP_OR_FAIL(c, hipMemcpy(nodes, c->d_nodes.p, nn * 64, hipMemcpyDeviceToHost));
    if (tris && c->tri_count) HIP_OR_FAIL(c, hipMemcpy(tris, c->d_tris.p, (size_t)c->tri_count * 48, hipMemcpyDeviceToHost));
    return VKR_OK;
}

// ------------------------------------------------------------------------------------------------ RTAO
int vkr_rtao_reset(vkr_ctx* c, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (int rc = set_device(c)) return rc;
    hipStream_t st = (hipStream_t)stream;
    size_t px = (size_t)c->W * c->H;
    for (int i = 0; i < 2; ++i) {  // FrameManager.cpp:1057: cleared to 1.0
        HIP_OR_FAIL(c, fill_r16f_launch(c->d_hist_img[i].p, px, 1.0f, st));
        HIP_OR_FAIL(c, fill_r16f_launch(c->d_ping[i].p, px, 1.0f, st));
    }
    c->frame_parity = 0;
    c->last_curr = 0;
    return VKR_OK;
}

int vkr_rtao_end_frame(vkr_ctx* c)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    c->last_curr = c->frame_parity;
    c->frame_parity ^= 1u;  // FrameManager.cpp:1155-1165: curr = currentFrame % MAX_FRAMES_IN_FLIGHT
    return VKR_OK;
}

int vkr_rtao_image(vkr_ctx* c, uint32_t which, uint16_t** out)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!out) return fail(c, VKR_ERR_INVALID_ARG, "out is NULL");
    switch (which) {
        case 0: *out = c->d_hist_img[c->last_curr].p; break;
        case 1: *out = c->d_hist_img[c->last_curr ^ 1u].p; break;
        case 2: *out = c->d_ping[0].p; break;
        case 3: *out = c->d_ping[1].p; break;
        default: return fail(c, VKR_ERR_INVALID_ARG, "image selector %u", which);
    }
    return VKR_OK;
}

int vkr_rtao_history_curr(vkr_ctx* c, uint16_t** out)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!out) return fail(c, VKR_ERR_INVALID_ARG, "out is NULL");
    *out = c->d_hist_img[c->frame_parity].p;
    return VKR_OK;
}

int vkr_rtao_trace(vkr_ctx* c, const vkr_rtao_frame_io* io, void* stream)
{
    if (int rc = check_io(c, io, false, false)) return rc;
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "vkr_rtao_trace before vkr_bvh_build");
    if (int rc = set_device(c)) return rc;
    FrameParams fp = make_frame_params(c, io->ubo);
    uint32_t curr = c->frame_parity;
    HIP_OR_FAIL(c, rtao_trace_launch(c, fp, io->linear_depth_r16f, c->d_hist_img[curr ^ 1u].p, c->d_hist_img[curr].p,
                                     io->ray_hit_mask, 0, (int)c->H, (hipStream_t)stream));
    return VKR_OK;
}

int vkr_rtao_atrous(vkr_ctx* c, const vkr_rtao_frame_io* io, uint32_t step, uint32_t iteration, void* stream)
{
    if (int rc = check_io(c, io, true, false)) return rc;
    if (int rc = set_device(c)) return rc;
    hipStream_t st = (hipStream_t)stream;
    // rtao_atrous.comp:57-69: dst = iter&1 ; src = iter==0 ? historyCurr : ping[(iter-1)&1]
    const uint16_t* src = (iteration == 0) ? c->d_hist_img[c->frame_parity].p : c->d_ping[(iteration - 1u) & 1u].p;
    uint16_t* dst = c->d_ping[iteration & 1u].p;
    if (iteration == 0 || c->at_depth != io->linear_depth_r16f || c->at_normal != io->normal_rgba16f)
        HIP_OR_FAIL(c, rtao_prep_launch(c, io->linear_depth_r16f, io->normal_rgba16f, 0, (int)c->H, st));
    HIP_OR_FAIL(c, rtao_atrous_launch(c, io->ubo[VKR_UBO_RTAO_PARAMS0 + 2], (int)(step > 0x40000000u ? 0x40000000u : step),
                                      src, dst, nullptr, 0, (int)c->H, st));
    return VKR_OK;
}

int vkr_rtao_upsample(vkr_ctx* c, const vkr_rtao_frame_io* io, uint32_t final_index, void* stream)
{
    if (int rc = check_io(c, io, true, true)) return rc;
    if (int rc = set_device(c)) return rc;
    uint32_t debugMode = (uint32_t)(io->ubo[VKR_UBO_IBL_PARAMS + 3] + 0.5f);
    HIP_OR_FAIL(c, rtao_upsample_launch(c, debugMode, c->d_ping[final_index & 1u].p, io->normal_rgba16f,
                                        io->rtao_full_r16f, 0, (int)c->H, (hipStream_t)stream));
    return VKR_OK;
}

int vkr_rtao_execute_band(vkr_ctx* c, const vkr_rtao_frame_io* io, uint32_t y0, uint32_t y1, void* stream)
{
    if (int rc = check_io(c, io, true, true)) return rc;
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "vkr_rtao_execute before vkr_bvh_build");
    if (y0 >= y1 || y1 > c->H) return fail(c, VKR_ERR_INVALID_ARG, "band [%u,%u) outside image height %u", y0, y1, c->H);
    if (io->atrous_iterations > 16u) return fail(c, VKR_ERR_INVALID_ARG, "atrous_iterations %u > 16", io->atrous_iterations);
    if (int rc = set_device(c)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int H = (int)c->H;
    const uint32_t n = io->atrous_iterations < 1u ? 1u : io->atrous_iterations;  // RtaoComputePass.cpp:529
    FrameParams fp = make_frame_params(c, io->ubo);
    const uint32_t curr = c->frame_parity;
    auto lo = [&](long halo) { long v = (long)y0 - halo; return (int)(v < 0 ? 0 : v); };
    auto hi = [&](long halo) { long v = (long)y1 + halo; return (int)(v > H ? H : v); };
    // halo needed below iteration i: taps reach 2*step rows, so rows of the source must extend by
    // sum_{k>=i} 2*2^k = 2*(2^n - 2^i)
    const long halo_all = 2L * ((1L << n) - 1L);
    // dispatchTrace (RtaoComputePass.cpp:503-517)
    HIP_OR_FAIL(c, rtao_trace_launch(c, fp, io->linear_depth_r16f, c->d_hist_img[curr ^ 1u].p, c->d_hist_img[curr].p,
                                     io->ray_hit_mask, lo(halo_all), hi(halo_all), st));
    HIP_OR_FAIL(c, rtao_prep_launch(c, io->linear_depth_r16f, io->normal_rgba16f, lo(halo_all), hi(halo_all), st));
    // dispatchAtrous (RtaoComputePass.cpp:519-554); the final iteration also writes rtao_full (K3 folded in)
    const bool fold_upsample = (fp.debugMode != 9u);
    for (uint32_t i = 0; i < n; ++i) {
        const uint16_t* src = (i == 0) ? c->d_hist_img[curr].p : c->d_ping[(i - 1u) & 1u].p;
        const long halo_out = 2L * ((1L << n) - (1L << (i + 1)));
        const bool last = (i + 1 == n);
        // the rtao_full copy must cover only [y0,y1); with last => halo_out == 0 this holds
        HIP_OR_FAIL(c, rtao_atrous_launch(c, fp.radiusRaw, 1 << i, src, c->d_ping[i & 1u].p,
                                          (last && fold_upsample) ? io->rtao_full_r16f : nullptr, lo(halo_out),
                                          hi(halo_out), st));
    }
    // dispatchUpsample (RtaoComputePass.cpp:556-571), only needed as a separate pass for debug view 9
    if (!fold_upsample)
        HIP_OR_FAIL(c, rtao_upsample_launch(c, fp.debugMode, c->d_ping[(n - 1u) & 1u].p, io->normal_rgba16f,
                                            io->rtao_full_r16f, (int)y0, (int)y1, st));
    vkr_rtao_end_frame(c);
    return VKR_OK;
}

// Two-phase sharding (multi-GPU, no reference counterpart): K1 on interleaved 8-row stripes, then -- after the caller
// has all-gathered history[curr] -- A-Trous/upsample on a contiguous band whose halo rows come from the gathered image.
int vkr_rtao_trace_stripes(vkr_ctx* c, const vkr_rtao_frame_io* io, uint32_t rank, uint32_t nranks, void* stream)
{
    if (int rc = check_io(c, io, false, false)) return rc;
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "vkr_rtao_trace_stripes before vkr_bvh_build");
    if (nranks == 0 || rank >= nranks) return fail(c, VKR_ERR_INVALID_ARG, "rank %u of %u", rank, nranks);
    if (int rc = set_device(c)) return rc;
    FrameParams fp = make_frame_params(c, io->ubo);
    fp.stripeMul = (int)nranks;
    const uint32_t curr = c->frame_parity;
    const int y0 = (int)(rank * 8u);
    if (y0 >= (int)c->H) return VKR_OK;                      // more ranks than stripes: nothing to trace
    HIP_OR_FAIL(c, rtao_trace_launch(c, fp, io->linear_depth_r16f, c->d_hist_img[curr ^ 1u].p, c->d_hist_img[curr].p,
                                     io->ray_hit_mask, y0, (int)c->H, (hipStream_t)stream));
    return VKR_OK;
}

int vkr_rtao_denoise_band(vkr_ctx* c, const vkr_rtao_frame_io* io, uint32_t y0, uint32_t y1, void* stream)
{
    if (int rc = check_io(c, io, true, true)) return rc;
    if (y0 > y1 || y1 > c->H) return fail(c, VKR_ERR_INVALID_ARG, "band [%u,%u) outside image height %u", y0, y1, c->H);
    if (io->atrous_iterations > 16u) return fail(c, VKR_ERR_INVALID_ARG, "atrous_iterations %u > 16", io->atrous_iterations);
    if (int rc = set_device(c)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int H = (int)c->H;
    const uint32_t n = io->atrous_iterations < 1u ? 1u : io->atrous_iterations;
    FrameParams fp = make_frame_params(c, io->ubo);
    const uint32_t curr = c->frame_parity;
    if (y0 < y1) {
        auto lo = [&](long halo) { long v = (long)y0 - halo; return (int)(v < 0 ? 0 : v); };
        auto hi = [&](long halo) { long v = (long)y1 + halo; return (int)(v > H ? H : v); };
        const long halo_all = 2L * ((1L << n) - 1L);
        HIP_OR_FAIL(c, rtao_prep_launch(c, io->linear_depth_r16f, io->normal_rgba16f, lo(halo_all), hi(halo_all), st));
        const bool fold_upsample = (fp.debugMode != 9u);
        for (uint32_t i = 0; i < n; ++i) {
            const uint16_t* src = (i == 0) ? c->d_hist_img[curr].p : c->d_ping[(i - 1u) & 1u].p;
            const long halo_out = 2L * ((1L << n) - (1L << (i + 1)));
            const bool last = (i + 1 == n);
            HIP_OR_FAIL(c, rtao_atrous_launch(c, fp.radiusRaw, 1 << i, src, c->d_ping[i & 1u].p,
                                              (last && fold_upsample) ? io->rtao_full_r16f : nullptr, lo(halo_out),
                                              hi(halo_out), st));
        }
        if (!fold_upsample)
            HIP_OR_FAIL(c, rtao_upsample_launch(c, fp.debugMode, c->d_ping[(n - 1u) & 1u].p, io->normal_rgba16f,
                                                io->rtao_full_r16f, (int)y0, (int)y1, st));
    }
    vkr_rtao_end_frame(c);
    return VKR_OK;
}

int vkr_rtao_execute(vkr_ctx* c, const vkr_rtao_frame_io* io, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    return vkr_rtao_execute_band(c, io, 0, c->H, stream);
}

// ------------------------------------------------------------------------------------------------ shadows
int vkr_shadow_sun(vkr_ctx* c, const float* world_pos, const float* geom_normal, const uint8_t* valid,
                   uint32_t pixel_count, const float L[3], float sun_radius, float sample_count, float* out_vis,
                   void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!world_pos || !geom_normal || !L || !out_vis) return fail(c, VKR_ERR_INVALID_ARG, "vkr_shadow_sun: NULL argument");
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "vkr_shadow_sun before vkr_bvh_build");
    if (int rc = set_device(c)) return rc;
    HIP_OR_FAIL(c, shadow_sun_launch(c, world_pos, geom_normal, valid, pixel_count, L, sun_radius, sample_count, out_vis,
                                     (hipStream_t)stream));
    return VKR_OK;
}

int vkr_shadow_sun_noalloc(vkr_ctx* c, const float* world_pos, const float* geom_normal, const uint8_t* valid,
                           uint32_t pixel_count, const float L[3], float sun_radius, float sample_count, float* out_vis,
                           void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!world_pos || !geom_normal || !L || !out_vis) return fail(c, VKR_ERR_INVALID_ARG, "vkr_shadow_sun_noalloc: NULL argument");
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "vkr_shadow_sun_noalloc before vkr_bvh_build");
    if (pixel_count > VKR_NOALLOC_MAX_PIXELS)
        return fail(c, VKR_ERR_LIMIT, "vkr_shadow_sun_noalloc: %u pixels exceed the maximum of %u (split the call, or use vkr_shadow_sun)",
                    pixel_count, VKR_NOALLOC_MAX_PIXELS);
    if (c->wide_count > (1u << vkr::KQ_NODE_BITS))
        return fail(c, VKR_ERR_LIMIT, "vkr_shadow_sun_noalloc: the BVH has %u wide nodes, the packet kernel's items hold 2^%u (use vkr_shadow_sun)",
                    c->wide_count, vkr::KQ_NODE_BITS);
    if (int rc = set_device(c)) return rc;
    HIP_OR_FAIL(c, shadow_sun_noalloc_launch(c, world_pos, geom_normal, valid, pixel_count, L, sun_radius, sample_count, out_vis,
                                             (hipStream_t)stream));
    return VKR_OK;
}

int vkr_shadow_point(vkr_ctx* c, const float* world_pos, const float* geom_normal, const uint8_t* valid,
                     uint32_t pixel_count, const float P[3], float* out_vis, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!world_pos || !geom_normal || !P || !out_vis) return fail(c, VKR_ERR_INVALID_ARG, "vkr_shadow_point: NULL argument");
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "vkr_shadow_point before vkr_bvh_build");
    if (int rc = set_device(c)) return rc;
    HIP_OR_FAIL(c, shadow_point_launch(c, world_pos, geom_normal, valid, pixel_count, P, out_vis, (hipStream_t)stream));
    return VKR_OK;
}

int vkr_gbuffer_primary(vkr_ctx* c, const float ubo[VKR_UBO_FLOATS], uint16_t* depth, uint16_t* normal,
                        float* world_pos, float* geom_normal, uint8_t* valid, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!ubo) return fail(c, VKR_ERR_INVALID_ARG, "ubo is NULL");
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "vkr_gbuffer_primary before vkr_bvh_build");
    if (int rc = set_device(c)) return rc;
    FrameParams fp = make_frame_params(c, ubo);
    HIP_OR_FAIL(c, gbuffer_launch(c, fp, depth, normal, world_pos, geom_normal, valid, (hipStream_t)stream));
    return VKR_OK;
}

int vkr_gbuffer_primary_msaa(vkr_ctx* c, const float ubo[VKR_UBO_FLOATS], uint32_t samples, uint16_t* depth, uint16_t* normal,
                             float* world_pos, float* geom_normal, uint8_t* valid, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!ubo) return fail(c, VKR_ERR_INVALID_ARG, "ubo is NULL");
    if (samples != 1u && samples != 2u && samples != 4u && samples != 8u)
        return fail(c, VKR_ERR_INVALID_ARG, "samples must be 1, 2, 4 or 8 (got %u)", samples);
    if (!c->has_bvh) return fail(c, VKR_ERR_STATE, "vkr_gbuffer_primary_msaa before vkr_bvh_build");
    if (int rc = set_device(c)) return rc;
    FrameParams fp = make_frame_params(c, ubo);
    HIP_OR_FAIL(c, gbuffer_msaa_launch(c, fp, (int)samples, depth, normal, world_pos, geom_normal, valid, (hipStream_t)stream));
    return VKR_OK;
}

// ------------------------------------------------------------------------------------------------ counters
int vkr_trace_stats_enable(vkr_ctx* c, int enable)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    c->stats_enabled = enable != 0;
    return VKR_OK;
}

int vkr_trace_stats_get(vkr_ctx* c, vkr_trace_stats* out, int reset)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!out) return fail(c, VKR_ERR_INVALID_ARG, "out is NULL");
    if (int rc = set_device(c)) return rc;
    HIP_OR_FAIL(c, hipDeviceSynchronize());
    unsigned long long h[24];
    HIP_OR_FAIL(c, hipMemcpy(h, c->d_stats.p, 192, hipMemcpyDeviceToHost));
    if (getenv("VKR_PHASE_DUMP")) {   // development: per-phase shader cycles of the K1 STATS build, summed over waves (k_trace_q)
        static const char* names[8] = {"raygen", "inject", "node_steps", "leaf_steps", "alpha_steps", "retry_list", "resolve", "tile_pull"};
        fprintf(stderr, "{\"k1_phase_cycles\": {");
        for (int k = 0; k < 8; ++k) fprintf(stderr, "\"%s\": %llu, ", names[k], h[8 + k]);
        fprintf(stderr, "\"n_node_steps\": %llu, \"n_leaf_steps\": %llu, \"n_alpha_steps\": %llu, \"node_items\": %llu, \"waves\": %llu, \"spilled_items\": %llu, \"setup_pixel\": %llu, \"visibility_loop\": %llu, \"bound_violation_bits\": %llu, \"cut_box_tests\": %llu, \"cut_hits\": %llu}}\n", h[16], h[17], h[18], h[19], h[20], h[5], h[21], h[22], h[23], h[6], h[7]);
    }
    out->overflow_tiles = h[5];
    out->rays = h[0]; out->nodes_visited = h[1]; out->tris_tested = h[2];
    out->wave_node_iters = h[3]; out->wave_tri_iters = h[4];
    if (reset) HIP_OR_FAIL(c, hipMemset(c->d_stats.p, 0, 192));
    return VKR_OK;
}

}  // extern "C"
