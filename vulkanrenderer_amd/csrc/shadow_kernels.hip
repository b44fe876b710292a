// shadow_kernels.hip -- K4 sun-disc Poisson shadow rays, K5 point-light shadow ray (pbr.frag:169-260 lifted from
// per-fragment to per-pixel), and the primary-ray G-buffer producer that stands in for DepthPrepass.
#include "internal.hpp"
#include <mutex>

namespace vkr {

__constant__ float kPoisson[16][2] = {  // pbr.frag:244-249 (data)
    {-0.613f, -0.358f}, {0.582f, -0.396f}, {-0.451f, 0.552f}, {0.208f, 0.623f},
    {-0.263f, -0.826f}, {-0.883f, 0.217f}, {0.759f, 0.274f},  {0.352f, -0.721f},
    {-0.713f, -0.045f}, {0.039f, -0.956f}, {-0.189f, 0.291f}, {0.595f, 0.645f},
    {-0.515f, 0.718f},  {0.941f, -0.135f}, {-0.137f, -0.535f}, {0.169f, 0.892f}};

// pbr.frag:199-215
VKR_DI float TraceShadowRayDirectional(const SceneDev& sc, V3 o, V3 d, uint32_t* stack)
{
    return traverse<Q_SHADOW, false>(sc, o, d, 0.001f, 500.0f, nullptr, stack, nullptr) ? 0.0f : 1.0f;
}

// pbr.frag:217-260
__global__ __launch_bounds__(VKR_TRACE_BLOCK) void k_shadow_sun(SceneDev sc, const float* wp, const float* ng,
                                                                const uint8_t* valid, uint32_t count, V3 L,
                                                                float sunAngularRadius, float sampleCountF,
                                                                float* out)
{
    __shared__ uint32_t s_stack[2 * VKR_LDS_STACK * VKR_TRACE_BLOCK];
    uint32_t* stack = s_stack + threadIdx.x;
    uint32_t i = blockIdx.x * VKR_TRACE_BLOCK + threadIdx.x;
    if (i >= count) return;
    if (valid && !valid[i]) { out[i] = 1.0f; return; }
    V3 worldPos = V3{wp[3 * (size_t)i], wp[3 * (size_t)i + 1], wp[3 * (size_t)i + 2]};
    V3 geomNormal = V3{ng[3 * (size_t)i], ng[3 * (size_t)i + 1], ng[3 * (size_t)i + 2]};
    float dirLen2 = dot3(L, L);                                      // :219-222
    if (dirLen2 < 1e-8f) { out[i] = 1.0f; return; }
    V3 rayDir = normalize3(L);                                       // :223
    float bias = 0.002f;                                             // :224
    V3 rayOrigin = worldPos + geomNormal * bias;                     // :225
    int sampleCount = iclamp((int)(sampleCountF + 0.5f), 1, 16);     // :228-229
    if (sunAngularRadius <= 0.0f || sampleCount <= 1) {              // :232-235
        float vis = TraceShadowRayDirectional(sc, rayOrigin, rayDir, stack);
        out[i] = vis > 0.5f ? 1.0f : 0.2f;
        return;
    }
    V3 up = (__builtin_fabsf(rayDir.y) < 0.999f) ? v3(0, 1, 0) : v3(0, 0, 1);  // :239
    V3 T = normalize3(cross3(up, rayDir));                           // :240
    V3 B = cross3(rayDir, T);                                        // :241
    float visSum = 0.0f;
    for (int k = 0; k < sampleCount; ++k) {                          // :252-257
        float ux = kPoisson[k][0], uy = kPoisson[k][1];
        V3 offset = V3{sunAngularRadius * (ux * T.x + uy * B.x), sunAngularRadius * (ux * T.y + uy * B.y),
                       sunAngularRadius * (ux * T.z + uy * B.z)};
        V3 sampleDir = normalize3(rayDir + offset);
        visSum += TraceShadowRayDirectional(sc, rayOrigin, sampleDir, stack);
    }
    float vis = visSum / (float)sampleCount;                         // :258
    out[i] = mixf(0.0f, 1.0f, vis);                                  // :259
}

// pbr.frag:169-196
__global__ __launch_bounds__(VKR_TRACE_BLOCK) void k_shadow_point(SceneDev sc, const float* wp, const float* ng,
                                                                  const uint8_t* valid, uint32_t count, V3 lightPos,
                                                                  float* out)
{
    __shared__ uint32_t s_stack[2 * VKR_LDS_STACK * VKR_TRACE_BLOCK];
    uint32_t* stack = s_stack + threadIdx.x;
    uint32_t i = blockIdx.x * VKR_TRACE_BLOCK + threadIdx.x;
    if (i >= count) return;
    if (valid && !valid[i]) { out[i] = 1.0f; return; }
    V3 worldPos = V3{wp[3 * (size_t)i], wp[3 * (size_t)i + 1], wp[3 * (size_t)i + 2]};
    V3 geomNormal = V3{ng[3 * (size_t)i], ng[3 * (size_t)i + 1], ng[3 * (size_t)i + 2]};
    V3 toLight = lightPos - worldPos;                                // :171
    float lightDistance = length3(toLight);                          // :172
    if (lightDistance <= 0.0001f) { out[i] = 1.0f; return; }         // :173-175
    V3 rayDir = V3{toLight.x / lightDistance, toLight.y / lightDistance, toLight.z / lightDistance};  // :177
    float bias = gmax(0.002f, 0.0005f * lightDistance);              // :178
    V3 rayOrigin = worldPos + geomNormal * bias;                     // :179
    float tMin = 0.001f;                                             // :180
    float tMax = gmax(lightDistance - bias, tMin);                   // :181
    out[i] = traverse<Q_SHADOW, false>(sc, rayOrigin, rayDir, tMin, tMax, nullptr, stack, nullptr) ? 0.0f : 1.0f;
}

// transpose(inverse(mat3(model))) up to a positive factor (depth_prepass.vert:47; the result is normalised anyway):
// cofactors in double from the instance's 3x4 row-major transform, sign of the determinant applied, one rounding.
VKR_DI void normal_matrix(const float* xf, float* N)
{
    const double m[9] = {xf[0], xf[1], xf[2], xf[4], xf[5], xf[6], xf[8], xf[9], xf[10]};
    double c[9];
    c[0] = m[4] * m[8] - m[5] * m[7]; c[1] = m[5] * m[6] - m[3] * m[8]; c[2] = m[3] * m[7] - m[4] * m[6];
    c[3] = m[2] * m[7] - m[1] * m[8]; c[4] = m[0] * m[8] - m[2] * m[6]; c[5] = m[1] * m[6] - m[0] * m[7];
    c[6] = m[1] * m[5] - m[2] * m[4]; c[7] = m[2] * m[3] - m[0] * m[5]; c[8] = m[0] * m[4] - m[1] * m[3];
    const double det = (m[0] * c[0] + m[1] * c[1]) + m[2] * c[2];
    const double sg = det < 0.0 ? -1.0 : 1.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) N[k] = (float)(c[k] * sg);
}
VKR_DI V3 world_normal(const float* N, V3 n)
{
    return normalize3(V3{(N[0] * n.x + N[1] * n.y) + N[2] * n.z, (N[3] * n.x + N[4] * n.y) + N[5] * n.z,
                         (N[6] * n.x + N[7] * n.y) + N[8] * n.z});
}

// Primary visibility: what DepthPrepass leaves in linearDepthResolve / normalResolve
// (depth_prepass.vert:42-52 viewZ = -viewPos.z; depth_only.frag:31-34 normal*0.5+0.5, linear Z), plus the
// worldPos / Ng varyings pbr.frag shades with (:339-346).
// PKT: the 64 primary rays of a wave's 8x8 tile walk the tree as one packet (traverse_packet, traverse.hpp) -- they share
// their origin and nearly their path, so the node fetches, the stack and the control flow are paid once per tile instead of
// once per ray; per ray the result is the per-lane walk's, bit for bit.  !PKT (VKR_GBUFFER_MODE=lane): one walk per lane.
template <bool PKT>
__global__ __launch_bounds__(VKR_TRACE_BLOCK) void k_gbuffer(SceneDev sc, FrameParams fp, uint16_t* depth,
                                                             uint16_t* normal, float* wpOut, float* ngOut,
                                                             uint8_t* validOut, uint32_t tiles_x)
{
    __shared__ uint32_t s_stack[PKT ? (VKR_TRACE_BLOCK / 64) * VKR_MAX_STACK : 2 * VKR_LDS_STACK * VKR_TRACE_BLOCK];
    uint32_t* stack = PKT ? s_stack + (threadIdx.x >> 6) * VKR_MAX_STACK : s_stack + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (int)(blockIdx.x % tiles_x) * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = (int)(blockIdx.x / tiles_x) * 16 + (wave >> 1) * 8 + (lane >> 3);
    const int W = fp.W, H = fp.H;
    const bool inside = !(x >= W || y >= H);
    if (!PKT && !inside) return;                               // a packet needs all 64 lanes: outside lanes cast no ray
    size_t o = (size_t)y * W + x;
    float u = ((float)x + 0.5f) / (float)W, v = ((float)y + 0.5f) / (float)H;
    float fv[4];
    mul_m4_v4(fp.invProj, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 1.0f, 1.0f, fv);
    V3 dv = normalize3(V3{fv[0] / fv[3], fv[1] / fv[3], fv[2] / fv[3]});
    V3 d = normalize3(mul_m3_v3(fp.invView, dv));
    V3 cam = V3{fp.camPos[0], fp.camPos[1], fp.camPos[2]};
    HitDev h;
    bool hit;
    if (PKT) hit = traverse_packet<Q_PRIMARY>(sc, cam, d, inside, 0.0f, 1.0e30f, &h, stack);
    else hit = traverse<Q_PRIMARY, false>(sc, cam, d, 0.0f, 1.0e30f, &h, stack, nullptr);
    if (!inside) return;
    float z = 0.0f;
    V3 n = v3(0, 0, 1), wp = v3(0, 0, 0), ng = v3(0, 1, 0);
    if (hit) {
        wp = cam + d * h.t;
        float vp[4];
        mul_m4_v4(fp.view, wp.x, wp.y, wp.z, 1.0f, vp);
        z = -vp[2];
        const float4* tp = sc.tris + (size_t)h.tri * 3;
        float4 a = tp[0], b = tp[1], c = tp[2];
        const V3 nface = normalize3(cross3(V3{b.x, b.y, b.z}, V3{c.x, c.y, c.z}));
        n = nface;
        const uint32_t mesh = f2u(a.w) & 0x0fffffffu, prim = f2u(b.w), flat = f2u(c.w);
        if (sc.mesh_has_normals[mesh]) {
            // depth_prepass.vert:47-48 per vertex, rasteriser interpolation, depth_only.frag:31
            const uint32_t lo = sc.tri_inst[h.tri];        // instance with inst_first_tri[i] <= flat < inst_first_tri[i+1] (k_tri_instance)
            (void)flat;
            float N[9];
            normal_matrix(sc.inst_xf + 12 * (size_t)lo, N);
            const uint32_t ioff = sc.lut[mesh].y + 3u * prim;
            V3 vn[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t gi = sc.indices[ioff + k];
                vn[k] = world_normal(N, V3{sc.normals[3 * (size_t)gi], sc.normals[3 * (size_t)gi + 1], sc.normals[3 * (size_t)gi + 2]});
            }
            const float w0 = (1.0f - h.u) - h.v;
            n = normalize3(V3{(vn[0].x * w0 + vn[1].x * h.u) + vn[2].x * h.v, (vn[0].y * w0 + vn[1].y * h.u) + vn[2].y * h.v,
                              (vn[0].z * w0 + vn[1].z * h.u) + vn[2].z * h.v});
        }
        // pbr.frag:339-346: Ng = interpolated normal, negated on back faces (gl_FrontFacing from the winding)
        ng = (dot3(nface, cam - wp) < 0.0f) ? neg(n) : n;
    }
    uint16_t zh = float_to_half(z);
    bool valid = hit && half_to_float(zh) > 1e-6f;
    if (depth) depth[o] = valid ? zh : (uint16_t)0;
    if (normal) {
        ushort4 e;
        if (valid) {
            e.x = float_to_half(n.x * 0.5f + 0.5f); e.y = float_to_half(n.y * 0.5f + 0.5f);
            e.z = float_to_half(n.z * 0.5f + 0.5f); e.w = float_to_half(1.0f);
        } else {  // clear value (0.5, 0.5, 1, 1), DepthPrepass.cpp:36
            e.x = float_to_half(0.5f); e.y = float_to_half(0.5f); e.z = float_to_half(1.0f); e.w = float_to_half(1.0f);
        }
        ((ushort4*)normal)[o] = e;
    }
    if (wpOut) { wpOut[3 * o] = wp.x; wpOut[3 * o + 1] = wp.y; wpOut[3 * o + 2] = wp.z; }
    if (ngOut) { ngOut[3 * o] = ng.x; ngOut[3 * o + 1] = ng.y; ngOut[3 * o + 2] = ng.z; }
    if (validOut) validOut[o] = valid ? 1 : 0;
}


// DepthPrepass with MSAA (DepthPrepass.cpp:31-65, both attachments resolve with eAverage; no sample shading,
// DepthPrepassPipeline.cpp:96): per sample a ray through the Vulkan standard sample position picks the closest
// non-discarded primitive; that primitive's fragment outputs are evaluated once, at the pixel centre (extrapolated on
// the triangle's plane when the centre is not covered), stored as fp16 per sample, averaged in float32 in sample order.
// Definition and rationale: oracle/rtao_oracle.cpp orc_gbuffer_msaa.  Only depth / normal / valid are written here; the
// pixel-centre outputs (world position, Ng) come from k_gbuffer.
__constant__ float c_sample_pos[4][8][2] = {
    {{0.5f, 0.5f}},
    {{0.75f, 0.75f}, {0.25f, 0.25f}},
    {{0.375f, 0.125f}, {0.875f, 0.375f}, {0.125f, 0.625f}, {0.625f, 0.875f}},
    {{0.5625f, 0.3125f}, {0.4375f, 0.6875f}, {0.8125f, 0.5625f}, {0.3125f, 0.1875f},
     {0.1875f, 0.8125f}, {0.0625f, 0.4375f}, {0.6875f, 0.9375f}, {0.9375f, 0.0625f}}};

VKR_DI V3 primary_dir(const FrameParams& fp, float fx, float fy)
{
    const float u = fx / (float)fp.W, v = fy / (float)fp.H;
    float fv[4];
    mul_m4_v4(fp.invProj, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 1.0f, 1.0f, fv);
    const V3 dv = normalize3(V3{fv[0] / fv[3], fv[1] / fv[3], fv[2] / fv[3]});
    return normalize3(mul_m3_v3(fp.invView, dv));
}

template <bool PKT>
__global__ __launch_bounds__(VKR_TRACE_BLOCK) void k_gbuffer_msaa(SceneDev sc, FrameParams fp, int set, uint16_t* depth,
                                                                  uint16_t* normal, uint8_t* validOut, uint32_t tiles_x)
{
    __shared__ uint32_t s_stack[PKT ? (VKR_TRACE_BLOCK / 64) * VKR_MAX_STACK : 2 * VKR_LDS_STACK * VKR_TRACE_BLOCK];
    uint32_t* stack = PKT ? s_stack + (threadIdx.x >> 6) * VKR_MAX_STACK : s_stack + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (int)(blockIdx.x % tiles_x) * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = (int)(blockIdx.x / tiles_x) * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = !(x >= fp.W || y >= fp.H);
    if (!PKT && !inside) return;                               // a packet needs all 64 lanes: outside lanes cast no ray
    const size_t o = (size_t)y * fp.W + x;
    const int S = 1 << set;
    const V3 cam = V3{fp.camPos[0], fp.camPos[1], fp.camPos[2]};
    const V3 dc = primary_dir(fp, (float)x + 0.5f, (float)y + 0.5f);
    float zsum = 0.0f, n0 = 0.0f, n1 = 0.0f, n2 = 0.0f, n3 = 0.0f;
    for (int k = 0; k < S; ++k) {
        const V3 ds = primary_dir(fp, (float)x + c_sample_pos[set][k][0], (float)y + c_sample_pos[set][k][1]);
        HitDev h;
        uint16_t zh = 0, e0 = float_to_half(0.5f), e1h = float_to_half(0.5f), e2h = float_to_half(1.0f), e3 = float_to_half(1.0f);
        const bool got = PKT ? traverse_packet<Q_PRIMARY_MS>(sc, cam, ds, inside, 0.0f, 1.0e30f, &h, stack, dc)
                             : traverse<Q_PRIMARY_MS, false>(sc, cam, ds, 0.0f, 1.0e30f, &h, stack, nullptr, dc);
        if (got) {
            const float4* tp = sc.tris + (size_t)h.tri * 3;
            const float4 a = tp[0], b = tp[1], c = tp[2];
            const V3 v0 = V3{a.x, a.y, a.z}, e1 = V3{b.x, b.y, b.z}, e2 = V3{c.x, c.y, c.z};
            float t = h.t, u = h.u, v = h.v;
            V3 dir = ds;
            const V3 pc = cross3(dc, e2);
            const float detc = dot3(e1, pc);
            if (detc != 0.0f) {                                   // centre ray on the triangle's plane
                const float invc = 1.0f / detc;
                const V3 tvec = cam - v0;
                const V3 qvec = cross3(tvec, e1);
                u = dot3(tvec, pc) * invc;
                v = dot3(dc, qvec) * invc;
                t = dot3(e2, qvec) * invc;
                dir = dc;
            }
            const V3 wp = cam + dir * t;
            float vp[4];
            mul_m4_v4(fp.view, wp.x, wp.y, wp.z, 1.0f, vp);
            const float z = -vp[2];
            V3 n = normalize3(cross3(e1, e2));
            const uint32_t mesh = f2u(a.w) & 0x0fffffffu, prim = f2u(b.w), flat = f2u(c.w);
            if (sc.mesh_has_normals[mesh]) {
                const uint32_t lo = sc.tri_inst[h.tri];
                (void)flat;
                float N[9];
                normal_matrix(sc.inst_xf + 12 * (size_t)lo, N);
                const uint32_t ioff = sc.lut[mesh].y + 3u * prim;
                V3 vn[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const uint32_t gi = sc.indices[ioff + q];
                    vn[q] = world_normal(N, V3{sc.normals[3 * (size_t)gi], sc.normals[3 * (size_t)gi + 1], sc.normals[3 * (size_t)gi + 2]});
                }
                const float w0 = (1.0f - u) - v;
                n = normalize3(V3{(vn[0].x * w0 + vn[1].x * u) + vn[2].x * v, (vn[0].y * w0 + vn[1].y * u) + vn[2].y * v,
                                  (vn[0].z * w0 + vn[1].z * u) + vn[2].z * v});
            }
            zh = float_to_half(z);
            e0 = float_to_half(n.x * 0.5f + 0.5f); e1h = float_to_half(n.y * 0.5f + 0.5f);
            e2h = float_to_half(n.z * 0.5f + 0.5f); e3 = float_to_half(1.0f);
        }
        zsum = zsum + half_to_float(zh);
        n0 = n0 + half_to_float(e0); n1 = n1 + half_to_float(e1h); n2 = n2 + half_to_float(e2h); n3 = n3 + half_to_float(e3);
    }
    if (!inside) return;
    const float invS = 1.0f / (float)S;
    const uint16_t zr = float_to_half(zsum * invS);
    if (depth) depth[o] = zr;
    if (normal) {
        ushort4 e;
        e.x = float_to_half(n0 * invS); e.y = float_to_half(n1 * invS); e.z = float_to_half(n2 * invS); e.w = float_to_half(n3 * invS);
        ((ushort4*)normal)[o] = e;
    }
    if (validOut) validOut[o] = half_to_float(zr) > 1e-6f ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// K4/K5, streaming form: the any-hit counterpart of k_trace_q (rtao_kernels.hip).  Persistent waves pull chunks
// of pixels; the chunk's rays live in two wave-wide LIFO queues in LDS (node items, leaf items) and every step pops
// up to 64 items of one kind.  A ray that has found an occluder (any triangle: gl_RayFlagsOpaqueEXT |
// TerminateOnFirstHit, pbr.frag:205-207) is flagged in an LDS bit set and its remaining items are dropped as they
// are popped.  Results are order-independent: "some triangle is hit inside (tMin, tMax)".
//   MODE 0 (sun)  : chunk = 64 pixels x n sample directions; the n directions are frame constants (pbr.frag:239-256
//                   has no per-pixel rotation), so a ray is (pixel origin in LDS, direction index)
//   MODE 1 (point): chunk = 256 pixels x 1 ray; per-ray records {origin|tMin, dir|tMax, 1/dir} in a global slice
// ---------------------------------------------------------------------------------------------
constexpr int SQ_POINT_PIX = 256;

struct ShadowArgs {
    const float* wp; const float* ng; const uint8_t* valid; uint32_t count;
    V3 L;                 // sun: direction to the light (caller-normalised input, normalised again like the shader)
    float sunRadius, sampleCountF;
    V3 lightPos;          // point
    float* out;
    unsigned long long* stats;   // development counters (-DVKR_SUN_STATS builds only), else unused
};

template <int MODE>
__global__ __launch_bounds__(64) void k_shadow_q(SceneDev sc, ShadowArgs sa, uint32_t n_chunks, float4* rayBuf,
                                                 uint32_t* chunk_counter, uint32_t width)
{
    // width != 0: the arrays are rows of `width` pixels and a chunk is 1 (sun) or SQ_POINT_PIX / 64 (point) neighbouring 8x8 tiles
    // instead of a run of pixels along a row (k_shadow_sun_pkt has the same mapping and the reason for it)
    const uint32_t tiles_x = (width + 7u) >> 3, rows = width ? sa.count / width : 0u;
    auto pixel_of = [&](uint32_t chunk, uint32_t p) -> uint32_t {           // index into the arrays, or >= sa.count: no such pixel
        const uint32_t per = MODE == 0 ? 64u : (uint32_t)SQ_POINT_PIX;
        if (width == 0u) return chunk * per + p;
        const uint32_t t = chunk * (per >> 6) + (p >> 6);
        const uint32_t x = (t % tiles_x) * 8u + (p & 7u), y = (t / tiles_x) * 8u + ((p >> 3) & 7u);
        return (x < width && y < rows) ? y * width + x : 0xffffffffu;
    };
    __shared__ uint32_t s_nq[KQ_NQ];
    __shared__ uint32_t s_lq[KQ_LQ];
    __shared__ uint16_t s_lqr[KQ_LQ];
    __shared__ uint32_t s_occ[32];          // one bit per ray of the chunk (<= 1024)
    __shared__ float4 s_org[64];            // sun: ray origin per pixel
    __shared__ float4 s_dir[16], s_idr[16]; // sun: sample directions and their reciprocals

    const int lane = threadIdx.x;
    const char* node_base = (const char*)sc.nodes;
    const char* tri_base = (const char*)sc.tris;
    float4* slice = rayBuf + (size_t)blockIdx.x * (SQ_POINT_PIX * 3);   // point mode only

    // ---- sun: frame constants (pbr.frag:219-229, 239-256)
    int n_samples = 1;
    bool hard = true, no_light = false;
    if (MODE == 0) {
        const V3 L = sa.L;
        no_light = dot3(L, L) < 1e-8f;                                           // :219-222
        const V3 rayDir = normalize3(L);                                         // :223
        n_samples = iclamp((int)(sa.sampleCountF + 0.5f), 1, 16);                // :228-229
        hard = (sa.sunRadius <= 0.0f || n_samples <= 1);                         // :232
        if (hard) n_samples = 1;
        if (lane < n_samples) {
            V3 d = rayDir;
            if (!hard) {
                V3 up = (__builtin_fabsf(rayDir.y) < 0.999f) ? v3(0, 1, 0) : v3(0, 0, 1);   // :239
                V3 T = normalize3(cross3(up, rayDir));                            // :240
                V3 B = cross3(rayDir, T);                                         // :241
                float ux = kPoisson[lane][0], uy = kPoisson[lane][1];
                V3 offset = V3{sa.sunRadius * (ux * T.x + uy * B.x), sa.sunRadius * (ux * T.y + uy * B.y),
                               sa.sunRadius * (ux * T.z + uy * B.z)};
                d = normalize3(rayDir + offset);                                  // :255
            }
            s_dir[lane] = make_float4(d.x, d.y, d.z, 0.0f);
            s_idr[lane] = make_float4(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z), 0.0f);
        }
    }

    for (;;) {
        uint32_t chunk = 0;
        if (lane == 0) chunk = atomicAdd(chunk_counter, 1u);
        chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
        if (chunk >= n_chunks) break;        // every wave reaches this: the counter only grows

        // ------------------------------------------------------------------ phase 1: rays of the chunk
        if (lane < 32) s_occ[lane] = 0u;
        uint32_t n_rays = 0;                 // rays of this chunk (wave-uniform)
        bool pvalid = false;                 // sun: this lane's pixel casts rays
        if (MODE == 0) {
            const uint32_t i = pixel_of(chunk, (uint32_t)lane);
            if (i < sa.count) {
                if ((sa.valid && !sa.valid[i]) || no_light) sa.out[i] = 1.0f;
                else {
                    V3 worldPos = V3{sa.wp[3 * (size_t)i], sa.wp[3 * (size_t)i + 1], sa.wp[3 * (size_t)i + 2]};
                    V3 geomNormal = V3{sa.ng[3 * (size_t)i], sa.ng[3 * (size_t)i + 1], sa.ng[3 * (size_t)i + 2]};
                    V3 rayOrigin = worldPos + geomNormal * 0.002f;                // :224-225
                    s_org[lane] = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, 0.0f);
                    pvalid = true;
                }
            }
            n_rays = 64u * (uint32_t)n_samples;          // ray r = sample * 64 + pixel lane; invalid pixels never inject
        } else {
            for (int j = 0; j < SQ_POINT_PIX / 64; ++j) {
                const uint32_t r = (uint32_t)j * 64u + (uint32_t)lane;
                const uint32_t i = pixel_of(chunk, r);
                float4 a = make_float4(0, 0, 0, -1.0f);   // tMin < 0 marks "no ray"
                float4 b = make_float4(0, 0, 1, 0), c = make_float4(0, 0, 0, 0);
                if (i < sa.count) {
                    if (sa.valid && !sa.valid[i]) sa.out[i] = 1.0f;
                    else {
                        V3 worldPos = V3{sa.wp[3 * (size_t)i], sa.wp[3 * (size_t)i + 1], sa.wp[3 * (size_t)i + 2]};
                        V3 geomNormal = V3{sa.ng[3 * (size_t)i], sa.ng[3 * (size_t)i + 1], sa.ng[3 * (size_t)i + 2]};
                        V3 toLight = sa.lightPos - worldPos;                      // :171
                        float lightDistance = length3(toLight);                   // :172
                        if (lightDistance <= 0.0001f) sa.out[i] = 1.0f;           // :173-175
                        else {
                            V3 rayDir = V3{toLight.x / lightDistance, toLight.y / lightDistance, toLight.z / lightDistance};  // :177
                            float bias = gmax(0.002f, 0.0005f * lightDistance);   // :178
                            V3 rayOrigin = worldPos + geomNormal * bias;          // :179
                            float tMin = 0.001f;                                  // :180
                            float tMax = gmax(lightDistance - bias, tMin);        // :181
                            a = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, tMin);
                            b = make_float4(rayDir.x, rayDir.y, rayDir.z, tMax);
                            c = make_float4(safe_rcp(rayDir.x), safe_rcp(rayDir.y), safe_rcp(rayDir.z), 0.0f);
                        }
                    }
                }
                slice[(size_t)r * 3u] = a; slice[(size_t)r * 3u + 1u] = b; slice[(size_t)r * 3u + 2u] = c;
            }
            n_rays = (uint32_t)SQ_POINT_PIX;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const unsigned long long pvmask = __ballot(pvalid);

        // ------------------------------------------------------------------ phase 2: queues
        if (sc.tri_count != 0u) {
            uint32_t next = 0, nq = 0, lq = 0;      // wave-uniform
            for (;;) {
                const bool no_more = (next >= n_rays);
                if (lq >= 64u || (lq > 0u && nq == 0u && no_more)) {
                    // ---------------- leaf step: any triangle inside (tMin, tMax) occludes (pbr.frag:199-215)
                    const uint32_t n = lq < 64u ? lq : 64u;
                    lq -= n;
                    uint32_t cntT = 0, first = 0, ray = 0;
                    V3 ro = v3(0, 0, 0), rdir = v3(0, 0, 1);
                    float tmin = 0.0f, tmax = 0.0f;
                    if ((uint32_t)lane < n) {
                        const uint32_t ref = s_lq[lq + (uint32_t)lane];
                        ray = (uint32_t)s_lqr[lq + (uint32_t)lane];
                        if (!((s_occ[ray >> 5] >> (ray & 31u)) & 1u)) {
                            first = leaf_ref_first(ref); cntT = leaf_ref_count(ref);
                            if (MODE == 0) {
                                const float4 o = s_org[ray & 63u], d = s_dir[ray >> 6];
                                ro = V3{o.x, o.y, o.z}; rdir = V3{d.x, d.y, d.z}; tmin = 0.001f; tmax = 500.0f;   // :203-204
                            } else {
                                const float4 a = slice[(size_t)ray * 3u], b = slice[(size_t)ray * 3u + 1u];
                                ro = V3{a.x, a.y, a.z}; rdir = V3{b.x, b.y, b.z}; tmin = a.w; tmax = b.w;
                            }
                        }
                    }
                    bool occluded = false;
                    for (uint32_t j = 0; j < (uint32_t)VKR_LEAF_MAX; ++j) {
                        if (__ballot(j < cntT && !occluded) == 0ull) break;
                        if (j < cntT && !occluded) {
                            const float4* tp = (const float4*)(tri_base + (size_t)(first + j) * 48u);
                            float u, v, t;
                            occluded = tri_hit(ro, rdir, tp[0], tp[1], tp[2], tmin, tmax, &u, &v, &t);
                        }
                    }
                    if (occluded) atomicOr(&s_occ[ray >> 5], 1u << (ray & 31u));
                    continue;
                }
                if (nq < 64u && next < n_rays) {
                    // ---------------- inject the next rays at the root
                    const uint32_t r = next + (uint32_t)lane;
                    bool inj = false;
                    if (r < n_rays) {
                        if (MODE == 0) inj = (pvmask >> (r & 63u)) & 1ull;
                        else inj = slice[(size_t)r * 3u].w >= 0.0f;
                    }
                    const unsigned long long im = __ballot(inj);
                    if (inj) s_nq[nq + mbcnt64(im)] = r << KQ_NODE_BITS;
                    nq += (uint32_t)__popcll(im);
                    next += 64u;
                    continue;
                }
                if (nq == 0u) { if (no_more) break; else continue; }
                // ---------------- node step: pops n, pushes <= 4n node items (and <= 4n leaf items, lq < 64 here)
                uint32_t n = nq < 64u ? nq : 64u;
                {
                    const uint32_t room = (uint32_t)KQ_NQ - KQ_GUARD - (nq < (uint32_t)KQ_NQ - KQ_GUARD ? nq : (uint32_t)KQ_NQ - KQ_GUARD);
                    const uint32_t fit = room / 3u;
                    if (n > fit) n = fit ? fit : 1u;
                }
                nq -= n;
                bool h0 = false, h1 = false, h2 = false, h3 = false;
                uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, ray = 0;
                if ((uint32_t)lane < n) {
                    const uint32_t item = s_nq[nq + (uint32_t)lane];
                    ray = item >> KQ_NODE_BITS;
                    if (!((s_occ[ray >> 5] >> (ray & 31u)) & 1u)) {
                        const uint32_t node = item & ((1u << KQ_NODE_BITS) - 1u);
                        const float4* np = (const float4*)(node_base + ((size_t)node << 6));
                        const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                        V3 ro, id;
                        float tmin, tb;
                        if (MODE == 0) {
                            const float4 o = s_org[ray & 63u], d = s_idr[ray >> 6];
                            ro = V3{o.x, o.y, o.z}; id = V3{d.x, d.y, d.z}; tmin = 0.001f; tb = 500.0f * 1.000001f;
                        } else {
                            const float4 a = slice[(size_t)ray * 3u], b = slice[(size_t)ray * 3u + 1u], c = slice[(size_t)ray * 3u + 2u];
                            ro = V3{a.x, a.y, a.z}; id = V3{c.x, c.y, c.z}; tmin = a.w; tb = b.w * 1.000001f;
                        }
                        // Slab tests on the quantised planes as in k_trace_q; these rays are long, so the interval gets
                        // the relative padding of box_key() on top (rounding may only add visits)
                        const float ax = q2.z * id.x, ay = q2.w * id.y, az = q0.w * id.z;   // scales are stored as floats
                        const float bx = (q0.x - ro.x) * id.x, by = (q0.y - ro.y) * id.y, bz = (q0.z - ro.z) * id.z;
                        const uint32_t nx4 = id.x >= 0.0f ? f2u(q1.x) : f2u(q1.w), fx4 = id.x >= 0.0f ? f2u(q1.w) : f2u(q1.x);
                        const uint32_t ny4 = id.y >= 0.0f ? f2u(q1.y) : f2u(q2.x), fy4 = id.y >= 0.0f ? f2u(q2.x) : f2u(q1.y);
                        const uint32_t nz4 = id.z >= 0.0f ? f2u(q1.z) : f2u(q2.y), fz4 = id.z >= 0.0f ? f2u(q2.y) : f2u(q1.z);
                        r0 = f2u(q3.x); r1 = f2u(q3.y); r2 = f2u(q3.z); r3 = f2u(q3.w);
#define VKR_PLANES(W, A, B, P01, P23)                                                                               \
    const f2v P01 = pk_fma(f2v{ubyte_f(W, 0), ubyte_f(W, 1)}, A, B), P23 = pk_fma(f2v{ubyte_f(W, 2), ubyte_f(W, 3)}, A, B);
                        VKR_PLANES(nx4, ax, bx, nx01, nx23) VKR_PLANES(ny4, ay, by, ny01, ny23) VKR_PLANES(nz4, az, bz, nz01, nz23)
                        VKR_PLANES(fx4, ax, bx, fx01, fx23) VKR_PLANES(fy4, ay, by, fy01, fy23) VKR_PLANES(fz4, az, bz, fz01, fz23)
#undef VKR_PLANES
#define VKR_SLABQ(HC, NX, NY, NZ, FX, FY, FZ)                                                                       \
    {                                                                                                               \
        float tn = __builtin_fmaxf(__builtin_fmaxf(NX, NY), NZ);                                                    \
        float tf = __builtin_fminf(__builtin_fminf(FX, FY), FZ);                                                    \
        tn = __builtin_fmaxf(tn - __builtin_fabsf(tn) * 1e-6f, tmin);                                               \
        tf = __builtin_fminf(tf + __builtin_fabsf(tf) * 1e-6f, tb);                                                 \
        HC = (tn <= tf);                                                                                            \
    }
                        VKR_SLABQ(h0, nx01.x, ny01.x, nz01.x, fx01.x, fy01.x, fz01.x)
                        VKR_SLABQ(h1, nx01.y, ny01.y, nz01.y, fx01.y, fy01.y, fz01.y)
                        VKR_SLABQ(h2, nx23.x, ny23.x, nz23.x, fx23.x, fy23.x, fz23.x)
                        VKR_SLABQ(h3, nx23.y, ny23.y, nz23.y, fx23.y, fy23.y, fz23.y)
#undef VKR_SLABQ
                    }
                }
#define VKR_APPEND(HC, RC)                                                                                          \
    {                                                                                                               \
        const bool isl = (int)RC < 0;                                                                               \
        const unsigned long long mn = __ballot(HC && !isl), ml = __ballot(HC && isl);                               \
        if (HC) {                                                                                                   \
            if (isl) { const uint32_t ix = lq + mbcnt64(ml); s_lq[ix] = RC; s_lqr[ix] = (uint16_t)ray; }            \
            else s_nq[nq + mbcnt64(mn)] = (ray << KQ_NODE_BITS) | RC;                                               \
        }                                                                                                           \
        nq += (uint32_t)__popcll(mn); lq += (uint32_t)__popcll(ml);                                                 \
    }
                VKR_APPEND(h0, r0) VKR_APPEND(h1, r1) VKR_APPEND(h2, r2) VKR_APPEND(h3, r3)
#undef VKR_APPEND
            }
        }

        // ------------------------------------------------------------------ phase 3: per pixel
        if (MODE == 0) {
            if (pvalid) {
                const uint32_t i = pixel_of(chunk, (uint32_t)lane);
                float visSum = 0.0f;
                for (int k = 0; k < n_samples; ++k) {                                // :252-257
                    const uint32_t r = (uint32_t)k * 64u + (uint32_t)lane;
                    visSum += ((s_occ[r >> 5] >> (r & 31u)) & 1u) ? 0.0f : 1.0f;
                }
                if (hard) sa.out[i] = visSum > 0.5f ? 1.0f : 0.2f;                   // :233-235
                else sa.out[i] = mixf(0.0f, 1.0f, visSum / (float)n_samples);        // :258-259
            }
        } else {
            for (int j = 0; j < SQ_POINT_PIX / 64; ++j) {
                const uint32_t r = (uint32_t)j * 64u + (uint32_t)lane;
                const uint32_t i = pixel_of(chunk, r);
                if (i < sa.count && slice[(size_t)r * 3u].w >= 0.0f)
                    sa.out[i] = ((s_occ[r >> 5] >> (r & 31u)) & 1u) ? 0.0f : 1.0f;   // :182-195
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K4, packet form (the default for the sun).  The n sun-disc directions are frame constants and differ by ~0.01 rad,
// so the n rays of a pixel share their origin and almost their path: ONE traversal per pixel with a conservative
// slab test over the whole direction set (interval arithmetic on 1/d per axis; an axis whose directions change sign
// is left unconstrained), and the n rays are told apart only at the leaves, where a triangle is tested against all
// n directions with the direction-independent parts of Moeller-Trumbore (tvec, qvec, dot(e2, qvec)) computed once.
// Per ray the arithmetic is that of tri_hit(), so each ray's any-hit answer is exactly the per-ray kernel's; the
// packet only changes which boxes are opened (a superset).  Same queue machinery as k_shadow_q; a pixel whose n rays
// are all occluded is dropped.  Node work falls ~n-fold, leaf work per hit triangle rises ~n/2-fold.
// ---------------------------------------------------------------------------------------------
#ifndef VKR_SP_PIX
#define VKR_SP_PIX 64
#endif
#ifndef VKR_SP_NQ
#define VKR_SP_NQ 768
#endif
#ifndef VKR_SUN_SLOT_ORDER
#define VKR_SUN_SLOT_ORDER 1   // which child slot the LIFO node queue pops first: 0 = slot 3 (round 2), 1 = slot 0, 2 = by the sun direction
// Measured at 4K / 16 samples over the eight sign combinations of the sun direction (tools/dbg/sun_dir_sweep.py,
// profiles/r3_sun_slot_order.txt): slot 0 first wins or ties in six (4.95 -> 3.01 ms for the reference's direction) and loses
// 4 % and 18 % in two; choosing by the signs of the direction (2) is no better than always slot 0.
#endif
constexpr int SP_PIX = VKR_SP_PIX;     // pixels per chunk at most (64, 128 or 256)
constexpr int SP_NQ = VKR_SP_NQ;
constexpr int SP_TQ = 64 + 64 * VKR_LEAF_MAX;   // a leaf step runs with < 64 triangle items queued and adds <= 64 * VKR_LEAF_MAX

struct SunFrame {                      // written by k_sun_setup, read with scalar loads
    float dir[16][4];                  // sample directions (pbr.frag:255), w unused
    float idlo[4], idhi[4];            // per axis: bounds of 1/d over the samples (valid where axis_ok)
    uint32_t axis_ok[4];               // all samples have the same non-zero sign on this axis
    uint32_t n, hard, no_light, slot0_first;   // slot0_first: open the child slots in Morton order (else in reverse)
    float axis[4];                     // cone around the samples: unit axis (= normalize(L)), w = tan of the half angle, padded
};

__global__ void k_sun_setup(V3 L, float sunRadius, float sampleCountF, SunFrame* out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool no_light = dot3(L, L) < 1e-8f;                                    // :219-222
    const V3 rayDir = normalize3(L);                                             // :223
    int n = iclamp((int)(sampleCountF + 0.5f), 1, 16);                           // :228-229
    const bool hard = (sunRadius <= 0.0f || n <= 1);                             // :232
    if (hard) n = 1;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int k = 0; k < 16; ++k) {
        V3 d = rayDir;
        if (!hard && k < n) {
            V3 up = (__builtin_fabsf(rayDir.y) < 0.999f) ? v3(0, 1, 0) : v3(0, 0, 1);   // :239
            V3 T = normalize3(cross3(up, rayDir));                                // :240
            V3 B = cross3(rayDir, T);                                             // :241
            float ux = kPoisson[k][0], uy = kPoisson[k][1];
            V3 offset = V3{sunRadius * (ux * T.x + uy * B.x), sunRadius * (ux * T.y + uy * B.y),
                           sunRadius * (ux * T.z + uy * B.z)};
            d = normalize3(rayDir + offset);                                      // :255
        }
        out->dir[k][0] = d.x; out->dir[k][1] = d.y; out->dir[k][2] = d.z; out->dir[k][3] = 0.0f;
        if (k < n) {
            const float c[3] = {d.x, d.y, d.z};
            for (int a = 0; a < 3; ++a) { lo[a] = __builtin_fminf(lo[a], c[a]); hi[a] = __builtin_fmaxf(hi[a], c[a]); }
        }
    }
    for (int a = 0; a < 3; ++a) {
        const bool ok = (lo[a] > 0.0f) || (hi[a] < 0.0f);       // one sign, never zero
        out->axis_ok[a] = ok ? 1u : 0u;
        // 1/d is decreasing on either side of zero: bounds are the reciprocals of the other end, widened a little
        const float a0 = ok ? 1.0f / hi[a] : 0.0f, a1 = ok ? 1.0f / lo[a] : 0.0f;
        const float l = __builtin_fminf(a0, a1), h = __builtin_fmaxf(a0, a1);
        out->idlo[a] = l - __builtin_fabsf(l) * 1e-6f;
        out->idhi[a] = h + __builtin_fabsf(h) * 1e-6f;
    }
    out->axis_ok[3] = 0; out->idlo[3] = 0; out->idhi[3] = 0;
    out->n = (uint32_t)n; out->hard = hard ? 1u : 0u; out->no_light = no_light ? 1u : 0u;
    // Morton order runs from low to high coordinates: front to back for a direction whose components are mostly positive
    const int negs = (rayDir.x < 0.0f ? 1 : 0) + (rayDir.y < 0.0f ? 1 : 0) + (rayDir.z < 0.0f ? 1 : 0);
    out->slot0_first = (VKR_SUN_SLOT_ORDER == 2) ? (negs <= 1 ? 1u : 0u) : (uint32_t)VKR_SUN_SLOT_ORDER;
    // bounding cone of the n directions (leaf pre-filter): tan of the largest angle to the axis, from the perpendicular part
    float tan_max = 0.0f;
    for (int k = 0; k < n; ++k) {
        const V3 d = V3{out->dir[k][0], out->dir[k][1], out->dir[k][2]};
        const float c = dot3(d, rayDir);
        const V3 pp = d - rayDir * c;
        const float t = (c > 1e-3f) ? __builtin_sqrtf(dot3(pp, pp)) / c : 3.0e38f;
        tan_max = __builtin_fmaxf(tan_max, t);
    }
    out->axis[0] = rayDir.x; out->axis[1] = rayDir.y; out->axis[2] = rayDir.z;
    out->axis[3] = tan_max * 1.001f + 1e-6f;
}

#ifndef VKR_SUN_WG
#define VKR_SUN_WG 4       // waves per k_shadow_sun_pkt workgroup: they take neighbouring chunks at the same time and share the CU's L1 (1 = single waves)
#endif
constexpr uint32_t SUN_RING = 16;          // VKR_SUN_WG > 1: units a workgroup keeps published at a time

// A workgroup of WG waves takes UNITS of WG consecutive work items from a global counter, without a barrier: a wave takes the next
// (unit, slot) of its workgroup from an LDS counter; the wave that gets slot 0 fetches the unit from the global counter and
// publishes it in a ring of LDS entries {tag, unit, readers}, the others wait for that entry (its publisher is resident and
// running, so the wait is bounded).  An entry is reused RING units later; its publisher first waits until the WG readers of the
// previous use have counted themselves in.  Returns false once the counter is drained (every later take of the workgroup says so
// too).  s_take: 4 + 4 * RING words, [0] = 0 and every tag = 0xffffffff before the first call (then __syncthreads).
template <uint32_t WG, uint32_t RING>
VKR_DI bool wg_take(uint32_t* s_take, uint32_t* counter, uint32_t n_units, int lane, uint32_t* unit_out, uint32_t* slot_out)
{
    uint32_t k = 0;
    if (lane == 0) k = atomicAdd(&s_take[0], 1u);
    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    const uint32_t j = k / WG, wslot = k % WG, ring = 4u + 4u * (j % RING);
    uint32_t unit = 0xfffffffeu;                                // nothing left
    if (lane == 0) {
        if (wslot == 0u) {
            const uint32_t u = atomicAdd(counter, 1u);
            if (u < n_units) unit = u;
            while (__hip_atomic_load(&s_take[ring + 2u], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (j / RING) * WG) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(&s_take[ring + 1u], unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&s_take[ring], j, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        while (__hip_atomic_load(&s_take[ring], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != j) __builtin_amdgcn_s_sleep(1);
        unit = __hip_atomic_load(&s_take[ring + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&s_take[ring + 2u], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
    *unit_out = unit; *slot_out = wslot;
    return unit != 0xfffffffeu;
}

__global__ __launch_bounds__(64 * VKR_SUN_WG) void k_shadow_sun_pkt(SceneDev sc, ShadowArgs sa, const SunFrame* __restrict__ sf,
                                                       uint32_t n_chunks, uint32_t pix_per_chunk, uint32_t* chunk_counter, uint32_t width)
{
    // width != 0: the arrays are rows of `width` pixels (count = rows * width) and a chunk is pix_per_chunk / 64 neighbouring 8x8
    // tiles instead of a run of pixels along a row -- origins centimetres apart in both directions walk the same nodes and
    // triangles, which is what the L1 hit rate of the node and triangle loads lives on.  Results per pixel do not depend on it.
    const uint32_t tiles_x = (width + 7u) >> 3, rows = width ? sa.count / width : 0u;
    auto pixel_of = [&](uint32_t chunk, uint32_t p) -> uint32_t {           // index into the arrays, or >= sa.count: no such pixel
        if (width == 0u) return chunk * pix_per_chunk + p;
        const uint32_t t = chunk * (pix_per_chunk >> 6) + (p >> 6);
        const uint32_t x = (t % tiles_x) * 8u + (p & 7u), y = (t / tiles_x) * 8u + ((p >> 3) & 7u);
        return (x < width && y < rows) ? y * width + x : 0xffffffffu;
    };
    // pix_per_chunk: 64, 128 or 256 (<= SP_PIX).  Small launches (row bands of a sharded frame) use small chunks so that
    // the persistent waves still get several chunks each; per-pixel results do not depend on it.
    // per wave (the waves of a workgroup share nothing but the chunk ring):
    __shared__ uint32_t s_nq_w[VKR_SUN_WG][SP_NQ];       // node items: pixel << 22 | node
    __shared__ uint32_t s_lq_w[VKR_SUN_WG][KQ_LQ];       // leaf items: leaf reference ...
    __shared__ uint8_t s_lqp_w[VKR_SUN_WG][KQ_LQ];       // ... and its pixel (index in the chunk: a byte; the LDS footprint decides how
                                                         // many waves a CU holds: 6 656 B per wave = 24 of them, 7 424 B were 22)
    static_assert(SP_PIX <= 256, "pixel indices are stored in bytes");
    __shared__ uint32_t s_mask_w[VKR_SUN_WG][SP_PIX];    // per pixel: bit k = sample k is occluded
    __shared__ float s_ox_w[VKR_SUN_WG][SP_PIX], s_oy_w[VKR_SUN_WG][SP_PIX], s_oz_w[VKR_SUN_WG][SP_PIX];   // per pixel: ray origin
    __shared__ uint32_t s_tq_w[VKR_SUN_WG][SP_TQ];       // triangle items (leaf triangles that passed the cone pre-filter): triangle ...
    __shared__ uint8_t s_tqp_w[VKR_SUN_WG][SP_TQ];       // ... and its pixel
    const uint32_t wv = VKR_SUN_WG > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;
    uint32_t* const s_nq = s_nq_w[wv]; uint32_t* const s_lq = s_lq_w[wv]; uint8_t* const s_lqp = s_lqp_w[wv];
    uint32_t* const s_mask = s_mask_w[wv]; float* const s_ox = s_ox_w[wv]; float* const s_oy = s_oy_w[wv]; float* const s_oz = s_oz_w[wv];
    uint32_t* const s_tq = s_tq_w[wv]; uint8_t* const s_tqp = s_tqp_w[wv];
#if VKR_SUN_WG > 1
    __shared__ uint32_t s_take[4u + 4u * SUN_RING];
    if (threadIdx.x < 4u + 4u * SUN_RING) s_take[threadIdx.x] = (threadIdx.x >= 4u && (threadIdx.x & 3u) == 0u) ? 0xffffffffu : 0u;
    __syncthreads();
#endif

    const int lane = VKR_SUN_WG > 1 ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : (int)threadIdx.x;
    const char* node_base = (const char*)sc.nodes;
    const char* tri_base = (const char*)sc.tris;
    const uint32_t n_dirs = sf->n;
    const uint32_t full = (n_dirs >= 32u) ? 0xffffffffu : ((1u << n_dirs) - 1u);
    const bool hard = sf->hard != 0u, no_light = sf->no_light != 0u, slot0_first = sf->slot0_first != 0u;
    const float tmin = 0.001f, tmax = 500.0f;                                    // :203-204
    const float tb = tmax * 1.000001f;
    const bool okx = sf->axis_ok[0] != 0u, oky = sf->axis_ok[1] != 0u, okz = sf->axis_ok[2] != 0u;
    const float ilx = sf->idlo[0], ihx = sf->idhi[0], ily = sf->idlo[1], ihy = sf->idhi[1], ilz = sf->idlo[2], ihz = sf->idhi[2];
    // near/far plane per axis by the (common) direction sign
    const bool posx = ihx > 0.0f, posy = ihy > 0.0f, posz = ihz > 0.0f;
#ifdef VKR_SUN_STATS
    unsigned long long st_node = 0, st_pre = 0, st_tri = 0, st_wnode = 0, st_wtri = 0, st_pix = 0;
#endif
    const V3 cone_axis = V3{sf->axis[0], sf->axis[1], sf->axis[2]};
    const float cone_tan = sf->axis[3];

    for (;;) {
        uint32_t chunk = 0;
#if VKR_SUN_WG > 1
        uint32_t unit, wslot;
        if (!wg_take<VKR_SUN_WG, SUN_RING>(s_take, chunk_counter, (n_chunks + VKR_SUN_WG - 1u) / VKR_SUN_WG, lane, &unit, &wslot)) break;
        chunk = unit * VKR_SUN_WG + wslot;
        if (chunk >= n_chunks) continue;     // the last unit may be short
#else
        if (lane == 0) chunk = atomicAdd(chunk_counter, 1u);
        chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
        if (chunk >= n_chunks) break;        // every wave reaches this: the counter only grows
#endif

        // ------------------------------------------------------------------ phase 1: origins
        uint32_t nq = 0, lq = 0, tq = 0;
        for (int j = 0; j < (int)(pix_per_chunk / 64u); ++j) {
            const uint32_t p = (uint32_t)j * 64u + (uint32_t)lane;
            const uint32_t i = pixel_of(chunk, p);
            bool live = false;
            if (i < sa.count) {
                if ((sa.valid && !sa.valid[i]) || no_light) sa.out[i] = 1.0f;
                else {
                    V3 worldPos = V3{sa.wp[3 * (size_t)i], sa.wp[3 * (size_t)i + 1], sa.wp[3 * (size_t)i + 2]};
                    V3 geomNormal = V3{sa.ng[3 * (size_t)i], sa.ng[3 * (size_t)i + 1], sa.ng[3 * (size_t)i + 2]};
                    V3 rayOrigin = worldPos + geomNormal * 0.002f;                // :224-225
                    s_ox[p] = rayOrigin.x; s_oy[p] = rayOrigin.y; s_oz[p] = rayOrigin.z;
                    live = true;
#ifdef VKR_SUN_STATS
                    st_pix++;
#endif
                }
            }
            s_mask[p] = live ? 0u : 0x80000000u;     // bit 31: no rays from this pixel
            const unsigned long long lm = __ballot(live && sc.tri_count != 0u);
            if (live && sc.tri_count != 0u) s_nq[nq + mbcnt64(lm)] = p << KQ_NODE_BITS;   // root
            nq += (uint32_t)__popcll(lm);
        }

        // ------------------------------------------------------------------ phase 2: queues
        for (;;) {
            if (tq >= 64u || (tq > 0u && lq == 0u && nq == 0u)) {
                // ---------------- triangle step: <= 64 (pixel, triangle) items, each against all n directions
                const uint32_t n = tq < 64u ? tq : 64u;
                tq -= n;
#ifdef VKR_SUN_STATS
                if (lane == 0) st_wtri++;
#endif
                VKR_PRIO_HI
                if ((uint32_t)lane < n) {
                    const uint32_t tri = s_tq[tq + (uint32_t)lane];
                    const uint32_t pix = (uint32_t)s_tqp[tq + (uint32_t)lane];
                    const uint32_t have = s_mask[pix];
                    if ((have & full) != full) {
#ifdef VKR_SUN_STATS
                        st_tri++;
#endif
                        const V3 ro = V3{s_ox[pix], s_oy[pix], s_oz[pix]};
                        const float4* tp = (const float4*)(tri_base + (size_t)tri * 48u);
                        const float4 ta = tp[0], tb4 = tp[1], tc = tp[2];
                        VKR_PRIO_LO
                        const V3 v0 = V3{ta.x, ta.y, ta.z}, e1 = V3{tb4.x, tb4.y, tb4.z}, e2 = V3{tc.x, tc.y, tc.z};
                        // tri_hit() with its direction-independent terms hoisted: same operations, same bits per ray
                        const V3 tvec = ro - v0;
                        const V3 qvec = cross3(tvec, e1);
                        const float te = dot3(e2, qvec);
                        uint32_t got = 0;
                        for (uint32_t k = 0; k < n_dirs; ++k) {
                            const V3 d = V3{sf->dir[k][0], sf->dir[k][1], sf->dir[k][2]};
                            const V3 pvec = cross3(d, e2);
                            const float det = dot3(e1, pvec);
                            const float inv = 1.0f / det;
                            const float u = dot3(tvec, pvec) * inv;
                            const float v = dot3(d, qvec) * inv;
                            const float t = te * inv;
                            const bool hit = (det != 0.0f) && (u >= 0.0f && u <= 1.0f) && (v >= 0.0f && u + v <= 1.0f) &&
                                             (t > tmin && t < tmax);
                            got |= hit ? (1u << k) : 0u;
                        }
                        VKR_PRIO_HI
                        if (got & ~have) atomicOr(&s_mask[pix], got);
                    }
                }
                continue;
            }
            if (lq >= 64u || (lq > 0u && nq == 0u)) {
                // ---------------- leaf step: the leaf's triangles against the bounding cone of the n directions; the ones
                // that may be hit become triangle items (the exact test then runs on full waves of them).  The filter is
                // conservative: a triangle is dropped only if its bounding sphere (centre c, radius r) lies outside the
                // cone {perp <= h+ * tan}: every point of the sphere has perp >= perp(c) - r and h <= h(c) + r.
                const uint32_t n = lq < 64u ? lq : 64u;
                lq -= n;
                uint32_t cntT = 0, first = 0, pix = 0;
                V3 ro = v3(0, 0, 0);
                VKR_PRIO_HI
                if ((uint32_t)lane < n) {
                    const uint32_t ref = s_lq[lq + (uint32_t)lane];
                    pix = (uint32_t)s_lqp[lq + (uint32_t)lane];
                    if ((s_mask[pix] & full) != full) {
                        first = leaf_ref_first(ref); cntT = leaf_ref_count(ref);
                        ro = V3{s_ox[pix], s_oy[pix], s_oz[pix]};
                    }
                }
                for (uint32_t j = 0; j < (uint32_t)VKR_LEAF_MAX; ++j) {
                    if (__ballot(j < cntT) == 0ull) break;
                    bool keep = false;
                    if (j < cntT) {
#ifdef VKR_SUN_STATS
                        st_pre++;
#endif
                        const float4* tp = (const float4*)(tri_base + (size_t)(first + j) * 48u);
                        const float4 ta = tp[0], tb4 = tp[1], tc = tp[2];
                        VKR_PRIO_LO
                        const V3 v0 = V3{ta.x, ta.y, ta.z}, e1 = V3{tb4.x, tb4.y, tb4.z}, e2 = V3{tc.x, tc.y, tc.z};
                        const V3 g = (e1 + e2) * (1.0f / 3.0f);                      // centroid - v0
                        const V3 a1 = e1 - g, a2 = e2 - g;
                        const float r2 = __builtin_fmaxf(dot3(g, g), __builtin_fmaxf(dot3(a1, a1), dot3(a2, a2)));
                        const float r = __builtin_amdgcn_sqrtf(r2) * 1.001f;
                        const V3 w = (v0 + g) - ro;
                        const float h = dot3(w, cone_axis);
                        const V3 wp = w - cone_axis * h;
                        const float perp2 = dot3(wp, wp);
                        const float lim = r + __builtin_fmaxf(h + r, 0.0f) * cone_tan +
                                          (__builtin_fabsf(h) + __builtin_fabsf(wp.x) + __builtin_fabsf(wp.y) + __builtin_fabsf(wp.z)) * 1e-5f + 1e-7f;
                        keep = !(perp2 > lim * lim);                                 // NaN keeps
                    }
                    VKR_PRIO_HI
                    const unsigned long long km = __ballot(keep);
                    if (keep) { const uint32_t ix = tq + mbcnt64(km); s_tq[ix] = first + j; s_tqp[ix] = (uint8_t)pix; }
                    tq += (uint32_t)__popcll(km);
                }
                continue;
            }
            if (nq == 0u) break;
            // ---------------- node step: pops n, pushes <= 4n node items (and <= 4n leaf items, lq < 64 here)
            uint32_t n = nq < 64u ? nq : 64u;
            {
                const uint32_t room = (uint32_t)SP_NQ - KQ_GUARD - (nq < (uint32_t)SP_NQ - KQ_GUARD ? nq : (uint32_t)SP_NQ - KQ_GUARD);
                const uint32_t fit = room / 3u;
                if (n > fit) n = fit ? fit : 1u;
            }
            nq -= n;
            // one basic block (see k_trace_q): lanes without an item, or whose pixel is already fully occluded, run on
            // their (or item 0's) addresses with an exit distance that fails every test
            const bool act = (uint32_t)lane < n;
            VKR_PRIO_HI
            const uint32_t item = act ? s_nq[nq + (uint32_t)lane] : 0u;
            const uint32_t pix = item >> KQ_NODE_BITS;
            const bool live = act && (s_mask[pix] & full) != full;
#ifdef VKR_SUN_STATS
            if (live) { st_node++; if ((int)(__ffsll((long long)__ballot(1)) - 1) == lane) st_wnode++; }
#endif
            const float tbx = live ? tb : -3.0e38f;
            const uint32_t node = item & ((1u << KQ_NODE_BITS) - 1u);
            const float4* np = (const float4*)(node_base + ((size_t)node << 6));
            const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
            const V3 o = V3{s_ox[pix], s_oy[pix], s_oz[pix]};
            VKR_PRIO_LO
            const float sx = q2.z, sy = q2.w, sz = q0.w;
            // plane - origin per axis: fma(q, s, node origin - ray origin)
            const float bx = q0.x - o.x, by = q0.y - o.y, bz = q0.z - o.z;
            const uint32_t nx4 = posx ? f2u(q1.x) : f2u(q1.w), fx4 = posx ? f2u(q1.w) : f2u(q1.x);
            const uint32_t ny4 = posy ? f2u(q1.y) : f2u(q2.x), fy4 = posy ? f2u(q2.x) : f2u(q1.y);
            const uint32_t nz4 = posz ? f2u(q1.z) : f2u(q2.y), fz4 = posz ? f2u(q2.y) : f2u(q1.z);
            const uint32_t r0 = f2u(q3.x), r1 = f2u(q3.y), r2 = f2u(q3.z), r3 = f2u(q3.w);
            // entry: the smallest near-plane distance any direction can have; exit: the largest far-plane one
#define VKR_PKT_AXIS(OK, NQW, FQW, C, S, B, IL, IH, TN, TF)                                                         \
    if (OK) {                                                                                                       \
        const float an = fma_(ubyte_f(NQW, C), S, B), af = fma_(ubyte_f(FQW, C), S, B);                             \
        TN = __builtin_fmaxf(TN, __builtin_fminf(an * IL, an * IH));                                                \
        TF = __builtin_fminf(TF, __builtin_fmaxf(af * IL, af * IH));                                                \
    } else {                                                                                                        \
        /* directions straddle zero on this axis: only "origin outside the slab on both sides" could cull */       \
    }
            // unused slots are stored inverted (near plane beyond far plane) and fail on any constrained axis; with all
            // three axes unconstrained nothing could cull them, so they are masked explicitly
#define VKR_PKT_SLAB(HC, C, RC)                                                                                     \
    bool HC;                                                                                                        \
    {                                                                                                               \
        float tn = -3.0e38f, tf = 3.0e38f;                                                                          \
        VKR_PKT_AXIS(okx, nx4, fx4, C, sx, bx, ilx, ihx, tn, tf)                                                    \
        VKR_PKT_AXIS(oky, ny4, fy4, C, sy, by, ily, ihy, tn, tf)                                                    \
        VKR_PKT_AXIS(okz, nz4, fz4, C, sz, bz, ilz, ihz, tn, tf)                                                    \
        tn = __builtin_fmaxf(tn - __builtin_fabsf(tn) * 1e-6f, tmin);                                               \
        tf = __builtin_fminf(tf + __builtin_fabsf(tf) * 1e-6f, tbx);                                                \
        HC = (tn <= tf) && (RC != VKR_CHILD_NONE);                                                                  \
    }
            VKR_PKT_SLAB(h0, 0, r0) VKR_PKT_SLAB(h1, 1, r1) VKR_PKT_SLAB(h2, 2, r2) VKR_PKT_SLAB(h3, 3, r3)
#undef VKR_PKT_SLAB
#undef VKR_PKT_AXIS
#define VKR_APPEND(HC, RC)                                                                                          \
    {                                                                                                               \
        const unsigned long long hm = __ballot(HC), lm = __ballot((int)RC < 0);                                     \
        const unsigned long long mn = hm & ~lm, ml = hm & lm;                                                       \
        if (HC) {                                                                                                   \
            if ((int)RC < 0) { const uint32_t ix = lq + mbcnt64(ml); s_lq[ix] = RC; s_lqp[ix] = (uint8_t)pix; }     \
            else s_nq[nq + mbcnt64(mn)] = (pix << KQ_NODE_BITS) | RC;                                               \
        }                                                                                                           \
        nq += (uint32_t)__popcll(mn); lq += (uint32_t)__popcll(ml);                                                 \
    }
            VKR_PRIO_HI
            // Any-hit rays end at the first occluder, so the order the children are opened in matters (round 2 lost 44 % of this
            // kernel when the builder put the child slots in Morton order and this step kept opening slot 3 first).  The queue is
            // LIFO, so the slot that should be opened first is appended last (wave-uniform branch on a constant of k_sun_setup).
            if (slot0_first) { VKR_APPEND(h3, r3) VKR_APPEND(h2, r2) VKR_APPEND(h1, r1) VKR_APPEND(h0, r0) }
            else { VKR_APPEND(h0, r0) VKR_APPEND(h1, r1) VKR_APPEND(h2, r2) VKR_APPEND(h3, r3) }
#undef VKR_APPEND
        }
        VKR_PRIO_LO

        // ------------------------------------------------------------------ phase 3: per pixel
        for (int j = 0; j < (int)(pix_per_chunk / 64u); ++j) {
            const uint32_t p = (uint32_t)j * 64u + (uint32_t)lane;
            const uint32_t i = pixel_of(chunk, p);
            const uint32_t m = s_mask[p];
            if (i < sa.count && !(m & 0x80000000u)) {
                const float visSum = (float)(n_dirs - (uint32_t)__popc(m & full));   // :252-257, a sum of 0/1 terms
                if (hard) sa.out[i] = visSum > 0.5f ? 1.0f : 0.2f;                   // :233-235
                else sa.out[i] = mixf(0.0f, 1.0f, visSum / (float)n_dirs);           // :258-259
            }
        }
    }
#ifdef VKR_SUN_STATS
    atomicAdd(&sa.stats[0], st_pix); atomicAdd(&sa.stats[1], st_node); atomicAdd(&sa.stats[2], st_pre);
    atomicAdd(&sa.stats[3], st_wnode); atomicAdd(&sa.stats[4], st_wtri); atomicAdd(&sa.stats[5], st_tri);
#endif
}

// The chunks of a shadow launch (sun packets, point-light / sun ray queues): arrays that are rows of the context's image (a whole frame, or a row band of it: count is a
// multiple of the width) are cut into 8x8 tiles, anything else into runs of `pix` pixels (VKR_SUN_TILES=0: always runs).
static void sun_chunks(const vkr_ctx* c, uint32_t count, uint32_t pix, uint32_t* width, uint32_t* n_chunks)
{
    *width = (c->sun_tiles && c->W >= 8u && count % c->W == 0u) ? c->W : 0u;
    const uint32_t tiles = *width ? ((*width + 7u) / 8u) * ((count / *width + 7u) / 8u) : 0u, per = pix / 64u;
    *n_chunks = *width ? (tiles + per - 1u) / per : (count + pix - 1u) / pix;
}

static hipError_t shadow_q_launch(vkr_ctx* c, int mode, const ShadowArgs& sa, hipStream_t st)
{
    hipError_t e;
    const uint32_t per = mode == 0 ? 64u : (uint32_t)SQ_POINT_PIX;
    uint32_t width, n_chunks;
    sun_chunks(c, sa.count, per, &width, &n_chunks);
    int& per_cu = c->shq_per_cu[mode == 0 ? 0 : 1];      // asked of the runtime once per context, not per call
    if (per_cu == 0) {
        if (mode == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_shadow_q<0>, 64, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_shadow_q<1>, 64, 0);
        if (e != hipSuccess) { per_cu = 0; return e; }
        if (per_cu < 1) per_cu = 1;
    }
    uint32_t blocks = (uint32_t)per_cu * (uint32_t)(c->num_cus > 0 ? c->num_cus : 256);
    if (blocks > n_chunks) blocks = n_chunks;
    const size_t need = mode == 0 ? 1 : (size_t)blocks * SQ_POINT_PIX * 3;
    // own scratch (chunk counter + ray slices): a shadow launch on a second stream may overlap K1, which keeps
    // d_tile_counter / d_raybuf for itself
    if (c->d_shq_raybuf.n < need) {
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        if ((e = c->d_shq_raybuf.alloc(need)) != hipSuccess) return e;
    }
    if (!c->d_shq_counter.p && (e = c->d_shq_counter.alloc(1)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(c->d_shq_counter.p, 0, 4, st)) != hipSuccess) return e;
    if (mode == 0) k_shadow_q<0><<<blocks, 64, 0, st>>>(scene_dev(c), sa, n_chunks, c->d_shq_raybuf.p, c->d_shq_counter.p, width);
    else k_shadow_q<1><<<blocks, 64, 0, st>>>(scene_dev(c), sa, n_chunks, c->d_shq_raybuf.p, c->d_shq_counter.p, width);
    return hipGetLastError();
}

// Grid of a sun-packet launch (c->sun_per_cu known): workgroups, pixels per chunk, chunk shape.
static void sun_grid(const vkr_ctx* c, uint32_t count, uint32_t* blocks, uint32_t* pix, uint32_t* width, uint32_t* n_chunks)
{
    // Wave slots per CU this persistent launch leaves free (4; VKR_SUN_RESERVE_WAVES).  A full launch holds 456 of the 512 VGPRs of
    // every SIMD until its last chunk is done: a kernel issued while it runs -- the A-Trous launches of the same frame on another
    // stream, a collective's kernel of a multi-GPU host -- would wait for exactly that.  Measured cost of the reserve: none.
    int waves = c->sun_per_cu * VKR_SUN_WG;
    if (c->sun_reserve_waves > 0) waves = waves > c->sun_reserve_waves ? waves - c->sun_reserve_waves : 1;
    const int per_cu = waves / VKR_SUN_WG > 0 ? waves / VKR_SUN_WG : 1;
    uint32_t b = (uint32_t)per_cu * (uint32_t)(c->num_cus > 0 ? c->num_cus : 256);
    uint32_t px = (uint32_t)SP_PIX;                           // aim for >= 4 chunks per resident wave
    while (px > 64u && (count + px - 1u) / px < 4u * b * VKR_SUN_WG) px >>= 1;
    sun_chunks(c, count, px, width, n_chunks);
    const uint32_t units = (*n_chunks + VKR_SUN_WG - 1u) / VKR_SUN_WG;
    *blocks = b > units ? units : b;
    *pix = px;
}

hipError_t shadow_sun_launch(vkr_ctx* c, const float* wp, const float* ng, const uint8_t* valid, uint32_t count,
                             const float L[3], float radius, float samples, float* out, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    if (c->trace_streaming && c->wide_count <= (1u << KQ_NODE_BITS)) {
        ShadowArgs sa{wp, ng, valid, count, V3{L[0], L[1], L[2]}, radius, samples, V3{0, 0, 0}, out, c->d_stats.p};
        if (c->sun_packets) {
            hipError_t e;
            if (!c->d_sun_frame.p && (e = c->d_sun_frame.alloc(sizeof(SunFrame))) != hipSuccess) return e;
            SunFrame* sf = (SunFrame*)c->d_sun_frame.p;
            k_sun_setup<<<1, 64, 0, st>>>(sa.L, radius, samples, sf);
            if (c->sun_per_cu == 0) {                         // asked of the runtime once per context, not per call
                if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->sun_per_cu, k_shadow_sun_pkt, 64 * VKR_SUN_WG, 0)) != hipSuccess) { c->sun_per_cu = 0; return e; }
                if (c->sun_per_cu < 1) c->sun_per_cu = 1;
            }
            uint32_t blocks, pix, width, n_chunks;
            sun_grid(c, count, &blocks, &pix, &width, &n_chunks);
            // its own chunk counter (and d_sun_frame): this launch may overlap vkr_rtao_* launches of another stream
            if (!c->d_sun_counter.p && (e = c->d_sun_counter.alloc(8)) != hipSuccess) return e;
            if ((e = hipMemsetAsync(c->d_sun_counter.p, 0, 4, st)) != hipSuccess) return e;
            k_shadow_sun_pkt<<<blocks, 64 * VKR_SUN_WG, 0, st>>>(scene_dev(c), sa, sf, n_chunks, pix, c->d_sun_counter.p, width);
            return hipGetLastError();
        }
        return shadow_q_launch(c, 0, sa, st);
    }
    k_shadow_sun<<<(count + VKR_TRACE_BLOCK - 1) / VKR_TRACE_BLOCK, VKR_TRACE_BLOCK, 0, st>>>(
        scene_dev(c), wp, ng, valid, count, V3{L[0], L[1], L[2]}, radius, samples, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// vkr_shadow_sun_noalloc: the packet form of the sun shadows for callers that cannot allocate (HIP graph capture, tight
// serving loops).  Its scratch -- the chunk counter of the persistent waves and the frame's SunFrame, which
// vkr_shadow_sun keeps in two buffers it hipMallocs on first use -- lives in a file-scope device array instead.
//   * `scratch` is a host-side stand-in, not a device address: the address each GPU holds its copy at comes from
//     hipGetSymbolAddress, once per device, at vkr_create (noalloc_scratch_init), and only that pointer is ever passed
//     to kernels or hipMemsetAsync.
//   * Every call on a GPU shares that GPU's array (all contexts of the device): the counter is cleared and the SunFrame
//     rewritten at the start of every call, and calls that overlap on one GPU race -- serialise them.
// ---------------------------------------------------------------------------------------------
constexpr size_t NOALLOC_COUNTER_OFF = 0, NOALLOC_FRAME_OFF = 256;
static_assert(NOALLOC_FRAME_OFF + sizeof(SunFrame) <= VKR_NOALLOC_SCRATCH_BYTES, "scratch holds the counter slot and one SunFrame");
static __device__ __align__(256) unsigned char scratch[VKR_NOALLOC_SCRATCH_BYTES];
static void* g_noalloc_scratch[VKR_NOALLOC_MAX_DEVICES] = {};    // per device: where that GPU holds `scratch`
static std::mutex g_noalloc_mutex;

hipError_t noalloc_scratch_init(int device)      // called by vkr_create with `device` current
{
    if (device < 0 || device >= VKR_NOALLOC_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (g_noalloc_scratch[device]) return hipSuccess;
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_noalloc_scratch[device] = p;
    return hipSuccess;
}

hipError_t shadow_sun_noalloc_launch(vkr_ctx* c, const float* wp, const float* ng, const uint8_t* valid, uint32_t count,
                                     const float L[3], float radius, float samples, float* out, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    void* base = (c->device >= 0 && c->device < VKR_NOALLOC_MAX_DEVICES) ? g_noalloc_scratch[c->device] : nullptr;
    if (!base) return hipErrorNotInitialized;
    uint32_t* counter = (uint32_t*)((unsigned char*)base + NOALLOC_COUNTER_OFF);
    SunFrame* sf = (SunFrame*)((unsigned char*)base + NOALLOC_FRAME_OFF);
    hipError_t e;
    ShadowArgs sa{wp, ng, valid, count, V3{L[0], L[1], L[2]}, radius, samples, V3{0, 0, 0}, out, nullptr};
    if ((e = hipMemsetAsync(counter, 0, 256, st)) != hipSuccess) return e;      // the last call left its chunk count here
    k_sun_setup<<<1, 64, 0, st>>>(sa.L, radius, samples, sf);
    if (c->sun_per_cu == 0) {                                 // normally known since vkr_bvh_build (shadow_prepare)
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->sun_per_cu, k_shadow_sun_pkt, 64 * VKR_SUN_WG, 0)) != hipSuccess) { c->sun_per_cu = 0; return e; }
        if (c->sun_per_cu < 1) c->sun_per_cu = 1;
    }
    uint32_t blocks, pix, width, n_chunks;                   // same grid and chunking as shadow_sun_launch
    sun_grid(c, count, &blocks, &pix, &width, &n_chunks);
    k_shadow_sun_pkt<<<blocks, 64 * VKR_SUN_WG, 0, st>>>(scene_dev(c), sa, sf, n_chunks, pix, counter, width);
    return hipGetLastError();
}

// Everything a shadow launch asks of the runtime besides the launch itself -- occupancy of the persistent kernels, the chunk
// counters, the sun's direction table -- once per build, so that the per-frame calls only enqueue work.
hipError_t shadow_prepare(vkr_ctx* c)
{
    hipError_t e;
    if (c->sun_per_cu == 0) {
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->sun_per_cu, k_shadow_sun_pkt, 64 * VKR_SUN_WG, 0)) != hipSuccess) { c->sun_per_cu = 0; return e; }
        if (c->sun_per_cu < 1) c->sun_per_cu = 1;
    }
    if (!c->d_sun_frame.p && (e = c->d_sun_frame.alloc(sizeof(SunFrame))) != hipSuccess) return e;
    if (!c->d_sun_counter.p && (e = c->d_sun_counter.alloc(8)) != hipSuccess) return e;
    if (!c->d_shq_counter.p && (e = c->d_shq_counter.alloc(1)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t shadow_point_launch(vkr_ctx* c, const float* wp, const float* ng, const uint8_t* valid, uint32_t count,
                               const float P[3], float* out, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    if (c->trace_streaming && c->wide_count <= (1u << KQ_NODE_BITS)) {
        ShadowArgs sa{wp, ng, valid, count, V3{0, 0, 1}, 0.0f, 1.0f, V3{P[0], P[1], P[2]}, out, c->d_stats.p};
        return shadow_q_launch(c, 1, sa, st);
    }
    k_shadow_point<<<(count + VKR_TRACE_BLOCK - 1) / VKR_TRACE_BLOCK, VKR_TRACE_BLOCK, 0, st>>>(
        scene_dev(c), wp, ng, valid, count, V3{P[0], P[1], P[2]}, out);
    return hipGetLastError();
}

hipError_t gbuffer_launch(vkr_ctx* c, const FrameParams& fp, uint16_t* depth, uint16_t* normal, float* wp, float* ng,
                          uint8_t* valid, hipStream_t st)
{
    uint32_t tiles_x = ((uint32_t)fp.W + 15u) / 16u, tiles_y = ((uint32_t)fp.H + 15u) / 16u;
    if (c->gbuffer_packets) k_gbuffer<true><<<tiles_x * tiles_y, VKR_TRACE_BLOCK, 0, st>>>(scene_dev(c), fp, depth, normal, wp, ng, valid, tiles_x);
    else k_gbuffer<false><<<tiles_x * tiles_y, VKR_TRACE_BLOCK, 0, st>>>(scene_dev(c), fp, depth, normal, wp, ng, valid, tiles_x);
    return hipGetLastError();
}

hipError_t gbuffer_msaa_launch(vkr_ctx* c, const FrameParams& fp, int samples, uint16_t* depth, uint16_t* normal, float* wp,
                               float* ng, uint8_t* valid, hipStream_t st)
{
    const int set = samples == 8 ? 3 : (samples == 4 ? 2 : (samples == 2 ? 1 : 0));
    uint32_t tiles_x = ((uint32_t)fp.W + 15u) / 16u, tiles_y = ((uint32_t)fp.H + 15u) / 16u;
    // pixel-centre outputs (and, for one sample, everything) from the one-sample kernel
    if (c->gbuffer_packets)
        k_gbuffer<true><<<tiles_x * tiles_y, VKR_TRACE_BLOCK, 0, st>>>(scene_dev(c), fp, set == 0 ? depth : nullptr, set == 0 ? normal : nullptr,
                                                                       wp, ng, set == 0 ? valid : nullptr, tiles_x);
    else
        k_gbuffer<false><<<tiles_x * tiles_y, VKR_TRACE_BLOCK, 0, st>>>(scene_dev(c), fp, set == 0 ? depth : nullptr, set == 0 ? normal : nullptr,
                                                                        wp, ng, set == 0 ? valid : nullptr, tiles_x);
    if (set != 0) {
        if (c->gbuffer_packets) k_gbuffer_msaa<true><<<tiles_x * tiles_y, VKR_TRACE_BLOCK, 0, st>>>(scene_dev(c), fp, set, depth, normal, valid, tiles_x);
        else k_gbuffer_msaa<false><<<tiles_x * tiles_y, VKR_TRACE_BLOCK, 0, st>>>(scene_dev(c), fp, set, depth, normal, valid, tiles_x);
    }
    return hipGetLastError();
}

}  // namespace vkr
