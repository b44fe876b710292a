// shadow_kernels.hip -- K4 sun-disc Poisson shadow rays, K5 point-light shadow ray (pbr.frag:169-260 lifted from
// per-fragment to per-pixel), and the primary-ray G-buffer producer that stands in for DepthPrepass.
#include "internal.hpp"

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

// Primary visibility: what DepthPrepass leaves in linearDepthResolve / normalResolve
// (depth_prepass.vert:42-52 viewZ = -viewPos.z; depth_only.frag:31-34 normal*0.5+0.5, linear Z), plus the
// worldPos / Ng varyings pbr.frag shades with (:339-346).
__global__ __launch_bounds__(VKR_TRACE_BLOCK) void k_gbuffer(SceneDev sc, FrameParams fp, uint16_t* depth,
                                                             uint16_t* normal, float* wpOut, float* ngOut,
                                                             uint8_t* validOut, uint32_t tiles_x)
{
    __shared__ uint32_t s_stack[2 * VKR_LDS_STACK * VKR_TRACE_BLOCK];
    uint32_t* stack = s_stack + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (int)(blockIdx.x % tiles_x) * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = (int)(blockIdx.x / tiles_x) * 16 + (wave >> 1) * 8 + (lane >> 3);
    const int W = fp.W, H = fp.H;
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    float u = ((float)x + 0.5f) / (float)W, v = ((float)y + 0.5f) / (float)H;
    float fv[4];
    mul_m4_v4(fp.invProj, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 1.0f, 1.0f, fv);
    V3 dv = normalize3(V3{fv[0] / fv[3], fv[1] / fv[3], fv[2] / fv[3]});
    V3 d = normalize3(mul_m3_v3(fp.invView, dv));
    V3 cam = V3{fp.camPos[0], fp.camPos[1], fp.camPos[2]};
    HitDev h;
    bool hit = traverse<Q_PRIMARY, false>(sc, cam, d, 0.0f, 1.0e30f, &h, stack, nullptr);
    float z = 0.0f;
    V3 n = v3(0, 0, 1), wp = v3(0, 0, 0), ng = v3(0, 1, 0);
    if (hit) {
        wp = cam + d * h.t;
        float vp[4];
        mul_m4_v4(fp.view, wp.x, wp.y, wp.z, 1.0f, vp);
        z = -vp[2];
        const float4* tp = sc.tris + (size_t)h.tri * 3;
        float4 b = tp[1], c = tp[2];
        n = normalize3(cross3(V3{b.x, b.y, b.z}, V3{c.x, c.y, c.z}));
        ng = (dot3(n, cam - wp) < 0.0f) ? neg(n) : n;
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

hipError_t shadow_sun_launch(vkr_ctx* c, const float* wp, const float* ng, const uint8_t* valid, uint32_t count,
                             const float L[3], float radius, float samples, float* out, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    k_shadow_sun<<<(count + VKR_TRACE_BLOCK - 1) / VKR_TRACE_BLOCK, VKR_TRACE_BLOCK, 0, st>>>(
        scene_dev(c), wp, ng, valid, count, V3{L[0], L[1], L[2]}, radius, samples, out);
    return hipGetLastError();
}

hipError_t shadow_point_launch(vkr_ctx* c, const float* wp, const float* ng, const uint8_t* valid, uint32_t count,
                               const float P[3], float* out, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    k_shadow_point<<<(count + VKR_TRACE_BLOCK - 1) / VKR_TRACE_BLOCK, VKR_TRACE_BLOCK, 0, st>>>(
        scene_dev(c), wp, ng, valid, count, V3{P[0], P[1], P[2]}, out);
    return hipGetLastError();
}

hipError_t gbuffer_launch(vkr_ctx* c, const FrameParams& fp, uint16_t* depth, uint16_t* normal, float* wp, float* ng,
                          uint8_t* valid, hipStream_t st)
{
    uint32_t tiles_x = ((uint32_t)fp.W + 15u) / 16u, tiles_y = ((uint32_t)fp.H + 15u) / 16u;
    k_gbuffer<<<tiles_x * tiles_y, VKR_TRACE_BLOCK, 0, st>>>(scene_dev(c), fp, depth, normal, wp, ng, valid, tiles_x);
    return hipGetLastError();
}

}  // namespace vkr
