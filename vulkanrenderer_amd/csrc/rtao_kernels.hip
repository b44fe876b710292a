// rtao_kernels.hip -- K1 trace+temporal, K2 A-Trous, K3 upsample (the three dispatches of RtaoComputePass,
// RtaoComputePass.cpp:503-571) as HIP kernels for gfx950.  Citations `:N` refer to the GLSL source each
// block restates: assets/shaders/CompShaders/rtao_trace_half.comp, rtao_atrous.comp, rtao_upsample.comp.
#include "internal.hpp"

namespace vkr {

// ---------------------------------------------------------------------------------------------
// samplers (R16F images as uint16 arrays)
// ---------------------------------------------------------------------------------------------
struct Img16 { const uint16_t* p; int w, h; };

// NEAREST, clamp-to-edge
VKR_DI float tex_nearest_r(const Img16& im, float u, float v)
{
    int x = iclamp((int)__builtin_floorf(u * (float)im.w), 0, im.w - 1);
    int y = iclamp((int)__builtin_floorf(v * (float)im.h), 0, im.h - 1);
    return half_to_float(im.p[(size_t)y * im.w + x]);
}
// LINEAR, clamp-to-edge, full-float weights
VKR_DI float tex_linear_r(const Img16& im, float u, float v)
{
    float x = u * (float)im.w - 0.5f, y = v * (float)im.h - 0.5f;
    float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    float ax = x - fx, ay = y - fy;
    int x0 = iclamp((int)fx, 0, im.w - 1), x1 = iclamp((int)fx + 1, 0, im.w - 1);
    int y0 = iclamp((int)fy, 0, im.h - 1), y1 = iclamp((int)fy + 1, 0, im.h - 1);
    float t00 = half_to_float(im.p[(size_t)y0 * im.w + x0]), t10 = half_to_float(im.p[(size_t)y0 * im.w + x1]);
    float t01 = half_to_float(im.p[(size_t)y1 * im.w + x0]), t11 = half_to_float(im.p[(size_t)y1 * im.w + x1]);
    float top = t00 + ax * (t10 - t00);
    float bot = t01 + ax * (t11 - t01);
    return top + ay * (bot - top);
}

// ---------------------------------------------------------------------------------------------
// rtao_trace_half.comp helpers
// ---------------------------------------------------------------------------------------------
VKR_DI float radicalInverseVdC(uint32_t bits)  // :69-77
{
    bits = __builtin_bitreverse32(bits);
    return (float)bits * 2.3283064365386963e-10f;
}
VKR_DI float hash11(float x) { return fractf(det_sin(x) * 43758.5453123f); }  // :64-67

VKR_DI void buildTbn(V3 n, V3* t, V3* b)  // :84-90
{
    V3 up = (__builtin_fabsf(n.y) < 0.999f) ? v3(0, 1, 0) : v3(1, 0, 0);
    *t = normalize3(cross3(up, n));
    *b = cross3(n, *t);
}

VKR_DI V3 reconstructViewPosFromLinearZ(const float* invProj, float u, float v, float linearViewZ)  // :92-100
{
    float fv[4];
    mul_m4_v4(invProj, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 1.0f, 1.0f, fv);
    float w = gmax(fv[3], 1e-6f);
    V3 dir = normalize3(V3{fv[0] / w, fv[1] / w, fv[2] / w});
    float t = linearViewZ / gmax(-dir.z, 1e-6f);
    return dir * t;
}

VKR_DI V3 offsetRay(V3 p, V3 n)  // :112-123
{
    const float ORIGIN_EPS = 1.0f / 32.0f, FLOAT_SCALE = 1.0f / 65536.0f, INT_SCALE = 256.0f;
    int ox = (int)(n.x * INT_SCALE), oy = (int)(n.y * INT_SCALE), oz = (int)(n.z * INT_SCALE);
    float px = u2f((uint32_t)((int32_t)f2u(p.x) + ((p.x < 0.0f) ? -ox : ox)));
    float py = u2f((uint32_t)((int32_t)f2u(p.y) + ((p.y < 0.0f) ? -oy : oy)));
    float pz = u2f((uint32_t)((int32_t)f2u(p.z) + ((p.z < 0.0f) ? -oz : oz)));
    return V3{(__builtin_fabsf(p.x) < ORIGIN_EPS) ? (p.x + FLOAT_SCALE * n.x) : px,
              (__builtin_fabsf(p.y) < ORIGIN_EPS) ? (p.y + FLOAT_SCALE * n.y) : py,
              (__builtin_fabsf(p.z) < ORIGIN_EPS) ? (p.z + FLOAT_SCALE * n.z) : pz};
}

// :184-213 over :125-182
template <bool STATS>
VKR_DI float traceAoRay(const SceneDev& sc, V3 origin, V3 dir, float tMin, float tMax, uint32_t* stack,
                        TraceCounters* cnt, uint32_t* nrays, int* occluded)
{
    *occluded = 0;
    HitDev h;
    if (STATS) (*nrays)++;
    if (!traverse<Q_AO, STATS>(sc, origin, dir, tMin, tMax, &h, stack, cnt)) return 1.0f;
    float t = h.t;
    if (t <= tMin * 1.5f) {
        float retryAdvance = gmax(tMin * 2.0f, 2e-4f);
        if (STATS) (*nrays)++;
        if (!traverse<Q_AO, STATS>(sc, origin + dir * retryAdvance, dir, gmax(tMin, 1e-4f), tMax, &h, stack, cnt))
            return 1.0f;
        t = h.t + retryAdvance;
    }
    if (t <= tMin * 1.5f) return 1.0f;
    *occluded = 1;
    float x = gclamp((t - tMin) / gmax(tMax - tMin, 1e-3f), 0.0f, 1.0f);
    return pow3(x);
}

VKR_DI bool computePrevUv(const float* prevViewProj, V3 worldPos, V2* prevUv)  // :215-228
{
    float c[4];
    mul_m4_v4(prevViewProj, worldPos.x, worldPos.y, worldPos.z, 1.0f, c);
    if (c[3] <= 1e-5f) { *prevUv = V2{0, 0}; return false; }
    float nx = c[0] / c[3], ny = c[1] / c[3];
    *prevUv = V2{nx * 0.5f + 0.5f, ny * 0.5f + 0.5f};
    if (prevUv->x < 0.0f || prevUv->y < 0.0f || prevUv->x > 1.0f || prevUv->y > 1.0f) return false;
    return true;
}

// XCD-aware block order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a contiguous
// run of tiles -> neighbouring tiles (which walk the same BVH subtrees) share one L2.  Bijective for any n.
VKR_DI uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    uint32_t q = n / 8u, r = n % 8u, x = b % 8u, s = b / 8u;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + s;
}

// ---------------------------------------------------------------------------------------------
// K1: trace + temporal (:230-472).  256 threads = 4 waves; each wave owns one 8x8 pixel tile (the
// reference's workgroup shape, :6), the block a 16x16 tile.
// ---------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(VKR_TRACE_BLOCK) void k_trace(SceneDev sc, FrameParams fp, const uint16_t* depthP,
                                                           const uint16_t* histPrevP, uint16_t* histCurr,
                                                           uint32_t* hitMask, int y0, int y1, uint32_t tiles_x,
                                                           uint32_t n_tiles, unsigned long long* stats)
{
    __shared__ uint32_t s_stack[VKR_LDS_STACK * VKR_TRACE_BLOCK];
    uint32_t* stack = s_stack + threadIdx.x;

    const uint32_t tile = xcd_remap(blockIdx.x, n_tiles);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = (int)(tile % tiles_x) * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = y0 + (int)(tile / tiles_x) * 16 + (wave >> 1) * 8 + (lane >> 3);
    const int W = fp.W, H = fp.H;
    if (px >= W || py >= y1) return;  // :233-235

    TraceCounters cnt{0, 0};
    uint32_t nrays = 0;
    const Img16 depth{depthP, W, H}, histPrev{histPrevP, W, H};
    const size_t o = (size_t)py * W + px;
    if (hitMask) hitMask[o] = 0;

    float u = gclamp(((float)px + 0.5f) / (float)W, 0.0f, 1.0f);  // :237-238
    float v = gclamp(((float)py + 0.5f) / (float)H, 0.0f, 1.0f);
    float linearViewZ = tex_nearest_r(depth, u, v);               // :239
    if (linearViewZ <= 1e-6f) { histCurr[o] = float_to_half(1.0f); return; }  // :240-243

    V3 viewPos = reconstructViewPosFromLinearZ(fp.invProj, u, v, linearViewZ);  // :247
    float viewZ = linearViewZ;
    float wp[4];
    mul_m4_v4(fp.invView, viewPos.x, viewPos.y, viewPos.z, 1.0f, wp);           // :249
    V3 worldPos = V3{wp[0], wp[1], wp[2]};
    const uint32_t debugMode = fp.debugMode;                                    // :250
    const int rayCount = fp.rayCount;                                           // :252
    const float baseRadius = fp.baseRadius, baseBias = fp.baseBias;             // :253-254

    float texelX = 1.0f / (float)W, texelY = 1.0f / (float)H;                   // :257
    float discontinuityAbsZ = gmax(0.02f, baseRadius * 0.25f);                  // :260
    const float DISCONTINUITY_REL_Z = 0.015f;

    float uXp = u + texelX, uXm = u - texelX, vYp = v + texelY, vYm = v - texelY;  // :263-266
    float depthXp = tex_nearest_r(depth, uXp, v);                               // :268-271
    float depthXm = tex_nearest_r(depth, uXm, v);
    float depthYp = tex_nearest_r(depth, u, vYp);
    float depthYm = tex_nearest_r(depth, u, vYm);
    depthXp = (depthXp <= 1e-6f) ? linearViewZ : depthXp;                       // :272-275
    depthXm = (depthXm <= 1e-6f) ? linearViewZ : depthXm;
    depthYp = (depthYp <= 1e-6f) ? linearViewZ : depthYp;
    depthYm = (depthYm <= 1e-6f) ? linearViewZ : depthYm;

    V3 viewPosXp = reconstructViewPosFromLinearZ(fp.invProj, uXp, v, depthXp);  // :277-280
    V3 viewPosXm = reconstructViewPosFromLinearZ(fp.invProj, uXm, v, depthXm);
    V3 viewPosYp = reconstructViewPosFromLinearZ(fp.invProj, u, vYp, depthYp);
    V3 viewPosYm = reconstructViewPosFromLinearZ(fp.invProj, u, vYm, depthYm);

    float zXp = -viewPosXp.z, zXm = -viewPosXm.z, zYp = -viewPosYp.z, zYm = -viewPosYm.z;  // :282-285
    float dXp = __builtin_fabsf(zXp - viewZ), dXm = __builtin_fabsf(zXm - viewZ);          // :287-290
    float dYp = __builtin_fabsf(zYp - viewZ), dYm = __builtin_fabsf(zYm - viewZ);
    float zden = gmax(viewZ, 1e-3f);
    float rXp = dXp / zden, rXm = dXm / zden, rYp = dYp / zden, rYm = dYm / zden;          // :292-295
    bool okXp = (rXp <= DISCONTINUITY_REL_Z) && (dXp <= discontinuityAbsZ);                // :297-300
    bool okXm = (rXm <= DISCONTINUITY_REL_Z) && (dXm <= discontinuityAbsZ);
    bool okYp = (rYp <= DISCONTINUITY_REL_Z) && (dYp <= discontinuityAbsZ);
    bool okYm = (rYm <= DISCONTINUITY_REL_Z) && (dYm <= discontinuityAbsZ);

    V3 viewPosX = viewPos, viewPosY = viewPos;                                  // :302-317
    if (okXp && okXm) viewPosX = (dXp <= dXm) ? viewPosXp : viewPosXm;
    else if (okXp) viewPosX = viewPosXp;
    else if (okXm) viewPosX = viewPosXm;
    if (okYp && okYm) viewPosY = (dYp <= dYm) ? viewPosYp : viewPosYm;
    else if (okYp) viewPosY = viewPosYp;
    else if (okYm) viewPosY = viewPosYm;

    V3 viewDx = viewPosX - viewPos, viewDy = viewPosY - viewPos;                // :319-321
    float footprint = gmax(length3(viewDx), length3(viewDy));

    V3 bestViewN = v3(0, 0, 0);                                                 // :326-351
    float bestScore = 1e30f;
    {
        const bool ok[4] = {okXp && okYp, okXp && okYm, okXm && okYp, okXm && okYm};
        const V3 ea[4] = {viewPosYp - viewPos, viewPosXp - viewPos, viewPosYp - viewPos, viewPosXm - viewPos};
        const V3 eb[4] = {viewPosXp - viewPos, viewPosYm - viewPos, viewPosXm - viewPos, viewPosYm - viewPos};
        const float sc4[4] = {dXp + dYp, dXp + dYm, dXm + dYp, dXm + dYm};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            V3 n = cross3(ea[k], eb[k]);
            float n2 = dot3(n, n);
            if (ok[k] && n2 > 1e-14f && sc4[k] < bestScore) { bestScore = sc4[k]; bestViewN = n * inversesqrt(n2); }
        }
    }
    V3 geomNormal = v3(0, 1, 0);                                                // :353-359
    if (bestScore < 1e29f) geomNormal = normalize3(mul_m3_v3(fp.invView, bestViewN));
    V3 camPos = V3{fp.camPos[0], fp.camPos[1], fp.camPos[2]};
    if (dot3(geomNormal, camPos - worldPos) < 0.0f) geomNormal = neg(geomNormal);  // :360-362

    float biasFromFootprint = footprint * 0.05f;                                // :365
    float biasMax = gmax(baseBias, gmin(baseRadius * 0.02f, 0.03f));            // :367
    float bias = gclamp(gmax(baseBias, biasFromFootprint), baseBias, biasMax);  // :368
    float radiusScale = gclamp(sqrt_(gmax(footprint, 1e-4f) / 0.03f), 1.0f, 1.6f);  // :371
    float radius = baseRadius * radiusScale;                                    // :372

    V3 origin = offsetRay(worldPos, geomNormal);                                // :375
    V3 tT, tB;
    buildTbn(geomNormal, &tT, &tB);                                             // :376
    float tMin = gmax(1e-4f, bias * 0.25f);                                     // :377

    if (debugMode == 10u) { histCurr[o] = float_to_half(gclamp(bias / 0.05f, 0.0f, 1.0f)); return; }       // :381-385
    if (debugMode == 11u) { histCurr[o] = float_to_half(gclamp(footprint / 0.05f, 0.0f, 1.0f)); return; }  // :386-390

    float frameTerm = (rayCount >= 4) ? (fp.frameIndex * 131.0f) : 0.0f;        // :395
    float seedBase = ((float)px * 1.0f + (float)py * 57.0f) + frameTerm;        // :396
    float angle = hash11(seedBase) * 6.28318530718f;                            // :397
    float cs, sn;
    det_sincos(angle, &sn, &cs);                                                // :398-399

    float visibility = 0.0f;
    uint32_t mask = 0;
    for (int i = 0; i < rayCount; ++i) {                                        // :403-432
        float xix = (float)i / gmax((float)rayCount, 1.0f);                     // hammersley :79-82
        float xiy = radicalInverseVdC((uint32_t)i);
        float ax = xix - 0.5f, ay = xiy - 0.5f;
        float rx = (cs * ax + sn * ay) + 0.5f;                                  // rot*(xi-0.5)+0.5, :400,405
        float ry = (-sn * ax + cs * ay) + 0.5f;
        rx = gclamp(rx, 0.0f, 1.0f);                                            // :406
        ry = gclamp(ry, 0.0f, 1.0f);
        float phi = 6.28318530718f * rx;                                        // :408
        float cosTheta = sqrt_(1.0f - ry), sinTheta = sqrt_(ry);                // :409-410
        float sphi, cphi;
        det_sincos(phi, &sphi, &cphi);
        V3 l = V3{cphi * sinTheta, sphi * sinTheta, cosTheta};                  // :411
        V3 rd = V3{(tT.x * l.x + tB.x * l.y) + geomNormal.x * l.z, (tT.y * l.x + tB.y * l.y) + geomNormal.y * l.z,
                   (tT.z * l.x + tB.z * l.y) + geomNormal.z * l.z};
        V3 rayDir = normalize3(rd);                                             // :412
        float nd = gclamp(dot3(geomNormal, rayDir), 0.0f, 1.0f);                // :418
        float grazing = 1.0f - nd;
        float slopeScaled = gmax(bias * 0.5f, footprint * 0.02f);               // :421
        float dirOffset = slopeScaled * (0.15f + (1.5f * grazing) * grazing);   // :422
        V3 rayOrigin = origin + rayDir * dirOffset;                             // :423
        float tMinRay = tMin;                                                   // :427-429
        tMinRay = gmax(tMinRay, dirOffset * 2.5f);
        tMinRay = gmax(tMinRay, ((radius * 0.03f) * grazing) * grazing);
        int occ = 0;
        visibility += traceAoRay<STATS>(sc, rayOrigin, rayDir, tMinRay, radius, stack, &cnt, &nrays, &occ);  // :431
        mask |= (uint32_t)occ << i;
    }
    if (hitMask) hitMask[o] = mask;
    if (STATS) {
        atomicAdd(&stats[0], (unsigned long long)nrays);
        atomicAdd(&stats[1], (unsigned long long)cnt.nodes);
        atomicAdd(&stats[2], (unsigned long long)cnt.tris);
    }
    float current = visibility / (float)rayCount;                               // :433
    if (debugMode == 12u) { histCurr[o] = float_to_half(gclamp(current, 0.0f, 1.0f)); return; }  // :435-438

    float hu = ((float)px + 0.5f) / (float)W, hv = ((float)py + 0.5f) / (float)H;  // :440
    V2 prevUv = V2{hu, hv};
    bool hasPrev = computePrevUv(fp.prevViewProj, worldPos, &prevUv);           // :442
    float su = hasPrev ? prevUv.x : hu, sv = hasPrev ? prevUv.y : hv;
    float history = tex_linear_r(histPrev, su, sv);                             // :443
    float prevViewZ = tex_nearest_r(depth, su, sv);                             // :444
    if (prevViewZ <= 1e-6f) prevViewZ = viewZ;                                  // :445-447
    float depthDiff = __builtin_fabsf(prevViewZ - viewZ) / gmax(viewZ, 1e-3f);  // :449
    float temporalAlpha = gclamp(fp.temporalAlphaRaw, 0.0f, 0.99f);             // :450
    float baseThreshold = gmax(fp.disocclusionRaw, 0.0f);                       // :452
    float noiseTerm = 0.6f / sqrt_((float)rayCount);                            // :453
    float disocclusionThreshold = gclamp(baseThreshold + noiseTerm, 0.0f, 1.5f);  // :454
    float variance = gmax(current * (1.0f - current), 0.0f) / (float)rayCount;  // :457
    float sigma = sqrt_(variance);                                              // :458
    float clampK = 2.5f;
    float historyClamped = gclamp(history, current - clampK * sigma, current + clampK * sigma);  // :460
    float delta = __builtin_fabsf(current - historyClamped);                    // :462
    float rejectDelta = smoothstep_(disocclusionThreshold, disocclusionThreshold * 2.0f + 1e-5f, delta);  // :463
    float rejectDepth = smoothstep_(0.02f, 0.15f, depthDiff);                   // :466
    float reject = gmax(rejectDelta, rejectDepth);                              // :467
    float historyWeight = (fp.frameIndex < 0.5f) ? 0.0f : (temporalAlpha * (1.0f - reject));  // :468
    float accum = mixf(current, historyClamped, historyWeight);                 // :469
    histCurr[o] = float_to_half(accum);                                         // :471
}

hipError_t rtao_trace_launch(vkr_ctx* c, const FrameParams& fp, const uint16_t* depth, const uint16_t* hist_prev,
                             uint16_t* hist_curr, uint32_t* hit_mask, int y0, int y1, hipStream_t st)
{
    if (y1 <= y0) return hipSuccess;
    uint32_t tiles_x = ((uint32_t)fp.W + 15u) / 16u, tiles_y = ((uint32_t)(y1 - y0) + 15u) / 16u;
    uint32_t n_tiles = tiles_x * tiles_y;
    SceneDev sc = scene_dev(c);
    if (c->stats_enabled)
        k_trace<true><<<n_tiles, VKR_TRACE_BLOCK, 0, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1,
                                                          tiles_x, n_tiles, c->d_stats.p);
    else
        k_trace<false><<<n_tiles, VKR_TRACE_BLOCK, 0, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1,
                                                           tiles_x, n_tiles, nullptr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// A-Trous operand: decodeNormalSafe (:38-46) + max(depth, 1e-3) (:70-71,92-93) evaluated ONCE per pixel per
// frame instead of 25 x N times; float4 {n.xyz, z} keeps the exact float values the taps would compute.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep(const uint16_t* depth, const uint16_t* normal, float4* nz, int W,
                                              int y0, int y1)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= y1) return;
    size_t o = (size_t)y * W + x;
    ushort4 e = ((const ushort4*)normal)[o];
    V3 n = V3{half_to_float(e.x) * 2.0f - 1.0f, half_to_float(e.y) * 2.0f - 1.0f, half_to_float(e.z) * 2.0f - 1.0f};
    float l2 = dot3(n, n);
    V3 r = (n.x != n.x || n.y != n.y || n.z != n.z || l2 < 1e-6f) ? v3(0, 1, 0) : n * inversesqrt(l2);
    float z = gmax(half_to_float(depth[o]), 1e-3f);
    nz[o] = make_float4(r.x, r.y, r.z, z);
}

hipError_t rtao_prep_launch(vkr_ctx* c, const uint16_t* depth, const uint16_t* normal, int y0, int y1, hipStream_t st)
{
    if (y1 <= y0) return hipSuccess;
    dim3 grid((c->W + 63u) / 64u, ((uint32_t)(y1 - y0) + 3u) / 4u);
    k_prep<<<grid, 256, 0, st>>>(depth, normal, c->d_nz.p, (int)c->W, y0, y1);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K2: one A-Trous iteration (rtao_atrous.comp:48-120).  Every sampler in the shader lands on an exact texel
// centre, so taps are texel fetches at clamp(pix + (x,y)*step) (clamp(uv,0,1) + clamp-to-edge, :81).
// dst2 (optional) receives the same value: the last iteration writes rtao_full directly (K3 folded in).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_atrous(const uint16_t* src, const float4* nz, uint16_t* dst, uint16_t* dst2,
                                                int W, int H, int y0, int y1, int step, float sigmaAbs)
{
    int px = blockIdx.x * 32 + (threadIdx.x & 31);
    int py = y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (px >= W || py >= y1) return;
    size_t o = (size_t)py * W + px;
    float center = half_to_float(src[o]);                                       // :62-69
    float4 cn = nz[o];                                                          // :70-72
    float centerZ = cn.w;
    V3 centerN = V3{cn.x, cn.y, cn.z};
    float sum = 0.0f, weightSum = 0.0f;
    const float sigmaRel = 0.0025f;                                             // :102
    float stepScale = 1.0f + 0.25f * (float)step;                               // :104
    float zden = gmax(centerZ, 1e-3f);
#pragma unroll
    for (int y = -2; y <= 2; ++y) {                                             // :78-112
        int ty = iclamp(py + y * step, 0, H - 1);
#pragma unroll
        for (int x = -2; x <= 2; ++x) {
            int tx = iclamp(px + x * step, 0, W - 1);
            size_t to = (size_t)ty * W + tx;
            float s = half_to_float(src[to]);                                   // :82-89
            float4 tn = nz[to];                                                 // :92-94
            float spatial = det_exp(-0.5f * (float)(x * x + y * y));            // :90
            float range = det_exp(-__builtin_fabsf(s - center) * 8.0f);         // :91
            float dz = __builtin_fabsf(tn.w - centerZ);                         // :98
            float dzRel = dz / zden;                                            // :99
            float depthNorm = gmax(dz / sigmaAbs, dzRel / sigmaRel);            // :103
            depthNorm *= stepScale;                                             // :104
            float depthWeight = det_exp(-depthNorm * depthNorm);                // :105
            float normalWeight = pow32(gclamp(dot3(centerN, V3{tn.x, tn.y, tn.z}), 0.0f, 1.0f));  // :107
            float w = ((spatial * range) * depthWeight) * normalWeight;         // :108
            sum += s * w;                                                       // :109
            weightSum += w;                                                     // :110
        }
    }
    float filtered = (weightSum > 0.0f) ? (sum / weightSum) : center;           // :114
    uint16_t r = float_to_half(filtered);
    dst[o] = r;                                                                 // :115-119
    if (dst2) dst2[o] = r;
}

hipError_t rtao_atrous_launch(vkr_ctx* c, float radius_raw, int step, const uint16_t* src, uint16_t* dst,
                              uint16_t* dst2, int y0, int y1, hipStream_t st)
{
    if (y1 <= y0) return hipSuccess;
    float radius = radius_raw < 0.01f ? 0.01f : radius_raw;                     // :100 max(rtaoParams0.z, 0.01)
    if (!(radius_raw == radius_raw)) radius = radius_raw;                       // GLSL max(NaN, y) = NaN
    float sigmaAbs = (0.01f < radius * 0.05f) ? radius * 0.05f : 0.01f;         // :101
    if (step < 1) step = 1;                                                     // :76
    dim3 grid((c->W + 31u) / 32u, ((uint32_t)(y1 - y0) + 7u) / 8u);
    k_atrous<<<grid, 256, 0, st>>>(src, c->d_nz.p, dst, dst2, (int)c->W, (int)c->H, y0, y1, step, sigmaAbs);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K3: upsample (rtao_upsample.comp:35-58): full-res copy, or debug view 9 = normal.y.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_upsample(const uint16_t* src, const uint16_t* normal, uint16_t* dst, int W,
                                                  int y0, int y1, uint32_t debugMode)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= y1) return;
    size_t o = (size_t)y * W + x;
    if (debugMode == 9u) {                                                      // :45-52
        ushort4 e = ((const ushort4*)normal)[o];
        V3 n = normalize3(V3{half_to_float(e.x) * 2.0f - 1.0f, half_to_float(e.y) * 2.0f - 1.0f,
                             half_to_float(e.z) * 2.0f - 1.0f});
        dst[o] = float_to_half(gclamp(n.y * 0.5f + 0.5f, 0.0f, 1.0f));
    } else {
        dst[o] = src[o];                                                        // :55-57
    }
}

hipError_t rtao_upsample_launch(vkr_ctx* c, uint32_t debug_mode, const uint16_t* src, const uint16_t* normal,
                                uint16_t* dst, int y0, int y1, hipStream_t st)
{
    if (y1 <= y0) return hipSuccess;
    dim3 grid((c->W + 63u) / 64u, ((uint32_t)(y1 - y0) + 3u) / 4u);
    k_upsample<<<grid, 256, 0, st>>>(src, normal, dst, (int)c->W, y0, y1, debug_mode);
    return hipGetLastError();
}

__global__ void k_fill16(uint16_t* img, size_t count, uint16_t v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) img[i] = v;
}

hipError_t fill_r16f_launch(uint16_t* img, size_t count, float value, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    uint16_t h = __builtin_bit_cast(uint16_t, (_Float16)value);
    size_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    k_fill16<<<(uint32_t)blocks, 256, 0, st>>>(img, count, h);
    return hipGetLastError();
}

}  // namespace vkr
