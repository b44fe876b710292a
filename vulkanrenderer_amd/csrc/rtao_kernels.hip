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

// ---------------------------------------------------------------------------------------------
// K1: trace + temporal (:230-472).  One wave per workgroup, one 8x8 pixel tile per wave (the reference's
// workgroup shape, :6): a tile's cost varies a lot (sky vs foliage), and single-wave workgroups release their CU
// slot the moment their tile is done.  Tiles are dealt round-robin over the XCDs (plain blockIdx order) because
// L1/L2 hit rates are already ~97 % / 99.5 % and balance matters more than L2 affinity.  Three phases per wave:
//   1. per pixel (lane = pixel): everything of main() up to the ray loop, then the rays of :403-430 are
//      GENERATED into the wave's slice of a ray buffer (2 x float4 per ray, compacted over valid pixels);
//   2. per ray (lane = worker): the 64 lanes pull rays from that slice as they become free (the tile's
//      rays/px * 64 rays form the wave's work list), so a lane whose ray ends early -- or whose own pixel is sky --
//      does not idle until the slowest ray of the wave finishes; each lane runs a one-step-per-iteration traversal
//      state machine (node step or triangle step, voted per wave; LDS-only short stack) incl. the self-hit
//      retry of traceAoRay (:193-200) and stores visibility (sign bit = "a hit contributed occlusion") in the
//      slice's result area;
//   3. per pixel again: visibility summed in ray order (:431-433) and the temporal accumulation (:440-471).
// The ray slice is written and read by the same wave (workgroup-scope fences = s_waitcnt only: one CU, one L1).
// The short stack spills to per-lane scratch beyond VKR_LDS_STACK entries; the spill path is entered only by a
// lane about to run past the LDS part, so it stays out of the common instruction stream.
// ---------------------------------------------------------------------------------------------
constexpr int CUR_DONE = -1;   // not a leaf reference: it would name triangle 2^30 - 1 (the scene limit is 2^28)

#ifndef K1_WG
#define K1_WG 4            // waves per k_trace_q workgroup: 1, or 4 = the waves of a workgroup take neighbouring items at the same time (below)
#endif
constexpr int K1_BLOCK = 64;                 // k_trace (one ray per lane)
constexpr int K1Q_BLOCK = 64 * K1_WG;        // k_trace_q
#ifndef K1_UNIT_TILES
#define K1_UNIT_TILES 1    // K1_WG > 1: 8x8 tiles a workgroup fetches from the counters at a time
#endif
#ifdef VKR_EXP_RUNS
#define K1_EXP_RUNS_ON 1
#else
#define K1_EXP_RUNS_ON 0
#endif
constexpr uint32_t K1_RING = 16;             // K1_WG > 1: units a workgroup keeps published at a time

VKR_DI uint32_t ray_slice_float4s(int rayCount) { return (uint32_t)rayCount * 64u * 2u + (uint32_t)rayCount * 16u; }

struct PixelCtx {
    bool valid;            // this pixel casts rays
    V3 worldPos, geomNormal, origin, tT, tB;
    float viewZ, bias, footprint, radius, tMin, cs, sn;
};

// main() :237-400 for one pixel: early-outs store their value themselves and leave valid = false.
VKR_DI void setup_pixel(const FrameParams& fp, const Img16& depth, uint16_t* histCurr, uint32_t* hitMask, int px, int py,
                        PixelCtx& c)
{
    const int W = fp.W, H = fp.H;
    const size_t o = (size_t)py * W + px;
    c.valid = false;
    c.worldPos = v3(0, 0, 0); c.geomNormal = v3(0, 1, 0); c.origin = v3(0, 0, 0); c.tT = v3(0, 0, 0); c.tB = v3(0, 0, 0);
    c.viewZ = 0.0f; c.bias = 0.0f; c.footprint = 0.0f; c.radius = 0.0f; c.tMin = 0.0f; c.cs = 1.0f; c.sn = 0.0f;
    if (hitMask) hitMask[o] = 0;
    float u = gclamp(((float)px + 0.5f) / (float)W, 0.0f, 1.0f);                // :237-238
    float v = gclamp(((float)py + 0.5f) / (float)H, 0.0f, 1.0f);
    float linearViewZ = tex_nearest_r(depth, u, v);                             // :239
    if (linearViewZ <= 1e-6f) { histCurr[o] = float_to_half(1.0f); return; }    // :240-243

    V3 viewPos = reconstructViewPosFromLinearZ(fp.invProj, u, v, linearViewZ);  // :247
    const float viewZ = linearViewZ;
    c.viewZ = viewZ;
    float wp[4];
    mul_m4_v4(fp.invView, viewPos.x, viewPos.y, viewPos.z, 1.0f, wp);           // :249
    c.worldPos = V3{wp[0], wp[1], wp[2]};
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
    const float footprint = gmax(length3(viewDx), length3(viewDy));
    c.footprint = footprint;

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
    if (dot3(geomNormal, camPos - c.worldPos) < 0.0f) geomNormal = neg(geomNormal);  // :360-362
    c.geomNormal = geomNormal;

    float biasFromFootprint = footprint * 0.05f;                                // :365
    float biasMax = gmax(baseBias, gmin(baseRadius * 0.02f, 0.03f));            // :367
    c.bias = gclamp(gmax(baseBias, biasFromFootprint), baseBias, biasMax);      // :368
    float radiusScale = gclamp(sqrt_(gmax(footprint, 1e-4f) / 0.03f), 1.0f, 1.6f);  // :371
    c.radius = baseRadius * radiusScale;                                        // :372

    c.origin = offsetRay(c.worldPos, geomNormal);                               // :375
    buildTbn(geomNormal, &c.tT, &c.tB);                                         // :376
    c.tMin = gmax(1e-4f, c.bias * 0.25f);                                       // :377

    if (fp.debugMode == 10u) { histCurr[o] = float_to_half(gclamp(c.bias / 0.05f, 0.0f, 1.0f)); return; }       // :381-385
    if (fp.debugMode == 11u) { histCurr[o] = float_to_half(gclamp(footprint / 0.05f, 0.0f, 1.0f)); return; }    // :386-390

    float frameTerm = (fp.rayCount >= 4) ? (fp.frameIndex * 131.0f) : 0.0f;     // :395
    float seedBase = ((float)px * 1.0f + (float)py * 57.0f) + frameTerm;        // :396
    float angle = hash11(seedBase) * 6.28318530718f;                            // :397
    det_sincos(angle, &c.sn, &c.cs);                                            // :398-399
    c.valid = true;
}

// ray i of the pixel (:404-429)
VKR_DI void make_ray(const PixelCtx& c, int rayCount, int i, V3* rayOrigin, V3* rayDir, float* tMinRay)
{
    float xix = (float)i / gmax((float)rayCount, 1.0f);                         // hammersley :79-82
    float xiy = radicalInverseVdC((uint32_t)i);
    float ax = xix - 0.5f, ay = xiy - 0.5f;
    float rx = (c.cs * ax + c.sn * ay) + 0.5f;                                  // rot*(xi-0.5)+0.5, :400,405
    float ry = (-c.sn * ax + c.cs * ay) + 0.5f;
    rx = gclamp(rx, 0.0f, 1.0f);                                                // :406
    ry = gclamp(ry, 0.0f, 1.0f);
    float phi = 6.28318530718f * rx;                                            // :408
    float cosTheta = sqrt_(1.0f - ry), sinTheta = sqrt_(ry);                    // :409-410
    float sphi, cphi;
    det_sincos(phi, &sphi, &cphi);
    V3 l = V3{cphi * sinTheta, sphi * sinTheta, cosTheta};                      // :411
    const V3 tT = c.tT, tB = c.tB, n = c.geomNormal;
    V3 rd = V3{(tT.x * l.x + tB.x * l.y) + n.x * l.z, (tT.y * l.x + tB.y * l.y) + n.y * l.z,
               (tT.z * l.x + tB.z * l.y) + n.z * l.z};
    V3 d = normalize3(rd);                                                      // :412
    float nd = gclamp(dot3(n, d), 0.0f, 1.0f);                                  // :418
    float grazing = 1.0f - nd;
    float slopeScaled = gmax(c.bias * 0.5f, c.footprint * 0.02f);               // :421
    float dirOffset = slopeScaled * (0.15f + (1.5f * grazing) * grazing);       // :422
    *rayOrigin = c.origin + d * dirOffset;                                      // :423
    float t = c.tMin;                                                           // :427-429
    t = gmax(t, dirOffset * 2.5f);
    t = gmax(t, ((c.radius * 0.03f) * grazing) * grazing);
    *tMinRay = t;
    *rayDir = d;
}

// :433-471 for one pixel, given the summed visibility and the per-ray occlusion bits
VKR_DI void resolve_pixel(const FrameParams& fp, const Img16& depth, const Img16& histPrev, uint16_t* histCurr,
                          uint32_t* hitMask, int px, int py, const PixelCtx& c, float visibility, uint32_t mask)
{
    const int W = fp.W, H = fp.H, rayCount = fp.rayCount;
    const size_t o = (size_t)py * W + px;
    const float viewZ = c.viewZ;
    if (hitMask) hitMask[o] = mask;
    float current = visibility / (float)rayCount;                               // :433
    if (fp.debugMode == 12u) { histCurr[o] = float_to_half(gclamp(current, 0.0f, 1.0f)); return; }  // :435-438

    float hu = ((float)px + 0.5f) / (float)W, hv = ((float)py + 0.5f) / (float)H;  // :440
    V2 prevUv = V2{hu, hv};
    bool hasPrev = computePrevUv(fp.prevViewProj, c.worldPos, &prevUv);         // :442
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

template <bool STATS>
__global__ __launch_bounds__(K1_BLOCK) void k_trace(SceneDev sc, FrameParams fp, const uint16_t* depthP,
                                                    const uint16_t* histPrevP, uint16_t* histCurr, uint32_t* hitMask,
                                                    int y0, int y1, uint32_t tiles_x, uint32_t tile_base, float4* rayBuf,
                                                    unsigned long long* stats)
{
    __shared__ uint32_t s_stack[2 * VKR_LDS_STACK * K1_BLOCK];
    __shared__ float s_radius[K1_BLOCK];
    uint32_t* stack = s_stack + threadIdx.x;
    constexpr int B = K1_BLOCK;

    const uint32_t tile = tile_base + blockIdx.x;   // the ray slice is indexed by blockIdx.x alone: launches of a frame reuse it
    const int lane = threadIdx.x;
    const int px = (int)(tile % tiles_x) * 8 + (lane & 7);
    const int py = y0 + (int)(tile / tiles_x) * 8 * fp.stripeMul + (lane >> 3);
    const int W = fp.W, H = fp.H;
    const Img16 depth{depthP, W, H}, histPrev{histPrevP, W, H};
    const int rayCount = fp.rayCount;                                           // :252
    float4* slice = rayBuf + (size_t)blockIdx.x * ray_slice_float4s(rayCount);
    float* results = (float*)(slice + (size_t)rayCount * 128u);                 // [ray i][pixel lane]

    // ------------------------------------------------------------------ phase 1: per pixel
    PixelCtx pc;
    pc.valid = false;
    pc.radius = 0.0f;
    if (px < W && py < y1) setup_pixel(fp, depth, histCurr, hitMask, px, py, pc);   // :233-235
    const bool valid = pc.valid;
    s_radius[threadIdx.x] = pc.radius;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long vmask = __ballot(valid);
    const uint32_t nvalid = (uint32_t)__popcll(vmask);
    if (nvalid == 0) return;   // wave-uniform: nothing to trace, nothing to resolve
    if (valid) {
        const uint32_t rank = (uint32_t)__popcll(vmask & lt_mask);
        for (int i = 0; i < rayCount; ++i) {                                    // :403-430
            V3 rayOrigin, rayDir;
            float tMinRay;
            make_ray(pc, rayCount, i, &rayOrigin, &rayDir, &tMinRay);
            float4* rec = slice + ((size_t)i * nvalid + rank) * 2u;
            rec[0] = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, tMinRay);
            rec[1] = make_float4(rayDir.x, rayDir.y, rayDir.z, u2f((uint32_t)i * 64u + (uint32_t)lane));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ------------------------------------------------------------------ phase 2: per ray, dynamic pull
    {
        const uint32_t total = (sc.tri_count != 0u) ? nvalid * (uint32_t)rayCount : 0u;   // an empty scene has a root, but nothing walks it
        uint32_t next = 0;                 // wave-uniform
        bool active = false, found = false, retried = false;
        uint32_t spill[2 * (VKR_MAX_STACK - VKR_LDS_STACK)];   // scratch: touched only by rays deeper than the LDS part
        int cur = CUR_DONE;
        // Postponed leaves (after Aila & Laine's speculative traversal, depth 2): a lane that reaches a leaf parks it
        // and keeps walking nodes, so node steps and triangle steps both run fuller.  Closest-hit results do not
        // depend on the visiting order.
        int pend0 = CUR_DONE, pend1 = CUR_DONE;
        uint32_t sp = 0;
        V3 ro = v3(0, 0, 0), rdir = v3(0, 0, 1), id = v3(0, 0, 0);
        float tmin = 0.0f, tmin0 = 0.0f, tmax = 0.0f, tbest = 0.0f, adv = 0.0f;
        uint32_t rid = 0;
        TraceCounters cnt{0, 0, 0, 0};
        uint32_t nrays = 0;
        const char* node_base = (const char*)sc.nodes;
        const char* tri_base = (const char*)sc.tris;

        for (;;) {
            // ---- service finished rays and pull new ones, in batches: doing this the moment any single lane
            // finishes would run ~100 instructions at 1/64 utilisation per ray.
            const bool ray_done = active && cur == CUR_DONE && pend0 == CUR_DONE;
            const unsigned long long donem = __ballot(ray_done);
            const unsigned long long idlem = __ballot(!active);
            const int n_serv = __popcll(donem) + ((next < total) ? __popcll(idlem) : 0);
            if ((donem | idlem) == ~0ull || n_serv >= VKR_REFILL_BATCH) {
                // a ray whose traversal ended: self-hit retry (:193-200) or result (:187-189,205-212)
                if (ray_done) {
                    if (found && !retried && tbest <= tmin0 * 1.5f) {
                        adv = gmax(tmin0 * 2.0f, 2e-4f);
                        ro = ro + rdir * adv;
                        tmin = gmax(tmin0, 1e-4f);
                        tbest = tmax; found = false; retried = true; cur = 0; sp = 0;
                        if (STATS) nrays++;
                    } else {
                        float vis = 1.0f;
                        if (found) {
                            float t = retried ? (tbest + adv) : tbest;
                            if (!(t <= tmin0 * 1.5f)) {
                                float x = gclamp((t - tmin0) / gmax(tmax - tmin0, 1e-3f), 0.0f, 1.0f);
                                vis = u2f(f2u(pow3(x)) | 0x80000000u);   // sign bit: this ray ended occluded
                            }
                        }
                        results[rid] = vis;
                        active = false;
                    }
                }
                // idle lanes pull the next rays of the tile
                const unsigned long long idle = __ballot(!active);
                if (idle != 0ull && next < total) {
                    if (!active) {
                        uint32_t k = next + (uint32_t)__popcll(idle & lt_mask);
                        if (k < total) {
                            float4 a = slice[(size_t)k * 2u], b = slice[(size_t)k * 2u + 1u];
                            ro = V3{a.x, a.y, a.z}; tmin = a.w; tmin0 = a.w;
                            rdir = V3{b.x, b.y, b.z}; rid = f2u(b.w);
                            id = V3{safe_rcp(rdir.x), safe_rcp(rdir.y), safe_rcp(rdir.z)};
                            tmax = s_radius[rid & 63u];
                            tbest = tmax; found = false; retried = false; cur = 0; sp = 0; active = true;
                            if (STATS) nrays++;
                        }
                    }
                    next += (uint32_t)__popcll(idle);
                }
            }
            if (__ballot(active) == 0ull) break;

            // ---- one step per iteration, chosen per wave: lanes holding a node and lanes holding a leaf do not
            // wait for each other's loops to drain; the cheaper triangle step runs as soon as 1/VKR_LEAF_VOTE as
            // many lanes want it as want a node step.
            const int n_node = __popcll(__ballot(cur >= 0));
            const int n_leaf = __popcll(__ballot(pend0 != CUR_DONE));
            const float tb = tbest * 1.000001f;   // padded: culling may only err towards extra visits
            bool need_pop = false;
            if (n_leaf == 0 || n_leaf * VKR_LEAF_VOTE < n_node * VKR_LEAF_VOTE_DEN) {
                if (cur >= 0) {
                    const float4* np = (const float4*)(node_base + ((size_t)(uint32_t)cur << 6));
                    const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                    if (STATS) { cnt.nodes++; if ((int)(__ffsll((long long)__ballot(1)) - 1) == lane) cnt.wave_node_iters++; }
                    // Slab tests straight on the quantised planes: plane = fma(q, s, o), t = (plane - ro) * id
                    // is evaluated as fma(q, s*id, (o - ro)*id): one fma per plane.  The direction sign picks the
                    // near/far byte-quads once per axis, so no per-plane min/max; unused slots are stored inverted
                    // (qmin = 255, qmax = 0) and miss by themselves.  Rounding of this form is covered by the leaf
                    // padding (bvh_build.hip tri_box) and by the outward quantisation.
                    const float ax = q2.z * id.x, ay = q2.w * id.y, az = q0.w * id.z;   // scales are stored as floats
                    const float bx = (q0.x - ro.x) * id.x, by = (q0.y - ro.y) * id.y, bz = (q0.z - ro.z) * id.z;
                    const uint32_t nx4 = id.x >= 0.0f ? f2u(q1.x) : f2u(q1.w), fx4 = id.x >= 0.0f ? f2u(q1.w) : f2u(q1.x);
                    const uint32_t ny4 = id.y >= 0.0f ? f2u(q1.y) : f2u(q2.x), fy4 = id.y >= 0.0f ? f2u(q2.x) : f2u(q1.y);
                    const uint32_t nz4 = id.z >= 0.0f ? f2u(q1.z) : f2u(q2.y), fz4 = id.z >= 0.0f ? f2u(q2.y) : f2u(q1.z);
                    uint32_t k0, k1, k2, k3, r0 = f2u(q3.x), r1 = f2u(q3.y), r2 = f2u(q3.z), r3 = f2u(q3.w);
#define VKR_SLAB(K, C)                                                                                              \
    {                                                                                                               \
        float tn = __builtin_fmaxf(__builtin_fmaxf(fma_(ubyte_f(nx4, C), ax, bx), fma_(ubyte_f(ny4, C), ay, by)),   \
                                   __builtin_fmaxf(fma_(ubyte_f(nz4, C), az, bz), tmin));                          \
        float tf = __builtin_fminf(__builtin_fminf(fma_(ubyte_f(fx4, C), ax, bx), fma_(ubyte_f(fy4, C), ay, by)),   \
                                   __builtin_fminf(fma_(ubyte_f(fz4, C), az, bz), tb));                            \
        K = (tn <= tf) ? f2u(tn) : 0xffffffffu;                                                                    \
    }
                    VKR_SLAB(k0, 0) VKR_SLAB(k1, 1) VKR_SLAB(k2, 2) VKR_SLAB(k3, 3)
#undef VKR_SLAB
                    // 5-exchange network on (tnear bits, child) pairs; tnear >= 0 orders as an unsigned integer
#define VKR_CSWAP(KA, RA, KB, RB)                                                                                  \
    {                                                                                                               \
        bool sw = KB < KA;                                                                                          \
        uint32_t tk = sw ? KB : KA, tr = sw ? RB : RA;                                                              \
        KB = sw ? KA : KB; RB = sw ? RA : RB; KA = tk; RA = tr;                                                     \
    }
                    VKR_CSWAP(k0, r0, k1, r1) VKR_CSWAP(k2, r2, k3, r3) VKR_CSWAP(k0, r0, k2, r2)
                    VKR_CSWAP(k1, r1, k3, r3) VKR_CSWAP(k1, r1, k2, r2)
#undef VKR_CSWAP
                    // deferred children, far first.  Common case: all three slots are still inside the LDS part of
                    // the stack; only a lane that is about to run past it takes the scratch-aware path.
                    if (sp + 3u <= (uint32_t)VKR_LDS_STACK) {
                        if (k3 != 0xffffffffu) { stack[(2 * sp) * B] = k3; stack[(2 * sp + 1) * B] = r3; ++sp; }
                        if (k2 != 0xffffffffu) { stack[(2 * sp) * B] = k2; stack[(2 * sp + 1) * B] = r2; ++sp; }
                        if (k1 != 0xffffffffu) { stack[(2 * sp) * B] = k1; stack[(2 * sp + 1) * B] = r1; ++sp; }
                    } else {
#define VKR_PUSH(K, R)                                                                                             \
    if (K != 0xffffffffu) {                                                                                         \
        if (sp < (uint32_t)VKR_LDS_STACK) { stack[(2 * sp) * B] = K; stack[(2 * sp + 1) * B] = R; }                 \
        else { spill[2 * (sp - VKR_LDS_STACK)] = K; spill[2 * (sp - VKR_LDS_STACK) + 1] = R; }                      \
        ++sp;                                                                                                       \
    }
                        VKR_PUSH(k3, r3) VKR_PUSH(k2, r2) VKR_PUSH(k1, r1)
#undef VKR_PUSH
                    }
                    if (k0 != 0xffffffffu) cur = (int)r0; else need_pop = true;
                }
            } else if (pend0 != CUR_DONE) {
                // ---- leaf: ONE triangle per step (queryAoClosestHit :125-182, closest committed hit)
                uint32_t first = leaf_ref_first((uint32_t)pend0), rest = leaf_ref_count((uint32_t)pend0) - 1u;   // rest = triangles after this one
                {
                    const float4* tp = (const float4*)(tri_base + (size_t)first * 48u);
                    float4 a = tp[0], b = tp[1], c = tp[2];
                    if (STATS) { cnt.tris++; if ((int)(__ffsll((long long)__ballot(1)) - 1) == lane) cnt.wave_tri_iters++; }
                    V3 v0 = V3{a.x, a.y, a.z}, e1 = V3{b.x, b.y, b.z}, e2 = V3{c.x, c.y, c.z};
                    V3 pvec = cross3(rdir, e2);
                    float det = dot3(e1, pvec);
                    float inv = 1.0f / det;
                    V3 tvec = ro - v0;
                    float u = dot3(tvec, pvec) * inv;
                    V3 qvec = cross3(tvec, e1);
                    float v = dot3(rdir, qvec) * inv;
                    float t = dot3(e2, qvec) * inv;
                    const uint32_t mw = f2u(a.w);
                    const uint32_t cls = mw >> 28;                              // candidate filter :142-173
                    bool hit = (det != 0.0f) && (u >= 0.0f && u <= 1.0f) && (v >= 0.0f && u + v <= 1.0f) &&
                               (t > tmin && t < tbest) && (cls != 1u);
                    if (hit && cls == 2u) hit = alpha_test(sc, mw & 0x0fffffffu, f2u(b.w), u, v);
                    if (hit) { tbest = t; found = true; }
                }
                if (rest != 0u) pend0 = (int)leaf_ref_make(first + 1u, rest);
                else { pend0 = pend1; pend1 = CUR_DONE; }
            }
            // ---- a leaf in hand moves to a free parking slot; the lane then continues with the next stack entry
            if (cur < 0 && cur != CUR_DONE && pend1 == CUR_DONE) {
                if (pend0 == CUR_DONE) pend0 = cur; else pend1 = cur;
                need_pop = true;
            }
            // ---- next entry still in front of the closest hit (entries carry their tnear)
            if (need_pop) {
                cur = CUR_DONE;
                const float tb2 = tbest * 1.000001f;
                while (sp > 0) {
                    --sp;
                    uint32_t key, ref;
                    if (sp < (uint32_t)VKR_LDS_STACK) { key = stack[(2 * sp) * B]; ref = stack[(2 * sp + 1) * B]; }
                    else { key = spill[2 * (sp - VKR_LDS_STACK)]; ref = spill[2 * (sp - VKR_LDS_STACK) + 1]; }
                    if (u2f(key) <= tb2) { cur = (int)ref; break; }
                }
            }
        }
        if (STATS) {
            atomicAdd(&stats[0], (unsigned long long)nrays);
            atomicAdd(&stats[1], (unsigned long long)cnt.nodes);
            atomicAdd(&stats[2], (unsigned long long)cnt.tris);
            atomicAdd(&stats[3], (unsigned long long)cnt.wave_node_iters);
            atomicAdd(&stats[4], (unsigned long long)cnt.wave_tri_iters);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ------------------------------------------------------------------ phase 3: per pixel resolve + temporal
    if (!valid) return;
    float visibility = 0.0f;
    uint32_t mask = 0;
    for (int i = 0; i < rayCount; ++i) {                                        // :431 in ray order
        uint32_t r = f2u(results[i * 64 + lane]);
        visibility += u2f(r & 0x7fffffffu);
        mask |= (r >> 31) << i;
    }
    resolve_pixel(fp, depth, histPrev, histCurr, hitMask, px, py, pc, visibility, mask);
}


// ---------------------------------------------------------------------------------------------
// K1, streaming form (the default).  The waves of persistent workgroups pull work items -- a block of P pixels with
// all their rays, at most 2^RB rays -- from counters; inside an item the rays do not own lanes: work lives in wave-wide
// LIFO queues in LDS,
//     node items  ray << (32-RB) | wide node          leaf items  ray << (32-RB) | first triangle << 1 | count-1
// and every step pops items of ONE kind, so node steps and leaf steps both run with (nearly) all lanes busy.  A node
// step tests the four child boxes against [tmin, closest hit so far] and appends the hit children to the two queues
// (ballot + mbcnt compaction; a leaf reference becomes a leaf item with one add); a leaf step tests the leaf's
// triangles and lowers the ray's closest hit with an LDS atomic min on the float bits; MASK candidates queue up
// once more and are alpha-tested a wave at a time.  No ordering, no per-lane stack: the closest committed hit does
// not depend on the order triangles are met in.  The self-hit retry (:193-200) re-traces a ray on the spot (K1_EARLY_RETRY).
//
// Shipped shape:
//   * 6 waves per SIMD (80 VGPRs; what spills sits in the rare refill / alpha paths, not in the node or leaf step), one item
//     per lane per step (K1_STEP 64), items of 2^7 = 128 rays (K1_RB 7: 16 pixels x 8 rays);
//   * the ray records the node step needs ({origin | tMin}, {1/direction | closest hit}) live in LDS (32 B per ray); only the
//     direction (leaf steps) stays in the wave's global slice;
//   * LDS queues of 288 node + 128 leaf + 128 alpha entries; overflow goes to a small global stack per wave, and the step
//     width is throttled only when LDS + global capacity could run out (strict depth-first bound);
//   * the per-pixel head and tail of main() run as kernels of their own (K1_SPLIT: k_pix_setup, k_pix_resolve);
//   * rays enter the tree at the cut of their 8x8 tile (K1_CUT: tile_cut_wave, run by k_pix_setup's waves), not at the root;
//   * workgroups of four waves (K1_WG), six per CU: the four waves take the four items of one tile at the same time and share
//     the CU's L1; the items are numbered tile by tile, eight counters hand them out (one per vertical slab = XCD).
// ---------------------------------------------------------------------------------------------
#ifndef K1_WAVES
#define K1_WAVES 6         // resident waves per SIMD the kernel is compiled for (VGPR budget 512 / K1_WAVES: 80)
#endif
#ifndef K1_RB
#define K1_RB 7            // ray bits of an item at most: 2^K1_RB rays (32 B of LDS each)
#endif
#ifndef K1_STEP
#define K1_STEP 64u        // node items per step: 64 (one per lane) or 128 (two per lane)
#endif
#ifndef K1_NQ
#define K1_NQ 288          // node queue entries in LDS (288 + 128 + 128 entries + 32 B x 128 rays = 6.1 KB: 24 workgroups per CU)
#endif
#ifndef K1_DIR_LDS
#define K1_DIR_LDS 0       // 1: ray directions in LDS (12 B per ray) instead of the wave's global slice -- measured slower (3.06 vs 2.90 ms)
#endif
#ifndef K1_SPLIT
#define K1_SPLIT 1         // 1: per-pixel setup and temporal resolve run as their own full-occupancy kernels around k_trace_q
#endif
#ifndef K1_SELECT_STORES
#define K1_SELECT_STORES 1  // node-step appends as unpredicated stores with a selected index (see the fast path of the node step)
#endif
#ifndef K1_EARLY_RETRY
#define K1_EARLY_RETRY 1   // 1: a self-hit re-traces its ray on the spot (same queue round) instead of in a second round
#endif
#ifndef K1_INJECT
#define K1_INJECT 64u      // rays injected at the root at a time (when fewer than K1_STEP node items wait)
#endif
#ifndef K1_LEAF_AT
#define K1_LEAF_AT 64u     // a leaf step runs as soon as this many leaf items wait (a full wave)
#endif
#ifndef K1_SLABS
#define K1_SLABS 8         // work counters: 1 = one row-major list, 8 = one vertical slab of tile columns per XCD
#endif
#ifndef K1_LQ
#define K1_LQ 128          // leaf queue entries in LDS (>= 64; a node step's leaf pushes beyond it go to the global stack)
#endif
#ifndef K1_SLOT0_FIRST
#define K1_SLOT0_FIRST 1  // node-step appends in reverse slot order, see the fast path of the node step
#endif
#ifndef K1_CUT
#define K1_CUT 1           // 1: rays enter the tree at a per-tile cut (tile_cut_wave) instead of at the root, see below
#endif
#ifndef K1_CUT_MAX
#define K1_CUT_MAX 16      // entries of a tile's cut at most (16 B each in LDS, after a 16-byte header)
#endif
static_assert(K1_CUT_MAX >= 1 && K1_CUT_MAX <= 16, "tile_cut_wave expands every open node of a round on 4 lanes each");
constexpr uint32_t K1_GNQ = 4096;      // node-item overflow stack per wave (global)
constexpr uint32_t K1_GLQ = 512;       // leaf-item overflow per wave (global): one node step pushes <= 4 * 128
constexpr uint32_t K1_AQ = 128;        // alpha-test queue (LDS): drained whenever it holds a full wave of candidates (draining at one
                                       // place only -- loop top or end of the leaf step -- costs 4-7 %: measured twice)
constexpr uint32_t K1_TRASH = K1_NQ + K1_LQ + K1_AQ + 8u;   // index (from the node queue) of an LDS word nobody reads: retry bits come first

struct QSliceLayout { uint32_t dir_f4, retry_f4, gnq_f4, glq_f4, total_f4; };   // per-wave global scratch, in float4 units
__host__ __device__ inline QSliceLayout q_slice_layout(int rb)
{
    QSliceLayout l;
    const uint32_t rays = 1u << rb;
    l.dir_f4 = rays;                  // {direction | tMin as generated} per ray
    l.retry_f4 = rays / 4u;           // one u32 per ray
    l.gnq_f4 = K1_GNQ / 4u;
    l.glq_f4 = K1_GLQ / 4u;
    l.total_f4 = l.dir_f4 + l.retry_f4 + l.gnq_f4 + l.glq_f4;
    return l;
}
__host__ __device__ inline size_t q_lds_bytes(int rb)
{
    return (size_t)(1u << rb) * (K1_DIR_LDS ? 44u : 32u) + (size_t)K1_NQ * 4u + (size_t)K1_LQ * 4u + (size_t)K1_AQ * 4u + 32u + 16u   // + retry bits + the trash word
           + (K1_CUT ? (size_t)(K1_CUT_MAX + 1 + 3) * 16u : 0u);                                                                              // + the tile's cut
}
// pixels per work item: the largest power of two <= 64 with P * rayCount <= 2^RB
__host__ __device__ inline int q_item_pixels(int rb, int rayCount)
{
    int p = 64;
    while (p > 1 && p * rayCount > (1 << rb)) p >>= 1;
    return p;
}

// -DVKR_DEBUG_BOUNDS: every LDS queue / ray / global-stack index of k_trace_q is checked on the device; a violation sets a bit
// in stats[23] (read back by vkr_trace_stats_get under VKR_PHASE_DUMP) and the index is forced to 0, so the run stays in
// bounds and reports instead of faulting.  (Round 1 recorded an undiagnosed GPU fault of a 7-waves-per-SIMD build.)
#ifdef VKR_DEBUG_BOUNDS
#define VKR_BOUND(IDX, LIMIT, CODE) (((uint32_t)(IDX) < (uint32_t)(LIMIT)) ? (uint32_t)(IDX) : (atomicOr((unsigned int*)&stats[23], (unsigned int)(CODE)), 0u))
#else
#define VKR_BOUND(IDX, LIMIT, CODE) (IDX)
#endif

// ---------------------------------------------------------------------------------------------
// K1_SPLIT: the per-pixel parts of the pass as streaming kernels of their own.  Inside k_trace_q an item is 16 pixels on 64
// lanes, so main():237-400 and :433-471 ran at a quarter of the lanes and held a persistent wave for 11 % of its time
// (VKR_PHASE_DUMP); here they run one pixel per lane at full occupancy.  Record per pixel (3 float4):
//   [0] worldPos.xyz | viewZ if the pixel casts rays, else 0      [1] geomNormal.xyz | radius      [2] bias, footprint, cos, sin
// Rows: the launch's stripes, row j of the launch = image row y0 + (j / 8) * 8 * stripeMul + j % 8 (k_trace_q's tiles).
// ---------------------------------------------------------------------------------------------
// K1_CUT: the block is 64 pixels x the 8 rows of one stripe = eight 8x8 tiles, and it also leaves the bound of every tile's ray
// segments for k_tile_cut (2 float4 per tile: minimum corner | valid flag, maximum corner): per pixel the half ball of radius
// `radius + largest origin offset` around the offset origin (:375,418-430), reduced over the tile.
VKR_DI uint32_t pack_h2(float a, float b)
{
    return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)b) << 16);
}
VKR_DI float h_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
VKR_DI float h_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }

// ---------------------------------------------------------------------------------------------
// K1_CUT: the entry cut of an 8x8 tile.  AO rays are short (tMax = radius <= 1.92 m, :371-372,430) and the pixels of a tile
// are centimetres apart, so all rays of a tile live in one small box -- and every one of them would walk the same upper
// levels of the tree to get there (about a third of its node visits).  tile_cut_wave walks them ONCE per tile: one wave per
// tile bounds the tile's ray segments (per pixel: the hemisphere of radius `radius + largest origin offset` around the
// offset origin, :375,418-430), then descends from the root through every child whose box overlaps that bound, level by
// level (up to 16 open nodes x 4 children on the 64 lanes), until the nodes are no larger than the bound (`small_factor`),
// are leaves, or the cut would exceed `cut_max` entries (then the nodes still open join the cut as they are).  A cut entry is
// 16 bytes: the child's box clipped to the bound, as six fp16 offsets from the bound's minimum corner (widened outward by
// more than the fp16 rounding), and the child reference.  k_trace_q tests every ray of an item against the tile's entries
// (uniform LDS reads, two rays per lane) and queues it at the entries it overlaps, instead of queueing it at the root.
// Box tests are conservative, so the closest committed hit of every ray -- and with it every output bit -- is unchanged.
// Record per tile: [0] bound minimum xyz | number of entries, [1 ..] entries.
// ---------------------------------------------------------------------------------------------
// One wave: the entry cut of the tile whose ray segments lie in [amin, amax] (amin.x > amax.x: the tile casts no rays).  `s_open` is 16
// uint4 of LDS owned by this wave.
VKR_DI void tile_cut_wave(const SceneDev& sc, uint32_t tile, const float* amin, const float* amax, uint4* cuts, uint32_t cut_max, float small_factor,
                          uint32_t l2_begin, uint32_t l2_count, uint4* s_open, int lane)
{
    uint4* out = cuts + (size_t)tile * (K1_CUT_MAX + 1);
    uint32_t ncut = 0;
    if (amin[0] <= amax[0] && sc.tri_count != 0u) {
        const float ex = amax[0] - amin[0], ey = amax[1] - amin[1], ez = amax[2] - amin[2];
        const float area_lim = small_factor * ((ex * ey + ey * ez) + ez * ex);
        const float wx = ex * 0.002f + 1.0e-6f, wy = ey * 0.002f + 1.0e-6f, wz = ez * 0.002f + 1.0e-6f;   // > fp16 rounding of a value <= extent
        if (lane == 0) s_open[0] = make_uint4(pack_h2(-wx, -wy), pack_h2(-wz, ex + wx), pack_h2(ey + wy, ez + wz), 0u);   // the root: everything
        uint32_t nopen = 1;
        // The first round opens the third level of the tree directly (its nodes are numbered l2_begin .. l2_begin + l2_count - 1 by the
        // level-order numbering of k_wide_level; every tile would get there through the same two rounds): three dependent loads
        // less per tile.  If that alone breaks the budget, the descent starts over at the root.
        bool shortcut = l2_count >= 1u && l2_count <= 16u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        while (nopen != 0u) {
            const uint32_t slot = (uint32_t)lane >> 2, k = (uint32_t)lane & 3u;
            const bool act = slot < (shortcut ? l2_count : nopen);
            const uint32_t node = act ? (shortcut ? l2_begin + slot : s_open[slot].w) : 0u;
            const float4* np = sc.nodes + (size_t)node * 4u;
            const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
            const uint32_t sh8 = 8u * k;
            const uint32_t ref = k == 0u ? f2u(q3.x) : (k == 1u ? f2u(q3.y) : (k == 2u ? f2u(q3.z) : f2u(q3.w)));
            // the child's box exactly as decode_node() sees it
            const float mnx = fma_((float)((f2u(q1.x) >> sh8) & 0xffu), q2.z, q0.x), mxx = fma_((float)((f2u(q1.w) >> sh8) & 0xffu), q2.z, q0.x);
            const float mny = fma_((float)((f2u(q1.y) >> sh8) & 0xffu), q2.w, q0.y), mxy = fma_((float)((f2u(q2.x) >> sh8) & 0xffu), q2.w, q0.y);
            const float mnz = fma_((float)((f2u(q1.z) >> sh8) & 0xffu), q0.w, q0.z), mxz = fma_((float)((f2u(q2.y) >> sh8) & 0xffu), q0.w, q0.z);
            const bool ov = act && ref != VKR_CHILD_NONE && mnx <= amax[0] && mxx >= amin[0] && mny <= amax[1] && mxy >= amin[1] &&
                            mnz <= amax[2] && mxz >= amin[2];
            const float dx = mxx - mnx, dy = mxy - mny, dz = mxz - mnz;
            const bool closed = ov && ((int)ref < 0 || ((dx * dy + dy * dz) + dz * dx) <= area_lim);
            const bool opened = ov && !closed;
            const unsigned long long cm = __ballot(closed), om = __ballot(opened);
            const uint32_t nc = (uint32_t)__popcll(cm), no = (uint32_t)__popcll(om);
            if (ncut + nc + no > cut_max) {
                if (shortcut) { shortcut = false; continue; }         // from the root after all (the root entry is in s_open[0])
                break;                                                // the nodes still open join the cut as they are
            }
            shortcut = false;
            // clipped to the bound, relative to its minimum corner, widened
            const uint4 e = make_uint4(pack_h2(gmax(mnx, amin[0]) - amin[0] - wx, gmax(mny, amin[1]) - amin[1] - wy),
                                       pack_h2(gmax(mnz, amin[2]) - amin[2] - wz, gmin(mxx, amax[0]) - amin[0] + wx),
                                       pack_h2(gmin(mxy, amax[1]) - amin[1] + wy, gmin(mxz, amax[2]) - amin[2] + wz), ref);
            if (closed) out[1u + ncut + mbcnt64(cm)] = e;
            ncut += nc;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // every lane has read its open entry
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (opened) s_open[mbcnt64(om)] = e;
            nopen = no;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if ((uint32_t)lane < nopen) out[1u + ncut + (uint32_t)lane] = s_open[lane];
        ncut += nopen;
    }
    if (lane == 0) out[0] = make_uint4(f2u(amin[0]), f2u(amin[1]), f2u(amin[2]), ncut);
}

__global__ __launch_bounds__(64) void k_tile_cut(SceneDev sc, uint32_t n_tiles, const float4* tile_bounds, uint4* cuts, uint32_t cut_max,
                                                 float small_factor, uint32_t l2_begin, uint32_t l2_count)
{
    __shared__ uint4 s_open[16];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const float4 b0 = tile_bounds[2u * (size_t)tile], b1 = tile_bounds[2u * (size_t)tile + 1u];   // k_pix_setup: bound of the tile's ray segments
    const float amin[3] = {b0.x, b0.y, b0.z}, amax[3] = {b1.x, b1.y, b1.z};
    tile_cut_wave(sc, tile, amin, amax, cuts, cut_max, small_factor, l2_begin, l2_count, s_open, (int)threadIdx.x);
}

VKR_DI float group8_min(float v)
{
    v = gmin(v, __shfl_xor(v, 1, 64)); v = gmin(v, __shfl_xor(v, 2, 64)); return gmin(v, __shfl_xor(v, 4, 64));
}
VKR_DI float group8_max(float v)
{
    v = gmax(v, __shfl_xor(v, 1, 64)); v = gmax(v, __shfl_xor(v, 2, 64)); return gmax(v, __shfl_xor(v, 4, 64));
}
constexpr int PIX_ROWS = K1_CUT ? 8 : 4;       // rows of a k_pix_setup / k_pix_resolve block

__global__ __launch_bounds__(64 * PIX_ROWS) void k_pix_setup(FrameParams fp, const uint16_t* depthP, uint16_t* histCurr, uint32_t* hitMask,
                                                   int y0, int y1, int rows, float4* pixctx, float4* tile_bounds, uint32_t tiles_x,
                                                   uint32_t* item_counters, uint32_t n_counters,
                                                   SceneDev sc, uint4* cuts, uint32_t cut_max, float cut_small, uint32_t l2_begin, uint32_t l2_count)
{
#if K1_CUT
    __shared__ float s_b[8][8][6];            // [tile of the block][row][lo xyz, hi xyz]
    __shared__ float s_tb[8][8];              // [tile of the block][lo xyz, hi xyz]: the reduced bounds, for the fused cut
    __shared__ uint4 s_open[8][16];           // tile_cut_wave's open list, one per wave
#endif
    // k_trace_q's work counters start every launch at zero: cleared here (the kernel before it in the stream) instead of by a
    // fill kernel of their own
    if (blockIdx.x == 0u && blockIdx.y == 0u && threadIdx.x < n_counters) item_counters[threadIdx.x] = 0u;
    const int px = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63u);
    const int j = (int)blockIdx.y * PIX_ROWS + (int)(threadIdx.x >> 6);
    const int py = y0 + (j >> 3) * 8 * fp.stripeMul + (j & 7);
    const bool inside = !(px >= fp.W || j >= rows || py >= y1);
    PixelCtx pc;
    pc.valid = false;
    if (inside) {
        const Img16 depth{depthP, fp.W, fp.H};
        setup_pixel(fp, depth, histCurr, hitMask, px, py, pc);
        float4* o = pixctx + 3u * ((size_t)py * fp.W + px);
        o[0] = make_float4(pc.worldPos.x, pc.worldPos.y, pc.worldPos.z, pc.valid ? pc.viewZ : 0.0f);
        if (pc.valid) {
            o[1] = make_float4(pc.geomNormal.x, pc.geomNormal.y, pc.geomNormal.z, pc.radius);
            o[2] = make_float4(pc.bias, pc.footprint, pc.cs, pc.sn);
        }
    }
#if K1_CUT
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (pc.valid) {
        // a ray runs from o + d * dirOffset to o + d * (dirOffset + radius) with dirOffset <= slope * 1.65 (:421-423,430) and
        // dot(n, d) >= 0 (:408-412; -0.01 allowed for rounding): inside the half ball of radius R around the offset origin o
        const float slope = gmax(pc.bias * 0.5f, pc.footprint * 0.02f);
        const float R = (pc.radius + slope * 1.65f) * 1.0001f + 1.0e-5f;
        const float oc[3] = {pc.origin.x, pc.origin.y, pc.origin.z}, nc[3] = {pc.geomNormal.x, pc.geomNormal.y, pc.geomNormal.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float side = gmin(1.0f, sqrt_(gmax(1.0f - nc[a] * nc[a], 0.0f)) + 0.02f);   // extent against the normal
            const float pad = 1.0e-4f + __builtin_fabsf(oc[a]) * 1.0e-6f;
            lo[a] = oc[a] - R * (nc[a] > 0.0f ? side : 1.0f) - pad;
            hi[a] = oc[a] + R * (nc[a] < 0.0f ? side : 1.0f) + pad;
        }
    }
    const int t8 = (int)(threadIdx.x & 63u) >> 3, r8 = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = group8_min(lo[a]), h = group8_max(hi[a]);
        if ((threadIdx.x & 7u) == 0u) { s_b[t8][r8][a] = l; s_b[t8][r8][3 + a] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 48u) {
        const int t = (int)threadIdx.x / 6, a = (int)threadIdx.x % 6;
        float v = s_b[t][0][a];
#pragma unroll
        for (int r = 1; r < 8; ++r) v = (a < 3) ? gmin(v, s_b[t][r][a]) : gmax(v, s_b[t][r][a]);
        const uint32_t tile_x = blockIdx.x * 8u + (uint32_t)t;
        if (cuts) s_tb[t][a] = v;
        else if (tile_x < tiles_x) ((float*)(tile_bounds + 2u * ((size_t)blockIdx.y * tiles_x + tile_x)))[a < 3 ? a : a + 1] = v;
    }
    // The entry cuts of the block's eight tiles, one per wave, in the same launch (cuts != nullptr; else k_tile_cut does it from
    // tile_bounds): the descent is a short chain of dependent loads, which the other blocks' streaming work covers.
    if (cuts) {
        __syncthreads();
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const uint32_t tile_x = blockIdx.x * 8u + w;
        if (tile_x < tiles_x) {
            const float amin[3] = {s_tb[w][0], s_tb[w][1], s_tb[w][2]}, amax[3] = {s_tb[w][3], s_tb[w][4], s_tb[w][5]};
            tile_cut_wave(sc, blockIdx.y * tiles_x + tile_x, amin, amax, cuts, cut_max, cut_small, l2_begin, l2_count, s_open[w],
                          (int)(threadIdx.x & 63u));
        }
    }
#endif
}

__global__ __launch_bounds__(256) void k_pix_resolve(FrameParams fp, const uint16_t* depthP, const uint16_t* histPrevP, uint16_t* histCurr,
                                                     uint32_t* hitMask, int y0, int y1, int rows, const float4* pixctx, const float2* vis)
{
    const int px = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63u);
    const int j = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);
    const int py = y0 + (j >> 3) * 8 * fp.stripeMul + (j & 7);
    if (px >= fp.W || j >= rows || py >= y1) return;
    const size_t o = (size_t)py * fp.W + px;
    const float4 c0 = pixctx[3u * o];
    if (!(c0.w > 0.0f)) return;                                 // early-out pixels stored their value in k_pix_setup
    const Img16 depth{depthP, fp.W, fp.H}, histPrev{histPrevP, fp.W, fp.H};
    const float2 v = vis[o];
    PixelCtx pc;
    pc.viewZ = c0.w; pc.worldPos = V3{c0.x, c0.y, c0.z};
    resolve_pixel(fp, depth, histPrev, histCurr, hitMask, px, py, pc, v.x, f2u(v.y));
}

struct NodeIn { float4 a, b, q0, q1, q2, q3; uint32_t rayhi; };   // one popped node item with its operands in flight

template <bool STATS, int RB>
__global__ __launch_bounds__(K1Q_BLOCK) __attribute__((amdgpu_waves_per_eu(K1_WAVES, K1_WAVES))) void k_trace_q(
    SceneDev sc, FrameParams fp, const uint16_t* depthP, const uint16_t* histPrevP, uint16_t* histCurr, uint32_t* hitMask,
    int y0, int y1, uint32_t tiles_x, uint32_t n_items, float4* rayBuf, uint32_t* tile_counter, unsigned long long* stats,
    const float4* pixctx, float2* visOut, const uint4* cuts)
{
    constexpr uint32_t RAYS = 1u << RB;
    constexpr uint32_t SH = 32u - (uint32_t)RB;              // item = ray << SH | payload
    constexpr uint32_t PMASK = (1u << SH) - 1u;
    extern __shared__ float4 smem_all[];
    const uint32_t wave_in_wg = K1_WG > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;
    float4* smem4 = smem_all + wave_in_wg * (uint32_t)(q_lds_bytes(RB) / 16u);     // this wave's part of the workgroup's LDS
    float4* s_ra = smem4;                                     // [RAYS] origin.xyz | tMin        (round 1: advanced origin | tMin')
    float4* s_rb = smem4 + RAYS;                              // [RAYS] 1/direction | closest hit so far (float bits)
#if K1_DIR_LDS
    float* s_rd = (float*)(smem4 + 2u * RAYS);                // [RAYS][3] direction (leaf steps, alpha tests, retry)
    uint32_t* s_nq = (uint32_t*)(s_rd + 3u * RAYS);           // node items
#else
    uint32_t* s_nq = (uint32_t*)(smem4 + 2u * RAYS);          // node items
#endif
    uint32_t* s_lq = s_nq + K1_NQ;                            // leaf items
    uint32_t* s_aq = s_lq + K1_LQ;                            // MASK candidates: ray << SH | triangle
    uint32_t* s_retry = s_aq + K1_AQ;                         // K1_EARLY_RETRY: one bit per ray, set when it was re-traced (8 words)
    (void)s_retry;
    uint4* s_cut = (uint4*)(s_retry + 12);                    // K1_CUT: [0] bound minimum | entries, [1 ..] the tile's cut entries
    (void)s_cut;

    const int rayCount = fp.rayCount;                         // :252
    const int lane = K1_WG > 1 ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : (int)threadIdx.x;
    const int W = fp.W, H = fp.H;
    const Img16 depth{depthP, W, H}, histPrev{histPrevP, W, H};
    (void)depth; (void)histPrev;
    const QSliceLayout lay = q_slice_layout(RB);
    float4* g_dir = rayBuf + ((size_t)blockIdx.x * K1_WG + wave_in_wg) * lay.total_f4;          // [RAYS] direction | tMin as generated
    uint32_t* retry_list = (uint32_t*)(g_dir + lay.dir_f4);
    uint32_t* g_nq = (uint32_t*)(g_dir + lay.dir_f4 + lay.retry_f4);
    uint32_t* g_lq = g_nq + K1_GNQ;
    const char* node_base = (const char*)sc.nodes;
    const char* tri_base = (const char*)sc.tris;
    TraceCounters cnt{0, 0, 0, 0};
    uint32_t nrays = 0, nspill = 0, ncuttest = 0, ncuthit = 0;
    (void)ncuttest; (void)ncuthit;
    // pixels per item P (a power of two), G = 64 / P lanes per pixel: lane = g * P + p generates rays g, g + G, ...
    const int P = q_item_pixels(RB, rayCount), G = 64 / P;
    const int plane = lane & (P - 1), grp = lane / P;
    const uint32_t items_per_block = 64u / (uint32_t)P;       // an 8x8 pixel block is cut into 64 / P items
    // STATS build only: shader cycles (s_memtime) per phase of this wave, summed over waves into stats[8..]
    uint32_t ph_prev = STATS ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;
    uint32_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // raygen, inject/refill, node, leaf, alpha, retry list, resolve, item pull, setup_pixel, visibility loop
    uint32_t ph_n[4] = {0, 0, 0, 0};             // node steps / leaf steps / alpha steps / node items
#define VKR_STAMP(K) { if (STATS) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); ph[K] += now_ - ph_prev; ph_prev = now_; } }

    // Work distribution.  K1_SLABS == 1: one counter, items in row-major order.  K1_SLABS == 8: the tile columns are cut
    // into 8 vertical slabs with a counter each; a wave starts on the slab of its own XCD (HW_REG_XCC_ID) and moves on to
    // the next slab when that one is drained, so the waves that share an L2 work on neighbouring pixels (speed only: any
    // wave may take any item).
    uint32_t slab = 0, slabs_done = 0;
    (void)slabs_done;
    if (K1_SLABS > 1) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        slab = (xcc & 0xfu) % (uint32_t)K1_SLABS;
    }
    const uint32_t item_rows = n_items / tiles_x;             // rows of items in the launch
#if K1_WG > 1
    // K1_WG waves per workgroup: the counters hand out UNITS of K1_WG consecutive items in tile-major order (the items of a
    // tile follow each other, then the next tile of the row), and the waves of a workgroup -- which share a CU, and with it
    // the L1 -- take the items of one unit at the same time: their rays start centimetres apart and walk the same nodes and
    // triangles, so what one wave pulled into L1 the others find there.  No barrier: a wave takes the next free (unit, slot)
    // of its workgroup from an LDS counter; the wave that gets slot 0 fetches the unit from the slab counters and publishes
    // it in a small LDS ring, the others wait for that entry (the publisher is resident and running: the wait is bounded).
    // A ring entry is reused K1_RING units later; its publisher first waits until the K1_WG readers of the previous use have
    // counted themselves in (they do so a few instructions after taking their slot, before any tracing).
    uint32_t* s_take = (uint32_t*)(smem_all + (uint32_t)K1_WG * (uint32_t)(q_lds_bytes(RB) / 16u));   // [0] taken, [4 + 4j] tag, unit, readers so far
    if (threadIdx.x < 4u + 4u * K1_RING) s_take[threadIdx.x] = (threadIdx.x >= 4u && (threadIdx.x & 3u) == 0u) ? 0xffffffffu : 0u;
    __syncthreads();
    const uint32_t home_slab = slab;
    const uint32_t unit_items = (items_per_block > (uint32_t)K1_WG ? items_per_block : (uint32_t)K1_WG) * (uint32_t)K1_UNIT_TILES;
#endif
    for (;;) {
        uint32_t item = 0;
#if K1_WG > 1
        VKR_PRIO_HI      // the counter's round trip and the pixel records' loads come next
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(&s_take[0], 1u);
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        const uint32_t j = k / unit_items, wslot = k % unit_items, ring = 4u + 4u * (j % K1_RING);
        uint32_t unit = 0xfffffffeu;                            // slab << 28 | unit of that slab; 0xfffffffe = nothing left
        if (wslot == 0u) {
            if (lane == 0) {
                for (uint32_t t = 0; t < (uint32_t)K1_SLABS; ++t) {           // home slab first; drained counters only grow
                    const uint32_t sl = (home_slab + t) % (uint32_t)K1_SLABS;
                    const uint32_t sc0 = sl * tiles_x / (uint32_t)K1_SLABS, sc1 = (sl + 1u) * tiles_x / (uint32_t)K1_SLABS;
                    const uint32_t u = atomicAdd(tile_counter + sl * 16u, 1u);
                    if ((unsigned long long)u * unit_items < (unsigned long long)(sc1 - sc0) * item_rows) { unit = (sl << 28) | u; break; }
                }
                while (__hip_atomic_load(&s_take[ring + 2u], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (j / K1_RING) * unit_items)
                    __builtin_amdgcn_s_sleep(1);
                __hip_atomic_store(&s_take[ring + 1u], unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&s_take[ring], j, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (lane == 0) {
            while (__hip_atomic_load(&s_take[ring], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != j) __builtin_amdgcn_s_sleep(1);
            unit = __hip_atomic_load(&s_take[ring + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&s_take[ring + 2u], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
        if (unit == 0xfffffffeu) break;                         // every counter is drained: every later unit of this workgroup says so too
        slab = unit >> 28;
        const uint32_t c0 = slab * tiles_x / (uint32_t)K1_SLABS, c1 = (slab + 1u) * tiles_x / (uint32_t)K1_SLABS;
        const uint32_t slab_items = (c1 - c0) * item_rows;
        item = (unit & 0x0fffffffu) * unit_items + wslot;
        if (item >= slab_items) continue;                       // the slab's last unit may be short
        VKR_STAMP(7)
        // tile-major: item = (row of tiles * tiles of the slab + tile) * items per tile + item of the tile
        const uint32_t sub = item % items_per_block, tl = item / items_per_block;
        const uint32_t tile_x = c0 + tl % (c1 - c0), trow = tl / (c1 - c0);
#else
        const uint32_t c0 = slab * tiles_x / (uint32_t)K1_SLABS, c1 = (slab + 1u) * tiles_x / (uint32_t)K1_SLABS;
        const uint32_t slab_items = (c1 - c0) * item_rows;
        VKR_PRIO_HI      // the counter's round trip and the pixel records' loads come next
        if (lane == 0) item = atomicAdd(tile_counter + slab * 16u, 1u);
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
        if (item >= slab_items) {                               // this slab is drained (every wave gets here: counters only grow)
            if (++slabs_done >= (uint32_t)K1_SLABS) break;
            slab = (slab + 1u) % (uint32_t)K1_SLABS;
            continue;
        }
        VKR_STAMP(7)
        const uint32_t r_ = item / (c1 - c0);                                 // row of items: 8-row stripe * sub-block
        const uint32_t tile_x = c0 + item % (c1 - c0);
        const uint32_t sub = r_ % items_per_block, trow = r_ / items_per_block;
#endif
        const uint32_t idx = sub * (uint32_t)P + (uint32_t)plane;             // pixel of the 8x8 block, row-major
        const int px = (int)tile_x * 8 + (int)(idx & 7u);
        const int py = y0 + (int)trow * 8 * fp.stripeMul + (int)(idx >> 3);

        // ------------------------------------------------------------------ phase 1: per pixel
        PixelCtx pc;
        pc.valid = false;
        pc.radius = 0.0f; pc.viewZ = 0.0f; pc.worldPos = v3(0, 0, 0);
#if K1_CUT
        // the tile's cut (k_tile_cut): header + entries, one uint4 per lane, in flight with the pixel records
        // Slots past the cut's last entry (what k_tile_cut did not write, and three more: the injection reads four entries per
        // turn) hold a box 65 km off the bound's corner, which no ray of <= 2 m reaches.
        if (lane <= K1_CUT_MAX + 3) {
            const uint4* ct = cuts + (size_t)(trow * tiles_x + tile_x) * (K1_CUT_MAX + 1);
            const uint4 hdr = ct[0];
            const uint4 ent = ct[lane <= K1_CUT_MAX ? lane : 0];
            s_cut[lane] = ((uint32_t)lane <= hdr.w) ? ent : make_uint4(0x7bff7bffu, 0x7bff7bffu, 0x7bff7bffu, 0u);
        }
#endif
#if K1_SPLIT
        // k_pix_setup ran main():237-400 for every pixel of the launch; what make_ray needs is rebuilt from its record
        pc.geomNormal = v3(0, 1, 0); pc.origin = v3(0, 0, 0); pc.tT = v3(0, 0, 0); pc.tB = v3(0, 0, 0);
        pc.bias = 0.0f; pc.footprint = 0.0f; pc.tMin = 0.0f; pc.cs = 1.0f; pc.sn = 0.0f;
        if (px < W && py < y1) {
            const float4* cp = pixctx + 3u * ((size_t)py * W + px);
            const float4 c0 = cp[0];
            if (c0.w > 0.0f) {
                const float4 c1 = cp[1], c2 = cp[2];
                pc.valid = true;
                pc.worldPos = V3{c0.x, c0.y, c0.z}; pc.viewZ = c0.w;
                pc.geomNormal = V3{c1.x, c1.y, c1.z}; pc.radius = c1.w;
                pc.bias = c2.x; pc.footprint = c2.y; pc.cs = c2.z; pc.sn = c2.w;
                pc.origin = offsetRay(pc.worldPos, pc.geomNormal);                      // :375
                buildTbn(pc.geomNormal, &pc.tT, &pc.tB);                                // :376
                pc.tMin = gmax(1e-4f, pc.bias * 0.25f);                                 // :377
            }
        }
#else
        if (px < W && py < y1) setup_pixel(fp, depth, histCurr, hitMask, px, py, pc);   // :233-235
#endif
        VKR_PRIO_LO      // ray generation is arithmetic
        VKR_STAMP(8)
        const bool valid = pc.valid;
        const float radius = pc.radius, viewZ = pc.viewZ;     // kept across the queue loop (K1_SPLIT: the radius only)
        const V3 worldPos = pc.worldPos;
        (void)viewZ; (void)worldPos;
        const unsigned long long pmask = (P == 64) ? ~0ull : ((1ull << P) - 1ull);
        const unsigned long long vmask = __ballot(valid) & pmask;
        const uint32_t nvalid = (uint32_t)__popcll(vmask);
        if (nvalid == 0) { VKR_STAMP(0) continue; }   // wave-uniform: nothing to trace, nothing to resolve
        const uint32_t rank = (uint32_t)__popcll(vmask & ((1ull << plane) - 1ull));
        const bool owner = valid && grp == 0;                 // the lane that resolves the pixel
        if (valid) {
            for (int i = grp; i < rayCount; i += G) {                               // :403-430
                V3 rayOrigin, rayDir;
                float tMinRay;
                make_ray(pc, rayCount, i, &rayOrigin, &rayDir, &tMinRay);
                const uint32_t k = (uint32_t)i * nvalid + rank;
                s_ra[VKR_BOUND(k, RAYS, 1)] = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, tMinRay);
                s_rb[k] = make_float4(safe_rcp(rayDir.x), safe_rcp(rayDir.y), safe_rcp(rayDir.z), pc.radius);   // tMax = radius (:430)
#if K1_EARLY_RETRY
                ((float*)retry_list)[k] = pc.radius;                    // the ray's tMax again, for its re-trace
#endif
#if K1_DIR_LDS
                s_rd[3u * k] = rayDir.x; s_rd[3u * k + 1u] = rayDir.y; s_rd[3u * k + 2u] = rayDir.z;
                ((float*)&g_dir[k])[3] = tMinRay;                       // tMin as generated: read back by the resolve only
#else
                g_dir[k] = make_float4(rayDir.x, rayDir.y, rayDir.z, tMinRay);
#endif
                if (STATS && K1_CUT) nrays++;
            }
        }
#if K1_EARLY_RETRY
        if (lane < 8) s_retry[lane] = 0u;
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        VKR_STAMP(0)

        // ------------------------------------------------------------------ phase 2: queues
        const uint32_t total = (sc.tri_count != 0u) ? nvalid * (uint32_t)rayCount : 0u;   // an empty scene has a root, but nothing walks it
        uint32_t retry_mask = 0;     // bit i: this pixel's ray i was re-traced (two-round form)
        (void)retry_mask;
        uint32_t nretry = 0;
        for (int round = 0; round < (K1_EARLY_RETRY ? 1 : 2); ++round) {
#if K1_CUT
            // entries of the tile's cut to inject the rays at (written before the fence above); 0 when there are no rays
            const uint32_t set_size = total ? (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cut[0].w) : 0u;
#else
            const uint32_t set_size = (round == 0) ? total : nretry;
#endif
            if (set_size == 0) break;
            uint32_t next = 0, nq = 0, lq = 0, aq = 0, gnq = 0, glq = 0;      // wave-uniform
            uint32_t half = 0;                                                // K1_CUT: which 64 rays of the item are being injected
            const uint32_t nhalves = (total + 63u) / 64u;
            (void)half; (void)nhalves;
            auto ray_dir = [&](uint32_t ray) -> V3 {
#if K1_DIR_LDS
                return V3{s_rd[3u * ray], s_rd[3u * ray + 1u], s_rd[3u * ray + 2u]};
#else
                const float4 d4 = g_dir[ray];
                return V3{d4.x, d4.y, d4.z};
#endif
            };
#if K1_EARLY_RETRY
            // traceAoRay's retry (:193-200) without a second round.  The first query's closest hit is <= 1.5 tMin exactly when SOME
            // committed hit is (closest hits only get closer), so the lane that commits such a hit re-traces the ray at once: it
            // claims the ray's bit, rewrites the LDS record (advanced origin, tMin', tMax again) and pushes a root item.  Items of
            // the ray that are still queued now run against the new record: they only repeat parts of the re-trace (the closest
            // hit is a minimum, so repeats change nothing).  `trig` lanes have already issued their atomicMin on the old record;
            // the rewrite comes after it in program order and discards it, as :199 discards the first query's distance.
            auto retry_now = [&](bool trig, uint32_t ray) {
                if (__ballot(trig) == 0ull) return;
                bool win = false;
                if (trig) {
                    const uint32_t bit = 1u << (ray & 31u);
                    win = (atomicOr(&s_retry[ray >> 5], bit) & bit) == 0u;
                }
                const unsigned long long wm = __ballot(win);
                if (wm == 0ull) return;
                if (win) {
                    const float4 a = s_ra[ray];
                    const V3 d = ray_dir(ray);
                    const float adv = gmax(a.w * 2.0f, 2e-4f);                      // :194-196
                    const V3 ro = V3{a.x, a.y, a.z} + V3{d.x, d.y, d.z} * adv;
                    s_ra[ray] = make_float4(ro.x, ro.y, ro.z, gmax(a.w, 1e-4f));
                    s_rb[ray].w = ((const float*)retry_list)[ray];
                    if (STATS) nrays++;
                }
                const uint32_t cnt = (uint32_t)__popcll(wm);
                if (nq + cnt <= (uint32_t)K1_NQ) { if (win) s_nq[nq + mbcnt64(wm)] = ray << SH; nq += cnt; }
                else { if (win) g_nq[VKR_BOUND(gnq + mbcnt64(wm), K1_GNQ, 1u << 22)] = ray << SH; gnq += cnt; if (STATS) nspill += (lane == 0) ? cnt : 0u; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            };
#endif
            // ---------------- alpha step: MASK candidates (:163-173), n items from s_aq[base..]
            auto alpha_step = [&](uint32_t n, uint32_t base) {
                bool trig = false;
                uint32_t tray = 0;
                VKR_PRIO_HI          // a chain of dependent loads (triangle + alpha record, texture table, texels) with little arithmetic between them
                if ((uint32_t)lane < n) {
                    const uint32_t it = s_aq[VKR_BOUND(base + (uint32_t)lane, K1_AQ, 2)];
                    const uint32_t tri = VKR_BOUND(it & PMASK, sc.tri_count, 4), ray = VKR_BOUND(it >> SH, RAYS, 8);
                    const float4 a = s_ra[ray];
                    const V3 d = ray_dir(ray);
                    const float4* tp = (const float4*)(tri_base + (size_t)tri * 48u);
                    const float4 ta = tp[0], tb4 = tp[1], tc = tp[2];
                    const float4* ar = sc.tri_alpha + (size_t)tri * 3u;     // UVs, cutoff, texture descriptor: in flight with the triangle
                    const float4 ar0 = ar[0], ar1 = ar[1], ar2 = ar[2];
                    float u, v, t;
                    const float tbest = s_rb[ray].w;
                    if (tri_hit(V3{a.x, a.y, a.z}, V3{d.x, d.y, d.z}, ta, tb4, tc, a.w, tbest, &u, &v, &t) &&
                        alpha_test_rec(sc, ar0, ar1, ar2, u, v)) {
                        atomicMin((uint32_t*)&s_rb[ray].w, f2u(t));
                        trig = t <= a.w * 1.5f;                                     // :193
                        tray = ray;
                    }
                }
#if K1_EARLY_RETRY
                retry_now(trig, tray);
#else
                (void)trig; (void)tray;
#endif
            };
            // ---------------- queue loop.  Common path per iteration: one compare for "leaf step?", one for "node queue
            // needs attention?", then a step.  Refills from the wave's global overflow stack and the throttled step width
            // are the rare paths.
            for (;;) {
                bool do_leaf = lq >= K1_LEAF_AT;
                if (!do_leaf && (nq < K1_STEP || (glq | gnq) != 0u)) {
                    if (glq > 0u) {
                        // ---------------- leaf items back from the global stack (before anything else pushes leaves)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        uint32_t m = (uint32_t)K1_LQ - lq;
                        if (m > glq) m = glq;
                        glq -= m;
                        for (uint32_t j = (uint32_t)lane; j < m; j += 64u) s_lq[VKR_BOUND(lq + j, K1_LQ, 16)] = g_lq[VKR_BOUND(glq + j, K1_GLQ, 32)];
                        lq += m;
                        continue;
                    }
                    if (gnq > 0u) {
                        // ---------------- node items back from the global stack
                        if (nq < K1_STEP) {
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            uint32_t m = 2u * K1_STEP;
                            if (m > gnq) m = gnq;
                            if (m > (uint32_t)K1_NQ - nq) m = (uint32_t)K1_NQ - nq;
                            gnq -= m;
                            for (uint32_t j = (uint32_t)lane; j < m; j += 64u) s_nq[VKR_BOUND(nq + j, K1_NQ, 64)] = g_nq[VKR_BOUND(gnq + j, K1_GNQ, 128)];
                            nq += m;
                        }
                    } else if (K1_CUT ? (half < nhalves) : (next < set_size)) {
#if K1_CUT
                        // ---------------- inject rays at the next entries of the tile's cut: 64 rays at a time (rays 0..63 go through
                        // all entries first, then rays 64..127 -- as many rays in flight as the root injection had), each against
                        // the entry's box, the ones that overlap it queued at the entry's node or leaf.  Entries follow each other
                        // until a step's worth of items waits.  A re-traced ray (early retry) is tested with its new record like
                        // any queued item of it; its re-trace from the root is what makes it complete.
                        const uint32_t ray0 = (uint32_t)lane + 64u * half;
                        const bool in0 = ray0 < total;
                        const float4 A0 = s_ra[in0 ? ray0 : 0u], B0 = s_rb[in0 ? ray0 : 0u];
                        const float4 cut_hdr = __builtin_bit_cast(float4, s_cut[0]);    // read here: three registers less across the queue loop
                        // t = (bound minimum + offset - origin) / direction = fma(offset, 1/d, (minimum - origin) * 1/d)
                        const float c0x = (cut_hdr.x - A0.x) * B0.x, c0y = (cut_hdr.y - A0.y) * B0.y, c0z = (cut_hdr.z - A0.z) * B0.z;
                        const float tb0 = in0 ? B0.w * 1.000001f : -3.0e38f;
                        bool more;
                        do {
                            // four entries per turn (the list is padded with entries no ray can hit): four independent box tests
                            // between one round of LDS reads and one round of appends
                            const uint4 e0 = s_cut[1u + next], e1 = s_cut[2u + next], e2 = s_cut[3u + next], e3 = s_cut[4u + next];   // wave-uniform addresses
#define VKR_CUT_TEST(H, E)                                                                                          \
    bool H;                                                                                                         \
    {                                                                                                               \
        const float x0 = fma_(h_lo(E.x), B0.x, c0x), x1 = fma_(h_hi(E.y), B0.x, c0x);                               \
        const float y0_ = fma_(h_hi(E.x), B0.y, c0y), y1_ = fma_(h_lo(E.z), B0.y, c0y);                             \
        const float z0 = fma_(h_lo(E.y), B0.z, c0z), z1 = fma_(h_hi(E.z), B0.z, c0z);                               \
        const float tn = vmax3(vmin2(x0, x1), vmin2(y0_, y1_), vmin2(z0, z1));                                      \
        const float tf = vmin3(vmax2(x0, x1), vmax2(y0_, y1_), vmax2(z0, z1));                                      \
        H = vmax2(tn, A0.w) <= vmin2(tf, tb0);                                                                      \
    }
                            VKR_CUT_TEST(hit0, e0) VKR_CUT_TEST(hit1, e1) VKR_CUT_TEST(hit2, e2) VKR_CUT_TEST(hit3, e3)
#undef VKR_CUT_TEST
                            const unsigned long long m0 = __ballot(hit0), m1 = __ballot(hit1), m2 = __ballot(hit2), m3 = __ballot(hit3);
                            const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2), n3 = (uint32_t)__popcll(m3);
                            const bool f0 = (int)e0.w < 0, f1 = (int)e1.w < 0, f2 = (int)e2.w < 0, f3 = (int)e3.w < 0;    // wave-uniform: leaf entries
                            const uint32_t totn = (f0 ? 0u : n0) + (f1 ? 0u : n1) + (f2 ? 0u : n2) + (f3 ? 0u : n3);
                            const uint32_t totl = (n0 + n1) + (n2 + n3) - totn;
                            if (STATS && !K1_EXP_RUNS_ON) { ncuttest += in0 ? (set_size - next < 4u ? set_size - next : 4u) : 0u; ncuthit += (hit0 ? 1u : 0u) + (hit1 ? 1u : 0u) + (hit2 ? 1u : 0u) + (hit3 ? 1u : 0u); }
                            const uint32_t rhi = ray0 << SH;                            // item: ray | node index, or ray | first * 2 | count - 1 of a leaf
                            if (nq + totn <= (uint32_t)K1_NQ && lq + totl <= (uint32_t)K1_LQ) {
#define VKR_CUT_PUSH(H, F, M, N, E)                                                                                 \
    if (F) { if (H) s_lq[VKR_BOUND(lq + mbcnt64(M), K1_LQ, 1u << 26)] = rhi | (E.w & 0x7fffffffu); lq += N; }        \
    else { if (H) s_nq[VKR_BOUND(nq + mbcnt64(M), K1_NQ, 1u << 24)] = rhi | E.w; nq += N; }
                                VKR_CUT_PUSH(hit0, f0, m0, n0, e0) VKR_CUT_PUSH(hit1, f1, m1, n1, e1)
                                VKR_CUT_PUSH(hit2, f2, m2, n2, e2) VKR_CUT_PUSH(hit3, f3, m3, n3, e3)
#undef VKR_CUT_PUSH
                            } else {
                                // rare: per entry, LDS while it fits, the wave's global stacks beyond
#define VKR_CUT_PUSH(H, F, M, N, E)                                                                                 \
    if (F) {                                                                                                        \
        if (lq + N <= (uint32_t)K1_LQ) { if (H) s_lq[VKR_BOUND(lq + mbcnt64(M), K1_LQ, 1u << 26)] = rhi | (E.w & 0x7fffffffu); lq += N; }   \
        else { if (H) g_lq[VKR_BOUND(glq + mbcnt64(M), K1_GLQ, 1u << 27)] = rhi | (E.w & 0x7fffffffu); glq += N; if (STATS) nspill += (lane == 0) ? N : 0u; }   \
    } else {                                                                                                        \
        if (nq + N <= (uint32_t)K1_NQ) { if (H) s_nq[VKR_BOUND(nq + mbcnt64(M), K1_NQ, 1u << 24)] = rhi | E.w; nq += N; }   \
        else { if (H) g_nq[VKR_BOUND(gnq + mbcnt64(M), K1_GNQ, 1u << 25)] = rhi | E.w; gnq += N; if (STATS) nspill += (lane == 0) ? N : 0u; }   \
    }
                                VKR_CUT_PUSH(hit0, f0, m0, n0, e0) VKR_CUT_PUSH(hit1, f1, m1, n1, e1)
                                VKR_CUT_PUSH(hit2, f2, m2, n2, e2) VKR_CUT_PUSH(hit3, f3, m3, n3, e3)
#undef VKR_CUT_PUSH
                            }
                            next += 4u;
                            more = next < set_size;
                        } while (more && nq < K1_STEP && lq < K1_LEAF_AT && (glq | gnq) == 0u);
                        if (!more) { next = 0u; ++half; }                               // on to the other 64 rays
                    }
#else
                        // ---------------- inject the next rays at the root
                        uint32_t m = set_size - next;
                        if (m > K1_INJECT) m = K1_INJECT;
                        if (m > (uint32_t)K1_NQ - nq) m = (uint32_t)K1_NQ - nq;
                        for (uint32_t j = (uint32_t)lane; j < m; j += 64u) {
                            const uint32_t ray = round ? retry_list[next + j] : next + j;
                            s_nq[VKR_BOUND(nq + j, K1_NQ, 256)] = VKR_BOUND(ray, RAYS, 512) << SH;   // root = node 0
                            if (STATS) nrays++;
                        }
                        nq += m; next += m;
                    }
#endif
                    VKR_STAMP(1)
                    if (nq == 0u) {                     // nothing left to descend: drain the leaves, then the MASK candidates
                        if (lq > 0u) do_leaf = true;
                        else if ((glq | gnq) != 0u || (K1_CUT && half < nhalves)) continue;   // K1_CUT: entries that queued nothing, or everything went to the global stacks
                        else if (aq > 0u) { alpha_step(aq, 0u); aq = 0u; if (STATS) ph_n[2]++; VKR_STAMP(4) continue; }
                        else break;
                    }
                }
                if (do_leaf) {
                    // ---------------- leaf step: queryAoClosestHit's candidate loop (:125-182) for <= 64 leaves
                    const uint32_t n = lq < 64u ? lq : 64u;
                    lq -= n;
                    const bool act = (uint32_t)lane < n;
                    VKR_PRIO_HI
                    const uint32_t it = act ? s_lq[VKR_BOUND(lq + (uint32_t)lane, K1_LQ, 1024)] : 0u;      // item 0: ray 0, triangle 0 (valid addresses)
                    const uint32_t ray = VKR_BOUND(it >> SH, RAYS, 2048), pl = VKR_BOUND(it & PMASK, 2u * sc.tri_count, 4096);
                    const uint32_t cntT = act ? (pl & 1u) + 1u : 0u;
                    const uint32_t toff = (pl & ~1u) * 24u;                        // first triangle * 48 bytes
                    const float4 a = s_ra[ray];
                    const V3 d = ray_dir(ray);
                    const float tbest0 = s_rb[ray].w;
#if K1_EARLY_RETRY
                    const uint32_t retried0 = (s_retry[ray >> 5] >> (ray & 31u)) & 1u;
#endif
                    const float4* tp = (const float4*)(tri_base + toff);
                    const float4 ta0 = tp[0], tb0 = tp[1], tc0 = tp[2];
                    const unsigned long long m2 = __ballot(cntT > 1u);
                    float4 ta1 = ta0, tb1 = tb0, tc1 = tc0;
                    if (m2 != 0ull) {                                               // wave-uniform
                        const float4* tq = tp + ((cntT > 1u) ? 3 : 0);
                        ta1 = tq[0]; tb1 = tq[1]; tc1 = tq[2];
                    }
                    VKR_PRIO_LO
                    const V3 ro = V3{a.x, a.y, a.z}, rdir = V3{d.x, d.y, d.z};
                    const float tmin = a.w;
                    float tbest = tbest0;
#define VKR_LEAF_TRI(J, TA, TB, TC, AM)                                                                             \
    {                                                                                                               \
        bool cand = false;   /* MASK candidate: geometric hit that still needs the alpha test */                    \
        if ((uint32_t)(J) < cntT) {                                                                                 \
            if (STATS) { cnt.tris++; if ((int)(__ffsll((long long)(AM)) - 1) == lane) cnt.wave_tri_iters++; }       \
            float u, v, t;                                                                                          \
            const bool geo = tri_hit(ro, rdir, TA, TB, TC, tmin, tbest, &u, &v, &t);                                \
            const uint32_t cls = f2u(TA.w) >> 28;                   /* candidate filter :142-173 */                 \
            if (geo && cls == 0u) tbest = t;                                                                        \
            cand = geo && cls == 2u;                                                                                \
        }                                                                                                           \
        /* alpha tests are not run here at one or two lanes' width: candidates queue up and are tested a wave at */ \
        /* a time (accepting a hit later only delays pruning; the minimum is the same) */                           \
        const unsigned long long cm = __ballot(cand);                                                               \
        if (cm != 0ull) {                                                                                           \
            if (cand) s_aq[VKR_BOUND(aq + mbcnt64(cm), K1_AQ, 8192)] = (ray << SH) | ((pl >> 1) + (uint32_t)(J));                           \
            aq += (uint32_t)__popcll(cm);                                                                           \
            if (aq >= 64u) { aq -= 64u; VKR_STAMP(3) alpha_step(64u, aq); if (STATS) ph_n[2]++; VKR_STAMP(4) }      \
        }                                                                                                           \
    }
                    VKR_LEAF_TRI(0, ta0, tb0, tc0, __ballot(act))
                    if (m2 != 0ull) VKR_LEAF_TRI(1, ta1, tb1, tc1, m2)
#undef VKR_LEAF_TRI
#if K1_EARLY_RETRY
                    // An alpha step in the middle of this step may have re-traced this lane's ray (rewritten its record): what the
                    // lane computed belongs to the old record and is dropped -- the re-trace starts at the root and meets this leaf again.
                    const bool same_record = ((s_retry[ray >> 5] >> (ray & 31u)) & 1u) == retried0;
#else
                    const bool same_record = true;
#endif
                    VKR_PRIO_HI
                    if (tbest < tbest0 && same_record) atomicMin((uint32_t*)&s_rb[ray].w, f2u(tbest));   // t > tmin >= 0: bits order like floats
#if K1_EARLY_RETRY
                    retry_now(tbest < tbest0 && tbest <= tmin * 1.5f && same_record, ray); // :193
#endif
                    if (STATS) ph_n[1]++;
                    VKR_STAMP(3)
                    continue;
                }
                // ---------------- node step: pops n <= K1_STEP items, pushes <= 4n node items and <= 4n leaf items.
                // Pushes go to LDS while they fit and to the wave's global stack beyond that (rare); the step is narrowed
                // only when LDS + global capacity could run out: n shrinks down to one item per step = strict depth-first,
                // whose growth is bounded by 3 x tree depth (KQ_GUARD).
                uint32_t n = nq < K1_STEP ? nq : K1_STEP;
                // (RAYS entries are kept back for the root items of self-hit re-traces, which retry_now pushes unthrottled)
                if (nq + gnq + 3u * K1_STEP > (uint32_t)K1_NQ + K1_GNQ - KQ_GUARD - RAYS) {            // rare
                    const uint32_t cap = (uint32_t)K1_NQ + K1_GNQ - KQ_GUARD - RAYS, used = nq + gnq;
                    const uint32_t fit = (cap > used ? cap - used : 0u) / 3u;
                    if (n > fit) n = fit ? fit : 1u;
                }
                nq -= n;
                // Every lane runs the step; lanes without an item work on item 0 (ray 0 at the root: valid addresses, no
                // side effects) and are masked out of the hit tests by an exit distance below any entry distance.
                auto fetch = [&](uint32_t it) -> NodeIn {
                    NodeIn in;
                    const uint32_t ray = VKR_BOUND(it >> SH, RAYS, 16384), node = VKR_BOUND(it & PMASK, sc.node_count, 32768);
                    const float4* np = (const float4*)(node_base + ((size_t)node << 6));
                    in.q0 = np[0]; in.q1 = np[1]; in.q2 = np[2]; in.q3 = np[3];
#ifdef VKR_EXP_EXTRA_LOAD   // experiment: what one more 16-byte load per node visit costs; the value is its byte offset from the node
                            // (32 = the q2 quarter again, 64 = the next node's first quarter)
                    { float4 ex; asm volatile("global_load_dwordx4 %0, %1, off offset:%2\n s_waitcnt vmcnt(0)" : "=v"(ex) : "v"(np), "n"(VKR_EXP_EXTRA_LOAD) : "memory");
                      in.q3.x = u2f(f2u(in.q3.x) | (f2u(ex.x) & 0u)); }
#endif
                    in.a = s_ra[ray]; in.b = s_rb[ray];
                    in.rayhi = it & ~PMASK;
                    return in;
                };
                auto process = [&](const NodeIn& in, bool act) {
                    const float4 q0 = in.q0, q1 = in.q1, q2 = in.q2, q3 = in.q3;
                    // padded: culling may only err towards extra visits; lanes without an item get an exit distance below
                    // any entry distance (tn >= tmin > 0), so none of their four tests can pass
                    const float tb = act ? in.b.w * 1.000001f : -3.0e38f;
                    if (STATS) { if (act) { cnt.nodes++; if ((int)(__ffsll((long long)__ballot(act)) - 1) == lane) cnt.wave_node_iters++; } }
                    const float tmin = in.a.w;
                    const V3 id = V3{in.b.x, in.b.y, in.b.z};
#ifdef VKR_EXP_EXTRA_VALU   // experiment: what N more slow-class VALU instructions per node step cost (sensitivity to VALU issue)
                    { float dmy = tb;
#pragma unroll
                      for (int e_ = 0; e_ < VKR_EXP_EXTRA_VALU; ++e_) asm volatile("v_max_f32 %0, %0, %0" : "+v"(dmy));
                      asm volatile("" :: "v"(dmy)); }
#endif
                    // Slab tests straight on the quantised planes: plane = fma(q, s, o), t = (plane - ro) * id is evaluated
                    // as fma(q, s*id, (o - ro)*id): one fma per plane.  The direction sign picks the near/far byte quads
                    // once per axis; unused slots are stored inverted and miss by themselves.  Rounding of this form is
                    // covered by the leaf padding (bvh_build.hip tri_box) and by the outward quantisation.
                    const float ax = q2.z * id.x, ay = q2.w * id.y, az = q0.w * id.z;
                    const float bx = (q0.x - in.a.x) * id.x, by = (q0.y - in.a.y) * id.y, bz = (q0.z - in.a.z) * id.z;
                    const uint32_t nx4 = id.x >= 0.0f ? f2u(q1.x) : f2u(q1.w), fx4 = id.x >= 0.0f ? f2u(q1.w) : f2u(q1.x);
                    const uint32_t ny4 = id.y >= 0.0f ? f2u(q1.y) : f2u(q2.x), fy4 = id.y >= 0.0f ? f2u(q2.x) : f2u(q1.y);
                    const uint32_t nz4 = id.z >= 0.0f ? f2u(q1.z) : f2u(q2.y), fz4 = id.z >= 0.0f ? f2u(q2.y) : f2u(q1.z);
                    const uint32_t r0 = f2u(q3.x), r1 = f2u(q3.y), r2 = f2u(q3.z), r3 = f2u(q3.w);
                    // two children per packed fma (v_pk_fma_f32 rounds each half like the scalar fma)
#define VKR_PLANES(W, A, B, P01, P23)                                                                               \
    const f2v P01 = pk_fma(f2v{ubyte_f(W, 0), ubyte_f(W, 1)}, A, B), P23 = pk_fma(f2v{ubyte_f(W, 2), ubyte_f(W, 3)}, A, B);
                    VKR_PLANES(nx4, ax, bx, nx01, nx23) VKR_PLANES(ny4, ay, by, ny01, ny23) VKR_PLANES(nz4, az, bz, nz01, nz23)
                    VKR_PLANES(fx4, ax, bx, fx01, fx23) VKR_PLANES(fy4, ay, by, fy01, fy23) VKR_PLANES(fz4, az, bz, fz01, fz23)
#undef VKR_PLANES
                    // a NaN entry distance fails the comparison
#define VKR_SLABQ(H, NX, NY, NZ, FX, FY, FZ) const bool H = vmax2(vmax3(NX, NY, NZ), tmin) <= vmin2(vmin3(FX, FY, FZ), tb);
                    VKR_SLABQ(h0, nx01.x, ny01.x, nz01.x, fx01.x, fy01.x, fz01.x)
                    VKR_SLABQ(h1, nx01.y, ny01.y, nz01.y, fx01.y, fy01.y, fz01.y)
                    VKR_SLABQ(h2, nx23.x, ny23.x, nz23.x, fx23.x, fy23.x, fz23.x)
                    VKR_SLABQ(h3, nx23.y, ny23.y, nz23.y, fx23.y, fy23.y, fz23.y)
#undef VKR_SLABQ
                    // append the hit children: internal nodes to the node queue, leaves to the leaf queue, slot by slot
                    // (measured in round 1: sorting them near-to-far first saves 1 % of the visits and costs more than that;
                    // one lane-major prefix sum instead of per-slot ballots makes a ray's siblings pop together and costs 9 %)
                    const uint32_t rayhi = in.rayhi, rayhi_leaf = in.rayhi + 0x80000000u;   // + leaf reference: bit 31 cancels
                    const bool l0 = (int)r0 < 0, l1 = (int)r1 < 0, l2 = (int)r2 < 0, l3 = (int)r3 < 0;
                    VKR_PRIO_HI
                    const unsigned long long hm0 = __ballot(h0), hm1 = __ballot(h1), hm2 = __ballot(h2), hm3 = __ballot(h3);
                    const unsigned long long lm0 = __ballot(l0), lm1 = __ballot(l1), lm2 = __ballot(l2), lm3 = __ballot(l3);
                    const unsigned long long mn0 = hm0 & ~lm0, mn1 = hm1 & ~lm1, mn2 = hm2 & ~lm2, mn3 = hm3 & ~lm3;
                    const unsigned long long ml0 = hm0 & lm0, ml1 = hm1 & lm1, ml2 = hm2 & lm2, ml3 = hm3 & lm3;
                    const uint32_t cn0 = (uint32_t)__popcll(mn0), cn1 = (uint32_t)__popcll(mn1), cn2 = (uint32_t)__popcll(mn2), cn3 = (uint32_t)__popcll(mn3);
                    const uint32_t cl0 = (uint32_t)__popcll(ml0), cl1 = (uint32_t)__popcll(ml1), cl2 = (uint32_t)__popcll(ml2), cl3 = (uint32_t)__popcll(ml3);
                    const uint32_t totn = (cn0 + cn1) + (cn2 + cn3), totl = (cl0 + cl1) + (cl2 + cl3);
                    if (nq + totn <= (uint32_t)K1_NQ && lq + totl <= (uint32_t)K1_LQ) {
                        // everything fits in LDS (the common case)
#if K1_SELECT_STORES
                        // One unpredicated store per slot: the lane's queue index is selected (node queue, leaf queue -- which
                        // follows it in LDS -- or a trash word for a child that was not hit).  Eight `if (hit) queue[..] = item`
                        // compile to exec-mask save / branch / restore sequences, ~60 scalar instructions per step on the CU's
                        // one scalar unit (VKR_EXP_EXTRA_SALU: +0.2 % of the launch per scalar instruction per step).
#define VKR_APPEND(H, L, MN, ML, PREN, PREL, R)                                                                     \
    {                                                                                                               \
        const uint32_t in_ = (nq + (PREN)) + mbcnt64(MN), il_ = ((uint32_t)K1_NQ + lq + (PREL)) + mbcnt64(ML);       \
        /* item: ray | node index, or ray | first * 2 | count - 1 -- a leaf reference is that plus bit 31 */          \
        s_nq[VKR_BOUND((H) ? ((L) ? il_ : in_) : K1_TRASH, K1_TRASH + 1u, 1u << 17)] = rayhi | ((R) & 0x7fffffffu);   \
    }
#if K1_SLOT0_FIRST       // slot 0 on top of the LIFO queues (opened first) instead of slot 3: -1 % (2.224 -> 2.200 ms, same box)
                        VKR_APPEND(h3, l3, mn3, ml3, 0u, 0u, r3)
                        VKR_APPEND(h2, l2, mn2, ml2, cn3, cl3, r2)
                        VKR_APPEND(h1, l1, mn1, ml1, cn3 + cn2, cl3 + cl2, r1)
                        VKR_APPEND(h0, l0, mn0, ml0, totn - cn0, totl - cl0, r0)
#else
                        VKR_APPEND(h0, l0, mn0, ml0, 0u, 0u, r0)
                        VKR_APPEND(h1, l1, mn1, ml1, cn0, cl0, r1)
                        VKR_APPEND(h2, l2, mn2, ml2, cn0 + cn1, cl0 + cl1, r2)
                        VKR_APPEND(h3, l3, mn3, ml3, totn - cn3, totl - cl3, r3)
#endif
#undef VKR_APPEND
#else
                        if (h0 && !l0) s_nq[VKR_BOUND(nq + mbcnt64(mn0), K1_NQ, 1u << 17)] = rayhi | r0;
                        if (h0 && l0) s_lq[lq + mbcnt64(ml0)] = rayhi_leaf + r0;
                        if (h1 && !l1) s_nq[nq + cn0 + mbcnt64(mn1)] = rayhi | r1;
                        if (h1 && l1) s_lq[lq + cl0 + mbcnt64(ml1)] = rayhi_leaf + r1;
                        if (h2 && !l2) s_nq[nq + (cn0 + cn1) + mbcnt64(mn2)] = rayhi | r2;
                        if (h2 && l2) s_lq[lq + (cl0 + cl1) + mbcnt64(ml2)] = rayhi_leaf + r2;
                        if (h3 && !l3) s_nq[VKR_BOUND(nq + (totn - cn3) + mbcnt64(mn3), K1_NQ, 1u << 18)] = rayhi | r3;
                        if (h3 && l3) s_lq[VKR_BOUND(lq + (totl - cl3) + mbcnt64(ml3), K1_LQ, 1u << 19)] = rayhi_leaf + r3;
#endif
                        nq += totn; lq += totl;
                    } else {
                        // rare: per slot, LDS while it fits, the wave's global stack beyond
#define VKR_PUSH(SQ, QN, CAP, GQ, GN, PRED, MASK, PC, ITEM)                                                         \
    {                                                                                                               \
        if (QN + (PC) <= (uint32_t)(CAP)) { if (PRED) SQ[QN + mbcnt64(MASK)] = (ITEM); QN += (PC); }                \
        else { if (PRED) GQ[VKR_BOUND(GN + mbcnt64(MASK), (GQ == g_nq ? K1_GNQ : K1_GLQ), 1u << 20)] = (ITEM); GN += (PC); if (STATS) nspill += (lane == 0) ? (PC) : 0u; } \
    }
                        VKR_PUSH(s_nq, nq, K1_NQ, g_nq, gnq, h0 && !l0, mn0, cn0, rayhi | r0)
                        VKR_PUSH(s_lq, lq, K1_LQ, g_lq, glq, h0 && l0, ml0, cl0, rayhi_leaf + r0)
                        VKR_PUSH(s_nq, nq, K1_NQ, g_nq, gnq, h1 && !l1, mn1, cn1, rayhi | r1)
                        VKR_PUSH(s_lq, lq, K1_LQ, g_lq, glq, h1 && l1, ml1, cl1, rayhi_leaf + r1)
                        VKR_PUSH(s_nq, nq, K1_NQ, g_nq, gnq, h2 && !l2, mn2, cn2, rayhi | r2)
                        VKR_PUSH(s_lq, lq, K1_LQ, g_lq, glq, h2 && l2, ml2, cl2, rayhi_leaf + r2)
                        VKR_PUSH(s_nq, nq, K1_NQ, g_nq, gnq, h3 && !l3, mn3, cn3, rayhi | r3)
                        VKR_PUSH(s_lq, lq, K1_LQ, g_lq, glq, h3 && l3, ml3, cl3, rayhi_leaf + r3)
#undef VKR_PUSH
                    }
                };
                const bool act0 = (uint32_t)lane < n, act1 = (uint32_t)lane + 64u < n;
                VKR_PRIO_HI
                const uint32_t it0 = act0 ? s_nq[VKR_BOUND(nq + (uint32_t)lane, K1_NQ, 65536)] : 0u;
                const NodeIn in0 = fetch(it0);
                VKR_PRIO_LO
#ifdef VKR_EXP_RUNS   // experiment (STATS build): how coherent a node step's items are -- distinct 128-byte node lines and distinct nodes
                      // among the step's items, reported in the cut_box_tests / cut_hits slots of the phase dump
                if (STATS) {
                    const uint32_t nd = it0 & PMASK;
                    uint32_t lines = 0, nodes = 0;
                    for (uint32_t i_ = 0; i_ < n && i_ < 64u; ++i_) {
                        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)nd, (int)i_);
                        const unsigned long long ml = __ballot(act0 && (nd >> 1) == (v >> 1)), mn_ = __ballot(act0 && nd == v);
                        lines += ((uint32_t)__ffsll((long long)ml) - 1u == i_) ? 1u : 0u;
                        nodes += ((uint32_t)__ffsll((long long)mn_) - 1u == i_) ? 1u : 0u;
                    }
                    if (lane == 0) { ncuttest += lines; ncuthit += nodes; }
                }
#endif
                if (K1_STEP > 64u && n > 64u) {                                     // wave-uniform
                    const uint32_t it1 = act1 ? s_nq[nq + 64u + (uint32_t)lane] : 0u;
                    const NodeIn in1 = fetch(it1);
                    process(in0, act0);
                    process(in1, act1);
                } else {
                    process(in0, act0);
                }
                if (STATS) { ph_n[0]++; ph_n[3] += n; }
                VKR_STAMP(2)
            }
            if (!K1_EARLY_RETRY && round == 0) {
                // rays that ended on a self-hit are traced once more from an advanced origin (:193-200): their LDS records
                // are rewritten in place (the values every step of round 1 would otherwise recompute)
                for (int i = 0; i < rayCount; ++i) {
                    bool need = false;
                    const uint32_t k = (uint32_t)i * nvalid + rank;
                    float4 a = make_float4(0, 0, 0, 0);
                    if (owner) {
                        a = s_ra[k];
                        const float tbv = s_rb[k].w;
                        need = (tbv < radius) && (tbv <= a.w * 1.5f);
                    }
                    const unsigned long long m = __ballot(need);
                    if (need) {
                        retry_list[VKR_BOUND(nretry + mbcnt64(m), RAYS, 1u << 21)] = k;
                        const V3 d = ray_dir(k);
                        const float adv = gmax(a.w * 2.0f, 2e-4f);                  // :194-196
                        const V3 ro = V3{a.x, a.y, a.z} + V3{d.x, d.y, d.z} * adv;
                        s_ra[k] = make_float4(ro.x, ro.y, ro.z, gmax(a.w, 1e-4f));
                        s_rb[k].w = radius;
                        retry_mask |= 1u << i;
                    }
                    nretry += (uint32_t)__popcll(m);
                }
                if (nretry) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
            }
        }
        VKR_STAMP(5)

        // ------------------------------------------------------------------ phase 3: per pixel resolve + temporal
        if (owner) {
            float visibility = 0.0f;
            uint32_t mask = 0;
            for (int i = 0; i < rayCount; ++i) {                                    // :431 in ray order
                const uint32_t k = (uint32_t)i * nvalid + rank;
                const float tbv = s_rb[k].w;
#if K1_EARLY_RETRY
                // tMin as generated: a re-trace stores max(tMin, 1e-4) (:197), and tMin >= 1e-4 already (:377,427-429), so the LDS
                // record still holds it -- no load from the wave's global slice in this loop
                const float tmin0 = s_ra[k].w;
#else
                const float tmin0 = g_dir[k].w;
#endif
                const float tmax = radius;
                float vis = 1.0f;
                if (tbv < tmax) {                                                   // a hit was committed
#if K1_EARLY_RETRY
                    const bool retried = (s_retry[k >> 5] >> (k & 31u)) & 1u;
#else
                    const bool retried = (retry_mask >> i) & 1u;
#endif
                    const float adv = gmax(tmin0 * 2.0f, 2e-4f);
                    const float t = retried ? (tbv + adv) : tbv;
                    if (!(t <= tmin0 * 1.5f)) {                                     // :205-212
                        float x = gclamp((t - tmin0) / gmax(tmax - tmin0, 1e-3f), 0.0f, 1.0f);
                        vis = pow3(x);
                        mask |= 1u << i;
                    }
                }
                visibility += vis;
            }
            VKR_STAMP(9)
#if K1_SPLIT
            visOut[(size_t)py * W + px] = make_float2(visibility, u2f(mask));      // k_pix_resolve does :433-471
#else
            PixelCtx pc2;                     // resolve_pixel reads viewZ and worldPos only
            pc2.viewZ = viewZ; pc2.worldPos = worldPos;
            resolve_pixel(fp, depth, histPrev, histCurr, hitMask, px, py, pc2, visibility, mask);
#endif
        }
        VKR_STAMP(6)
    }
#undef VKR_STAMP
    if (STATS) {
        if (lane == 0) {
            for (int k = 0; k < 8; ++k) atomicAdd(&stats[8 + k], (unsigned long long)ph[k]);
            atomicAdd(&stats[21], (unsigned long long)ph[8]);
            atomicAdd(&stats[22], (unsigned long long)ph[9]);
            for (int k = 0; k < 4; ++k) atomicAdd(&stats[16 + k], (unsigned long long)ph_n[k]);
            atomicAdd(&stats[20], 1ull);
            atomicAdd(&stats[5], (unsigned long long)nspill);
        }
        atomicAdd(&stats[0], (unsigned long long)nrays);
        atomicAdd(&stats[1], (unsigned long long)cnt.nodes);
        atomicAdd(&stats[2], (unsigned long long)cnt.tris);
        atomicAdd(&stats[3], (unsigned long long)cnt.wave_node_iters);
        atomicAdd(&stats[4], (unsigned long long)cnt.wave_tri_iters);
        atomicAdd(&stats[6], (unsigned long long)ncuttest);
        atomicAdd(&stats[7], (unsigned long long)ncuthit);
    }
}

// ray bits of the streaming kernel for this tree: K1_RB = 7 (128 rays per item) while triangle and node indices fit the item
// payload of 32 - RB bits, 6 for larger trees; 0 = the tree needs the one-ray-per-lane kernel
static int q_ray_bits(const vkr_ctx* c)
{
    for (int rb = K1_RB; rb >= 6; --rb)
        if ((uint64_t)c->tri_count * 2u <= (1ull << (32 - rb)) && (uint64_t)c->wide_count <= (1ull << (32 - rb))) {
            if (c->q_ray_bits_override >= 6 && c->q_ray_bits_override < rb) return c->q_ray_bits_override;   // VKR_Q_RAY_BITS (tests: a narrower item)
            return rb;
        }
    return 0;
}

template <bool STATS, int RB>
static hipError_t trace_q_launch(vkr_ctx* c, const SceneDev& sc, const FrameParams& fp, const uint16_t* depth, const uint16_t* hist_prev,
                                 uint16_t* hist_curr, uint32_t* hit_mask, int y0, int y1, uint32_t tiles_x, uint32_t tiles_y, hipStream_t st)
{
    hipError_t e;
    const uint32_t n_items = tiles_x * tiles_y * (64u / (uint32_t)q_item_pixels(RB, fp.rayCount));
    // persistent waves: as many blocks as the GPU holds at once, each with its own scratch slice (grow-only)
    const size_t lds = q_lds_bytes(RB) * K1_WG + (K1_WG > 1 ? (4u + 4u * K1_RING) * 4u : 0u) + (size_t)c->k1_extra_lds;   // VKR_K1_EXTRA_LDS: occupancy experiments (development)
    // resident workgroups per CU: asked of the runtime once per kernel instance and context, not per frame (a frame makes no
    // host query, allocation or synchronisation once its buffers exist: vkr_rtao_execute can be captured into a HIP graph)
    int& per_cu = c->k1_per_cu[STATS ? 1 : 0][RB - 6];
    if (per_cu == 0) {
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_q<STATS, RB>, K1Q_BLOCK, lds)) != hipSuccess) { per_cu = 0; return e; }
        if (per_cu < 1) per_cu = 1;
    }
    uint32_t blocks = (uint32_t)per_cu * (uint32_t)(c->num_cus > 0 ? c->num_cus : 256);
    if (blocks > (n_items + K1_WG - 1u) / K1_WG) blocks = (n_items + K1_WG - 1u) / K1_WG;
    const size_t need = (size_t)blocks * K1_WG * q_slice_layout(RB).total_f4;
    if (c->d_raybuf.n < need) {
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        if ((e = c->d_raybuf.alloc(need)) != hipSuccess) return e;
    }
    constexpr uint32_t n_ctr = 16u * (K1_SLABS > 8 ? K1_SLABS : 8);                                              // counters 64 B apart
    if (c->d_tile_counter.n < n_ctr && (e = c->d_tile_counter.alloc(n_ctr)) != hipSuccess) return e;
    static_assert(n_ctr <= 64u * PIX_ROWS, "k_pix_setup's first block clears the counters");
#if !K1_SPLIT
    if ((e = hipMemsetAsync(c->d_tile_counter.p, 0, n_ctr * 4u, st)) != hipSuccess) return e;
#endif
#if K1_SPLIT
    const size_t npx = (size_t)fp.W * fp.H;
    if (c->d_pixctx.n < 3u * npx || c->d_vis.n < npx) {
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        if ((e = c->d_pixctx.alloc(3u * npx)) != hipSuccess || (e = c->d_vis.alloc(npx)) != hipSuccess) return e;
    }
    const int rows = (int)tiles_y * 8;
    const dim3 pgrid(((uint32_t)fp.W + 63u) / 64u, ((uint32_t)rows + 3u) / 4u);
#if K1_CUT
    static_assert(K1_SPLIT == 1 && K1_EARLY_RETRY == 1, "the entry cut reads k_pix_setup's tile bounds and leaves re-traces to the root");
    const size_t ncut4 = (size_t)tiles_x * tiles_y * (K1_CUT_MAX + 1), nb4 = (size_t)tiles_x * tiles_y * 2u;
    if (c->d_cuts.n < ncut4 || c->d_tile_bounds.n < nb4) {
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        if ((e = c->d_cuts.alloc(ncut4)) != hipSuccess || (e = c->d_tile_bounds.alloc(nb4)) != hipSuccess) return e;
    }
#endif
    const uint32_t cut_max = c->cut_max < (uint32_t)K1_CUT_MAX ? c->cut_max : (uint32_t)K1_CUT_MAX;
    const bool fused = K1_CUT && c->cut_fused;
    k_pix_setup<<<dim3(pgrid.x, ((uint32_t)rows + PIX_ROWS - 1u) / PIX_ROWS), 64 * PIX_ROWS, 0, st>>>(
        fp, depth, hist_curr, hit_mask, y0, y1, rows, c->d_pixctx.p, c->d_tile_bounds.p, tiles_x, c->d_tile_counter.p, n_ctr,
        sc, fused ? c->d_cuts.p : nullptr, cut_max, c->cut_small, c->cut_l2_begin, c->cut_l2_count);
#if K1_CUT
    if (!fused) {
        // third level of the wide tree (levels are contiguous index ranges, recorded by the collapse): k_tile_cut's first round
        const uint32_t l2_begin = c->cut_l2_begin, l2_count = c->cut_l2_count;    // 0 nodes: descend from the root
        k_tile_cut<<<tiles_x * tiles_y, 64, 0, st>>>(sc, tiles_x * tiles_y, c->d_tile_bounds.p, c->d_cuts.p, cut_max, c->cut_small, l2_begin, l2_count);
    }
#endif
#endif
    k_trace_q<STATS, RB><<<blocks, K1Q_BLOCK, lds, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1, tiles_x, n_items,
                                                        c->d_raybuf.p, c->d_tile_counter.p,
#ifdef VKR_DEBUG_BOUNDS
                                                        c->d_stats.p
#else
                                                        STATS ? c->d_stats.p : nullptr
#endif
                                                        , c->d_pixctx.p, c->d_vis.p, c->d_cuts.p);
#if K1_SPLIT
    k_pix_resolve<<<pgrid, 256, 0, st>>>(fp, depth, hist_prev, hist_curr, hit_mask, y0, y1, rows, c->d_pixctx.p, c->d_vis.p);
#endif
    return hipGetLastError();
}

// What a frame would otherwise ask of the runtime on its first launch -- the occupancy of the persistent kernel and its
// scratch for a full-height launch -- done once per build (vkr_bvh_build), so that the frame calls only enqueue work.  The
// grow-only checks in trace_q_launch stay as the safety net (a STATS launch, a tree that changed its item width).
hipError_t rtao_prepare(vkr_ctx* c, hipStream_t st)
{
    const int rb = c->trace_streaming ? q_ray_bits(c) : 0;
    if (rb == 0) return hipSuccess;
    hipError_t e;
    const size_t lds = q_lds_bytes(rb) * K1_WG + (K1_WG > 1 ? (4u + 4u * K1_RING) * 4u : 0u) + (size_t)c->k1_extra_lds;
    int& per_cu = c->k1_per_cu[0][rb - 6];
    if (per_cu == 0) {
#if K1_RB >= 8
        if (rb == 8) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_q<false, 8>, K1Q_BLOCK, lds); else
#endif
#if K1_RB >= 7
        if (rb == 7) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_q<false, 7>, K1Q_BLOCK, lds); else
#endif
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_q<false, 6>, K1Q_BLOCK, lds);
        if (e != hipSuccess) { per_cu = 0; return e; }
        if (per_cu < 1) per_cu = 1;
    }
    const size_t blocks = (size_t)per_cu * (size_t)(c->num_cus > 0 ? c->num_cus : 256);
    const size_t need = blocks * K1_WG * q_slice_layout(rb).total_f4;
    const size_t npx = (size_t)c->W * c->H;
    const size_t tiles = (size_t)((c->W + 7u) / 8u) * ((c->H + 7u) / 8u + 1u);     // a band that starts between tile rows has one row more
    constexpr uint32_t n_ctr = 16u * (K1_SLABS > 8 ? K1_SLABS : 8);
    const bool grow = c->d_raybuf.n < need || c->d_tile_counter.n < n_ctr || (K1_SPLIT && (c->d_pixctx.n < 3u * npx || c->d_vis.n < npx)) ||
                      (K1_CUT && (c->d_cuts.n < tiles * (K1_CUT_MAX + 1) || c->d_tile_bounds.n < tiles * 2u));
    if (!grow) return hipSuccess;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    if (c->d_raybuf.n < need && (e = c->d_raybuf.alloc(need)) != hipSuccess) return e;
    if (c->d_tile_counter.n < n_ctr && (e = c->d_tile_counter.alloc(n_ctr)) != hipSuccess) return e;
#if K1_SPLIT
    if (c->d_pixctx.n < 3u * npx && (e = c->d_pixctx.alloc(3u * npx)) != hipSuccess) return e;
    if (c->d_vis.n < npx && (e = c->d_vis.alloc(npx)) != hipSuccess) return e;
#endif
#if K1_CUT
    if (c->d_cuts.n < tiles * (K1_CUT_MAX + 1) && (e = c->d_cuts.alloc(tiles * (K1_CUT_MAX + 1))) != hipSuccess) return e;
    if (c->d_tile_bounds.n < tiles * 2u && (e = c->d_tile_bounds.alloc(tiles * 2u)) != hipSuccess) return e;
#endif
    return hipSuccess;
}

hipError_t rtao_trace_launch(vkr_ctx* c, const FrameParams& fp, const uint16_t* depth, const uint16_t* hist_prev,
                             uint16_t* hist_curr, uint32_t* hit_mask, int y0, int y1, hipStream_t st)
{
    if (y1 <= y0) return hipSuccess;
    const uint32_t mul = fp.stripeMul > 1 ? (uint32_t)fp.stripeMul : 1u;
    uint32_t tiles_x = ((uint32_t)fp.W + 7u) / 8u, tiles_y = (((uint32_t)(y1 - y0) + 7u) / 8u + mul - 1u) / mul;
    uint32_t n_tiles = tiles_x * tiles_y;
    const int rb = c->trace_streaming ? q_ray_bits(c) : 0;
    SceneDev sc = scene_dev(c);
    hipError_t e;
    if (rb != 0) {
        const bool s = c->stats_enabled;
#define VKR_Q(S, R) trace_q_launch<S, R>(c, sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1, tiles_x, tiles_y, st)
#if K1_RB >= 8
        if (rb == 8) return s ? VKR_Q(true, 8) : VKR_Q(false, 8);
#endif
#if K1_RB >= 7
        if (rb == 7) return s ? VKR_Q(true, 7) : VKR_Q(false, 7);
#endif
        return s ? VKR_Q(true, 6) : VKR_Q(false, 6);
#undef VKR_Q
    }
    // One ray per lane (VKR_TRACE_MODE=lane, or a tree too large for the queue items).  Ray buffer: one slice per wave of a
    // launch (grow-only, so steady-state frames allocate nothing); the frame is cut into launches of at most
    // K1_LANE_SCRATCH_BYTES of slices, so the scratch stays bounded at any resolution (one launch up to ~1080p at 8 rays).
    const size_t slice_f4 = (size_t)fp.rayCount * 64u * 2u + (size_t)fp.rayCount * 16u;
    const size_t K1_LANE_SCRATCH_BYTES = 256u << 20;
    uint32_t per_launch = (uint32_t)(K1_LANE_SCRATCH_BYTES / (slice_f4 * sizeof(float4)));
    if (per_launch < 1u) per_launch = 1u;
    if (per_launch > n_tiles) per_launch = n_tiles;
    const size_t need = (size_t)per_launch * slice_f4;
    if (c->d_raybuf.n < need) {
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        if ((e = c->d_raybuf.alloc(need)) != hipSuccess) return e;
    }
    for (uint32_t base = 0; base < n_tiles; base += per_launch) {
        const uint32_t nb = (n_tiles - base < per_launch) ? n_tiles - base : per_launch;
        if (c->stats_enabled)
            k_trace<true><<<nb, K1_BLOCK, 0, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1, tiles_x, base,
                                                  c->d_raybuf.p, c->d_stats.p);
        else
            k_trace<false><<<nb, K1_BLOCK, 0, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1, tiles_x, base,
                                                   c->d_raybuf.p, nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
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

// The frame's A-Trous operands.  The tiled kernel (the default) decodes them itself while it stages a tile, so nothing is launched
// here; the gather form reads the float4 plane, which is written on its first use in the frame (rtao_atrous_launch).
hipError_t rtao_prep_launch(vkr_ctx* c, const uint16_t* depth, const uint16_t* normal, int y0, int y1, hipStream_t st)
{
    (void)st;
    c->at_depth = depth; c->at_normal = normal;
    c->at_y0 = y0; c->at_y1 = y1;
    c->at_nz_valid = false;
    return hipSuccess;
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
    // GLSL `/` may be a reciprocal multiply (2.5 ULP): the contract takes that form for the three per-tap quotients,
    // with correctly rounded reciprocals -- 25 x 3 IEEE divisions were a third of this kernel
    const float invZden = 1.0f / gmax(centerZ, 1e-3f);
    const float invSigmaAbs = 1.0f / sigmaAbs, invSigmaRel = 1.0f / sigmaRel;
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
            float dzRel = dz * invZden;                                         // :99
            float depthNorm = gmax(dz * invSigmaAbs, dzRel * invSigmaRel);      // :103
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

// K2, tiled form (the default).  The 25 taps of a pixel sit at multiples of `step`, so the pixels with equal
// (x mod step, y mod step) form a sub-image on which the filter is a dense 5x5 stencil.  One workgroup filters a 32x8
// tile of one sub-image: its 36x12 operand texels {AO as float, normal.xyz, depth} are fetched ONCE into LDS (each
// texel is the one the tap's clamp-to-edge address names, so edges need no special case), and the 25 taps then read LDS
// at immediate offsets -- no per-tap address arithmetic, no per-tap fp16 conversion, one global fetch per texel
// instead of 25 (steps 1..16 alike).  Arithmetic per tap is the shader's (:90-110), in the same order as k_atrous;
// exp's final scaling by 2^k is v_ldexp_f32, which rounds exactly like the multiplication by the float 2^k.
// TOL (build/ env switch, NOT the default): range and depth weights through v_exp_f32 -- outside the bit-exact contract,
// measured against the oracle in DESIGN.md.
#ifndef AT_ROWS
#define AT_ROWS 8         // rows of a tile: 8 (one pixel per thread, 36x12 staged texels for 256 outputs: 1.69x) or 16 (two per thread, 1.41x: measured slower, 0.285 vs 0.273 ms)
#endif
constexpr int AT_TX = 32, AT_TY = AT_ROWS, AT_LW = AT_TX + 4, AT_LH = AT_TY + 4;

VKR_DI float det_exp_ld(float x)   // det_exp with v_ldexp_f32 for the last step (same bits: both forms round once)
{
    float k = __builtin_rintf(x * 1.44269504088896341f);
    float r = fma_(k, -0.693359375f, x);
    r = fma_(k, 2.12194440e-4f, r);
    float p = fma_(r, 1.9875691500E-4f, 1.3981999507E-3f);
    p = fma_(r, p, 8.3334519073E-3f);
    p = fma_(r, p, 4.1665795894E-2f);
    p = fma_(r, p, 1.6666665459E-1f);
    p = fma_(r, p, 5.0000001201E-1f);
    p = fma_(r * r, p, r) + 1.0f;
    float res = __builtin_ldexpf(p, (int)k);
    return (x < -87.0f) ? 0.0f : res;
}

template <bool TOL>
__global__ __launch_bounds__(256) void k_atrous_tile(const uint16_t* src, const uint16_t* depth, const uint16_t* normal, uint16_t* dst, uint16_t* dst2,
                                                     int W, int H, int y0, int y1, int step, float sigmaAbs,
                                                     uint32_t tiles_x, uint32_t tiles_y, uint32_t per_xcd)
{
    __shared__ float4 s_nz[AT_LH][AT_LW];
    __shared__ float s_ao[AT_LH][AT_LW];
    // Workgroups go to the 8 XCDs round robin by their linear id, and each XCD has its own L2.  The `step` sub-images with
    // equal row offset read the same 128-byte lines (a line holds 16 neighbouring normals, a sub-image uses every step-th of
    // them), so work item w = (x offset fastest, then tile, then row offset) is dealt out in contiguous runs per XCD: the
    // sub-images that share lines run on one XCD at about the same time and the line comes from HBM once, not `step` times.
    const uint32_t w = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (w >= tiles_x * tiles_y * (uint32_t)(step * step)) return;                  // the whole workgroup: before any barrier
    const int sa = (int)(w % (uint32_t)step);                                      // sub-image: x mod step ...
    uint32_t t_ = w / (uint32_t)step;
    const int i0 = (int)(t_ % tiles_x) * AT_TX;                                    // tile origin in sub-image coordinates
    t_ /= tiles_x;
    const int j0 = (int)(t_ % tiles_y) * AT_TY, sb = (int)(t_ / tiles_y);          // ... and row offset in the band
    for (int t = (int)threadIdx.x; t < AT_LW * AT_LH; t += 256) {
        const int ly = t / AT_LW, lx = t - ly * AT_LW;
        const int gx = iclamp(sa + step * (i0 + lx - 2), 0, W - 1);                // = clamp(px + x*step) of :81 for every tap that reads it
        const int gy = iclamp(y0 + sb + step * (j0 + ly - 2), 0, H - 1);
        const size_t to = (size_t)gy * W + gx;
        s_ao[ly][lx] = half_to_float(src[to]);                                      // :82-89
        // decodeNormalSafe (:38-46) + max(depth, 1e-3) (:70-71,92-93) once per staged texel, straight from the R16F / RGBA16F
        // inputs (12 B per texel instead of an 18-B pre-decoded operand plane and the launch that wrote it); k_prep's arithmetic
        const ushort4 e = ((const ushort4*)normal)[to];
        const V3 n = V3{half_to_float(e.x) * 2.0f - 1.0f, half_to_float(e.y) * 2.0f - 1.0f, half_to_float(e.z) * 2.0f - 1.0f};
        const float l2 = dot3(n, n);
        // 1 / sqrt: rcp_ieee is the correctly rounded quotient where the select below keeps it (l2 >= 1e-6), traverse.hpp
        const V3 r = (n.x != n.x || n.y != n.y || n.z != n.z || l2 < 1e-6f) ? v3(0, 1, 0) : n * rcp_ieee(sqrt_(l2));
        s_nz[ly][lx] = make_float4(r.x, r.y, r.z, gmax(half_to_float(depth[to]), 1e-3f));   // :92-94
    }
    __syncthreads();
    for (int part = 0; part < AT_TY / 8; ++part) {                                  // 256 threads, AT_TY / 8 pixels each
    const int tx = (int)threadIdx.x & (AT_TX - 1), ty = ((int)threadIdx.x >> 5) + 8 * part;
    const int px = sa + step * (i0 + tx), py = y0 + sb + step * (j0 + ty);
    if (px >= W || py >= y1) continue;
    const float center = s_ao[ty + 2][tx + 2];                                      // :62-69
    const float4 cn = s_nz[ty + 2][tx + 2];                                         // :70-72
    const float centerZ = cn.w;
    const V3 centerN = V3{cn.x, cn.y, cn.z};
    float sum = 0.0f, weightSum = 0.0f;
    const float sigmaRel = 0.0025f;                                                 // :102
    const float stepScale = 1.0f + 0.25f * (float)step;                             // :104
    const float invZden = 1.0f / gmax(centerZ, 1e-3f);                              // reciprocal form of the three quotients: DESIGN.md 2
    const float invSigmaAbs = 1.0f / sigmaAbs, invSigmaRel = 1.0f / sigmaRel;
#pragma unroll
    for (int y = -2; y <= 2; ++y) {                                                 // :78-112
#pragma unroll
        for (int x = -2; x <= 2; ++x) {
            const float s = s_ao[ty + 2 + y][tx + 2 + x];
            const float4 tn = s_nz[ty + 2 + y][tx + 2 + x];
            const float spatial = det_exp(-0.5f * (float)(x * x + y * y));          // :90 (a compile-time constant per tap)
            const float rarg = -__builtin_fabsf(s - center) * 8.0f;
            const float range = TOL ? __builtin_amdgcn_exp2f(rarg * 1.44269504088896341f) : det_exp_ld(rarg);   // :91
            const float dz = __builtin_fabsf(tn.w - centerZ);                       // :98
            const float dzRel = dz * invZden;                                       // :99
            float depthNorm = gmax(dz * invSigmaAbs, dzRel * invSigmaRel);          // :103
            depthNorm *= stepScale;                                                 // :104
            const float darg = -depthNorm * depthNorm;
            const float depthWeight = TOL ? __builtin_amdgcn_exp2f(darg * 1.44269504088896341f) : det_exp_ld(darg);   // :105
            // dot of two unit (or (0,1,0)) vectors is never NaN: the median form equals clamp(x, 0, 1) up to the sign of a zero,
            // which the squarings below remove
            const float nd = __builtin_amdgcn_fmed3f(dot3(centerN, V3{tn.x, tn.y, tn.z}), 0.0f, 1.0f);
            const float normalWeight = pow32(nd);                                   // :107
            const float w = ((spatial * range) * depthWeight) * normalWeight;       // :108
            sum += s * w;                                                           // :109
            weightSum += w;                                                         // :110
        }
    }
    const size_t o = (size_t)py * W + px;
    const float filtered = (weightSum > 0.0f) ? (sum / weightSum) : center;         // :114
    const uint16_t r = float_to_half(filtered);
    dst[o] = r;                                                                     // :115-119
    if (dst2) dst2[o] = r;
    }
}

hipError_t rtao_atrous_launch(vkr_ctx* c, float radius_raw, int step, const uint16_t* src, uint16_t* dst,
                              uint16_t* dst2, int y0, int y1, hipStream_t st)
{
    if (y1 <= y0) return hipSuccess;
    float radius = radius_raw < 0.01f ? 0.01f : radius_raw;                     // :100 max(rtaoParams0.z, 0.01)
    if (!(radius_raw == radius_raw)) radius = radius_raw;                       // GLSL max(NaN, y) = NaN
    float sigmaAbs = (0.01f < radius * 0.05f) ? radius * 0.05f : 0.01f;         // :101
    if (step < 1) step = 1;                                                     // :76
    if (!c->at_depth || !c->at_normal) return hipErrorInvalidValue;             // rtao_prep_launch names the frame's operands first
    if (c->atrous_mode == 0 || step > 64) {                                     // VKR_ATROUS=gather: one thread per pixel, 25 global gathers
        if (!c->at_nz_valid) {                                                  // its pre-decoded operand plane, once per frame
            dim3 pg((c->W + 63u) / 64u, ((uint32_t)(c->at_y1 - c->at_y0) + 3u) / 4u);
            k_prep<<<pg, 256, 0, st>>>(c->at_depth, c->at_normal, c->d_nz.p, (int)c->W, c->at_y0, c->at_y1);
            c->at_nz_valid = true;
        }
        dim3 grid((c->W + 31u) / 32u, ((uint32_t)(y1 - y0) + 7u) / 8u);
        k_atrous<<<grid, 256, 0, st>>>(src, c->d_nz.p, dst, dst2, (int)c->W, (int)c->H, y0, y1, step, sigmaAbs);
        return hipGetLastError();
    }
    const uint32_t sw = ((uint32_t)c->W + (uint32_t)step - 1u) / (uint32_t)step;            // sub-image extent
    const uint32_t sh = ((uint32_t)(y1 - y0) + (uint32_t)step - 1u) / (uint32_t)step;
    const uint32_t tiles_x = (sw + AT_TX - 1u) / AT_TX, tiles_y = (sh + AT_TY - 1u) / AT_TY;
    const uint32_t per_xcd = (tiles_x * tiles_y * (uint32_t)(step * step) + 7u) / 8u, grid = per_xcd * 8u;
    if (c->atrous_mode == 2)
        k_atrous_tile<true><<<grid, 256, 0, st>>>(src, c->at_depth, c->at_normal, dst, dst2, (int)c->W, (int)c->H, y0, y1, step, sigmaAbs, tiles_x, tiles_y, per_xcd);
    else
        k_atrous_tile<false><<<grid, 256, 0, st>>>(src, c->at_depth, c->at_normal, dst, dst2, (int)c->W, (int)c->H, y0, y1, step, sigmaAbs, tiles_x, tiles_y, per_xcd);
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
