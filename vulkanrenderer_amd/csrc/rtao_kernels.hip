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
constexpr int CUR_DONE = (int)0x80000000;

constexpr int K1_BLOCK = 64;

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
                                                    int y0, int y1, uint32_t tiles_x, float4* rayBuf,
                                                    unsigned long long* stats)
{
    __shared__ uint32_t s_stack[2 * VKR_LDS_STACK * K1_BLOCK];
    __shared__ float s_radius[K1_BLOCK];
    uint32_t* stack = s_stack + threadIdx.x;
    constexpr int B = K1_BLOCK;

    const uint32_t tile = blockIdx.x;
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
        const uint32_t total = nvalid * (uint32_t)rayCount;
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
                    const uint32_t eb = f2u(q0.w);
                    const float ax = u2f((eb & 0xffu) << 23) * id.x, ay = u2f(((eb >> 8) & 0xffu) << 23) * id.y,
                                az = u2f(((eb >> 16) & 0xffu) << 23) * id.z;
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
                uint32_t x = (uint32_t)(~pend0);
                uint32_t first = x & 0x0fffffffu, rest = (x >> 28) & 7u;     // rest = triangles after this one
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
                if (rest != 0u) pend0 = (int)(~((first + 1u) | ((rest - 1u) << 28)));
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
// K1, streaming form (the default).  Persistent waves pull 8x8 tiles from a counter; inside a tile the rays do not
// own lanes: work lives in two wave-wide LIFO queues in LDS,
//     node items (wide node, ray)   and   leaf items (leaf, ray),
// and every step pops up to 64 items of ONE kind, so node steps and leaf steps both run with (nearly) all lanes
// busy -- the per-lane state machine above keeps 50-60 % of the lanes busy because a lane holds either node work
// or triangle work, rarely both.  A node step tests the four child boxes against [tmin, closest hit so far] and
// appends the hit children to the two queues (ballot + mbcnt compaction); a leaf step tests the leaf's triangles
// and lowers the ray's closest hit with an LDS atomic min on the float bits.  No ordering, no per-lane stack:
// the closest committed hit does not depend on the order triangles are met in, and AO rays mostly miss, so
// near-first ordering would buy little.  The self-hit retry (:193-200) is a second round over the rays that
// need it.  The node queue cannot overflow: a step pops n items and pushes at most 4n, so n shrinks as the queue
// fills, down to one item per step = strict depth-first, whose growth is bounded by 3 x tree depth (KQ_GUARD).
// ---------------------------------------------------------------------------------------------
struct QSliceLayout { uint32_t rays_f4, retry_f4, total_f4; };   // per-wave global scratch, in float4 units
__host__ __device__ inline QSliceLayout q_slice_layout(int rayCount)
{
    QSliceLayout l;
    l.rays_f4 = (uint32_t)rayCount * 64u * 3u;    // {origin | tMin}, {direction | -}, {1/direction | -} per ray
    l.retry_f4 = (uint32_t)rayCount * 16u;        // one u32 per ray
    l.total_f4 = l.rays_f4 + l.retry_f4;
    return l;
}

// Occupancy of k_trace_q is set by LDS and VGPRs together; measured best on MI355X at 8 rays/px: 640 node items
// (7.3 KB of LDS per wave -> 22 waves per CU) and a VGPR budget for 6 waves per SIMD.
#ifndef VKR_Q_WAVES
#define VKR_Q_WAVES 6
#endif
#ifndef K1_NQ
#define K1_NQ 640
#endif
// Pixels per work item of k_trace_q: 64 (8x8) or 32 (8x4; lanes 32..63 mirror the pixels of lanes 0..31 and generate
// the odd rays, lanes 0..31 resolve).  Smaller items shorten the drain of the persistent launch.
#ifndef K1_TILE_PX
#define K1_TILE_PX 32
#endif
template <bool STATS>
__global__ __launch_bounds__(K1_BLOCK) __attribute__((amdgpu_waves_per_eu(VKR_Q_WAVES, VKR_Q_WAVES))) void k_trace_q(SceneDev sc, FrameParams fp, const uint16_t* depthP,
                                                      const uint16_t* histPrevP, uint16_t* histCurr, uint32_t* hitMask,
                                                      int y0, int y1, uint32_t tiles_x, uint32_t n_tiles, float4* rayBuf,
                                                      uint32_t* tile_counter, unsigned long long* stats)
{
    extern __shared__ uint32_t smem[];
    const int rayCount = fp.rayCount;                                           // :252
    uint32_t* s_tbest = smem;                                                   // [rayCount * K1_TILE_PX] closest hit, float bits
    uint32_t* s_nq = smem + (uint32_t)rayCount * (uint32_t)K1_TILE_PX;                           // node items: ray << 22 | node
    uint32_t* s_lq = s_nq + K1_NQ;                                              // leaf items: leaf reference ...
    uint16_t* s_lqr = (uint16_t*)(s_lq + KQ_LQ);                                // ... and its ray
    uint32_t* s_aq = (uint32_t*)(s_lqr + KQ_LQ);                                // MASK candidates: triangle index ...
    uint16_t* s_aqr = (uint16_t*)(s_aq + KQ_AQ);                                // ... and its ray

    const int lane = threadIdx.x;
    const int W = fp.W, H = fp.H;
    const Img16 depth{depthP, W, H}, histPrev{histPrevP, W, H};
    const QSliceLayout lay = q_slice_layout(rayCount);
    float4* slice = rayBuf + (size_t)blockIdx.x * lay.total_f4;
    uint32_t* retry_list = (uint32_t*)(slice + lay.rays_f4);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const char* node_base = (const char*)sc.nodes;
    const char* tri_base = (const char*)sc.tris;
    TraceCounters cnt{0, 0, 0, 0};
    uint32_t nrays = 0;

    for (;;) {
        uint32_t tile = 0;
        if (lane == 0) tile = atomicAdd(tile_counter, 1u);
        tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
        if (tile >= n_tiles) break;          // every wave reaches this: the counter only grows
        constexpr bool HALF = (K1_TILE_PX == 32);
        const int plane = HALF ? (lane & 31) : lane;          // pixel of the item this lane works on
        const int half = HALF ? (lane >> 5) : 0;              // which rays it generates (i = half, half + 2, ...)
        const int trow = (int)(tile / tiles_x);
        const int px = (int)(tile % tiles_x) * 8 + (plane & 7);
        const int py = HALF ? y0 + (trow >> 1) * 8 * fp.stripeMul + (trow & 1) * 4 + (plane >> 3)
                            : y0 + trow * 8 * fp.stripeMul + (plane >> 3);

        // ------------------------------------------------------------------ phase 1: per pixel
        PixelCtx pc;
        pc.valid = false;
        pc.radius = 0.0f;
        if (px < W && py < y1) setup_pixel(fp, depth, histCurr, hitMask, px, py, pc);   // :233-235
        const bool valid = pc.valid;
        const float radius = pc.radius;
        const unsigned long long vmask = HALF ? (__ballot(valid) & 0xffffffffull) : __ballot(valid);
        const uint32_t nvalid = (uint32_t)__popcll(vmask);
        if (nvalid == 0) continue;   // wave-uniform: nothing to trace, nothing to resolve
        const uint32_t rank = (uint32_t)__popcll(vmask & (HALF ? ((1ull << plane) - 1ull) : lt_mask));
        const bool owner = valid && half == 0;                // the lane that resolves the pixel
        if (valid) {
            for (int i = half; i < rayCount; i += (HALF ? 2 : 1)) {                 // :403-430
                V3 rayOrigin, rayDir;
                float tMinRay;
                make_ray(pc, rayCount, i, &rayOrigin, &rayDir, &tMinRay);
                const uint32_t k = (uint32_t)i * nvalid + rank;
                slice[(size_t)k * 3u] = make_float4(rayOrigin.x, rayOrigin.y, rayOrigin.z, tMinRay);
                slice[(size_t)k * 3u + 1u] = make_float4(rayDir.x, rayDir.y, rayDir.z, 0.0f);
                slice[(size_t)k * 3u + 2u] = make_float4(safe_rcp(rayDir.x), safe_rcp(rayDir.y), safe_rcp(rayDir.z), 0.0f);
                s_tbest[k] = f2u(pc.radius);                                        // tMax = radius (:430)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        // ------------------------------------------------------------------ phase 2: queues
        const uint32_t total = nvalid * (uint32_t)rayCount;
        uint32_t retry_mask = 0;     // bit i: this pixel's ray i was re-traced
        uint32_t nretry = 0;
        for (int round = 0; round < 2; ++round) {
            const uint32_t set_size = (round == 0) ? total : nretry;
            if (set_size == 0) break;
            uint32_t next = 0, nq = 0, lq = 0, aq = 0;      // wave-uniform
            // ---------------- alpha step: MASK candidates (:163-173), n items from s_aq[base..]
            auto alpha_step = [&](uint32_t n, uint32_t base) {
                if ((uint32_t)lane < n) {
                    const uint32_t tri = s_aq[base + (uint32_t)lane], ray = (uint32_t)s_aqr[base + (uint32_t)lane];
                    const float4 a = slice[(size_t)ray * 3u], b = slice[(size_t)ray * 3u + 1u];
                    const float4* tp = (const float4*)(tri_base + (size_t)tri * 48u);
                    const float4 ta = tp[0], tb4 = tp[1], tc = tp[2];
                    V3 ro = V3{a.x, a.y, a.z}, rdir = V3{b.x, b.y, b.z};
                    float tmin = a.w;
                    if (round) {
                        const float adv = gmax(a.w * 2.0f, 2e-4f);
                        ro = ro + rdir * adv;
                        tmin = gmax(a.w, 1e-4f);
                    }
                    float u, v, t;
                    const float tbest = u2f(s_tbest[ray]);
                    if (tri_hit(ro, rdir, ta, tb4, tc, tmin, tbest, &u, &v, &t) &&
                        alpha_test(sc, f2u(ta.w) & 0x0fffffffu, f2u(tb4.w), u, v))
                        atomicMin(&s_tbest[ray], f2u(t));
                }
            };
            for (;;) {
                const bool no_more = (next >= set_size);
                if (aq > 0u && lq < 64u && nq == 0u && no_more) { alpha_step(aq, 0u); aq = 0u; continue; }
                if (lq >= 64u || (lq > 0u && nq == 0u && no_more)) {
                    // ---------------- leaf step: queryAoClosestHit's candidate loop (:125-182) for <= 64 leaves
                    const uint32_t n = lq < 64u ? lq : 64u;
                    lq -= n;
                    uint32_t cntT = 0;
                    uint2 it = make_uint2(0u, 0u);
                    if ((uint32_t)lane < n) it = make_uint2(s_lq[lq + (uint32_t)lane], (uint32_t)s_lqr[lq + (uint32_t)lane]);
                    float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 1, 0);
                    uint32_t first = 0;
                    float tbest0 = 0.0f;
                    if ((uint32_t)lane < n) {
                        a = slice[(size_t)it.y * 3u]; b = slice[(size_t)it.y * 3u + 1u];
                        const uint32_t x = ~it.x;
                        first = x & 0x0fffffffu; cntT = ((x >> 28) & 7u) + 1u;
                        tbest0 = u2f(s_tbest[it.y]);
                    }
                    V3 ro = V3{a.x, a.y, a.z}, rdir = V3{b.x, b.y, b.z};
                    float tmin = a.w;
                    if (round) {                                                    // :193-197
                        const float adv = gmax(a.w * 2.0f, 2e-4f);
                        ro = ro + rdir * adv;
                        tmin = gmax(a.w, 1e-4f);
                    }
                    float tbest = tbest0;
                    for (uint32_t j = 0; j < (uint32_t)VKR_LEAF_MAX; ++j) {
                        const unsigned long long am = __ballot(j < cntT);
                        if (am == 0ull) break;
                        bool cand = false;   // MASK candidate: geometric hit that still needs the alpha test
                        if (j < cntT) {
                            const float4* tp = (const float4*)(tri_base + (size_t)(first + j) * 48u);
                            float4 ta = tp[0], tb4 = tp[1], tc = tp[2];
                            if (STATS) { cnt.tris++; if ((int)(__ffsll((long long)am) - 1) == lane) cnt.wave_tri_iters++; }
                            float u, v, t;
                            const bool geo = tri_hit(ro, rdir, ta, tb4, tc, tmin, tbest, &u, &v, &t);
                            const uint32_t cls = f2u(ta.w) >> 28;                   // candidate filter :142-173
                            if (geo && cls == 0u) tbest = t;
                            cand = geo && cls == 2u;
                        }
                        // alpha tests are not run here at one or two lanes' width: candidates queue up and are tested
                        // a wave at a time (accepting a hit later only delays pruning; the minimum is the same)
                        const unsigned long long cm = __ballot(cand);
                        if (cm != 0ull) {
                            if (cand) { const uint32_t ix = aq + mbcnt64(cm); s_aq[ix] = first + j; s_aqr[ix] = (uint16_t)it.y; }
                            aq += (uint32_t)__popcll(cm);
                            if (aq >= 64u) { aq -= 64u; alpha_step(64u, aq); }
                        }
                    }
                    if (tbest < tbest0) atomicMin(&s_tbest[it.y], f2u(tbest));      // t > tmin >= 0: bits order like floats
                    continue;
                }
                if (nq < 64u && next < set_size) {
                    // ---------------- inject the next rays at the root
                    const uint32_t m = (set_size - next) < 64u ? (set_size - next) : 64u;
                    if ((uint32_t)lane < m) {
                        const uint32_t ray = round ? retry_list[next + (uint32_t)lane] : next + (uint32_t)lane;
                        s_nq[nq + (uint32_t)lane] = ray << KQ_NODE_BITS;   // root = node 0
                    }
                    nq += m; next += m;
                    if (STATS) { if ((uint32_t)lane < m) nrays++; }
                }
                if (nq == 0u) break;
                // ---------------- node step: pops n, pushes <= 4n node items (and <= 4n leaf items, lq < 64 here)
                uint32_t n = nq < 64u ? nq : 64u;
                {
                    const uint32_t room = (uint32_t)K1_NQ - KQ_GUARD - (nq < (uint32_t)K1_NQ - KQ_GUARD ? nq : (uint32_t)K1_NQ - KQ_GUARD);
                    const uint32_t fit = room / 3u;
                    if (n > fit) n = fit ? fit : 1u;
                }
                nq -= n;
                // Every lane runs the step; lanes without an item work on item 0 (ray 0 at the root: valid addresses, no
                // side effects) and are masked out of the hit tests.  Keeping the step in one basic block lets the hit /
                // leaf ballots come straight out of the compares (v_cmp -> SGPR pair) instead of being re-materialised
                // from booleans carried across a divergent region: -16 VALU instructions per step.
                const bool act = (uint32_t)lane < n;
                const uint32_t item = act ? s_nq[nq + (uint32_t)lane] : 0u;
                const uint32_t ray = item >> KQ_NODE_BITS;
                const uint32_t node = item & ((1u << KQ_NODE_BITS) - 1u);
                const float4 a = slice[(size_t)ray * 3u], ir = slice[(size_t)ray * 3u + 2u];
                const float4* np = (const float4*)(node_base + ((size_t)node << 6));
                const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                // padded: culling may only err towards extra visits; lanes without an item get an exit distance below any
                // entry distance (tn >= tmin > 0), so none of their four tests can pass
                const float tb = act ? u2f(s_tbest[ray]) * 1.000001f : -3.0e38f;
                if (STATS) { if (act) { cnt.nodes++; if ((int)(__ffsll((long long)__ballot(1)) - 1) == lane) cnt.wave_node_iters++; } }
                V3 ro = V3{a.x, a.y, a.z};
                float tmin = a.w;
                if (round) {
                    const float4 b = slice[(size_t)ray * 3u + 1u];
                    const float adv = gmax(a.w * 2.0f, 2e-4f);
                    ro = ro + V3{b.x, b.y, b.z} * adv;
                    tmin = gmax(a.w, 1e-4f);
                }
                const V3 id = V3{ir.x, ir.y, ir.z};
                // Slab tests straight on the quantised planes (see k_trace): t = fma(q, s*id, (o - ro)*id)
                const uint32_t eb = f2u(q0.w);
                const float ax = u2f((eb & 0xffu) << 23) * id.x, ay = u2f(((eb >> 8) & 0xffu) << 23) * id.y,
                            az = u2f(((eb >> 16) & 0xffu) << 23) * id.z;
                const float bx = (q0.x - ro.x) * id.x, by = (q0.y - ro.y) * id.y, bz = (q0.z - ro.z) * id.z;
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
                // a NaN entry distance fails the comparison, like the k != 0xffffffff test of the keyed form did
#define VKR_SLABQ(H, NX, NY, NZ, FX, FY, FZ)                                                                        \
    const bool H = __builtin_fmaxf(__builtin_fmaxf(NX, NY), __builtin_fmaxf(NZ, tmin)) <=                           \
                   __builtin_fminf(__builtin_fminf(FX, FY), __builtin_fminf(FZ, tb));
                VKR_SLABQ(h0, nx01.x, ny01.x, nz01.x, fx01.x, fy01.x, fz01.x)
                VKR_SLABQ(h1, nx01.y, ny01.y, nz01.y, fx01.y, fy01.y, fz01.y)
                VKR_SLABQ(h2, nx23.x, ny23.x, nz23.x, fx23.x, fy23.x, fz23.x)
                VKR_SLABQ(h3, nx23.y, ny23.y, nz23.y, fx23.y, fy23.y, fz23.y)
#undef VKR_SLABQ
                // append the hit children: internal nodes to the node queue, leaves to the leaf queue, slot by slot
                // (measured: sorting them near-to-far first saves 1 % of the visits and costs more than that; one
                // lane-major prefix sum instead of per-slot ballots makes a ray's siblings pop together and costs 9 %)
#define VKR_APPEND(HC, RC)                                                                                          \
    {                                                                                                               \
        const unsigned long long hm = __ballot(HC), lm = __ballot((int)RC < 0);                                     \
        const unsigned long long mn = hm & ~lm, ml = hm & lm;                                                       \
        if (HC) {                                                                                                   \
            if ((int)RC < 0) { const uint32_t ix = lq + mbcnt64(ml); s_lq[ix] = RC; s_lqr[ix] = (uint16_t)ray; }    \
            else s_nq[nq + mbcnt64(mn)] = (ray << KQ_NODE_BITS) | RC;                                               \
        }                                                                                                           \
        nq += (uint32_t)__popcll(mn); lq += (uint32_t)__popcll(ml);                                                 \
    }
                VKR_APPEND(h0, r0) VKR_APPEND(h1, r1) VKR_APPEND(h2, r2) VKR_APPEND(h3, r3)
#undef VKR_APPEND
            }
            if (round == 0) {
                // rays that ended on a self-hit are traced once more from an advanced origin (:193-200)
                for (int i = 0; i < rayCount; ++i) {
                    bool need = false;
                    const uint32_t k = (uint32_t)i * nvalid + rank;
                    if (owner) {
                        const float tbv = u2f(s_tbest[k]);
                        const float tmin0 = slice[(size_t)k * 3u].w;
                        need = (tbv < radius) && (tbv <= tmin0 * 1.5f);
                    }
                    const unsigned long long m = __ballot(need);
                    if (need) {
                        retry_list[nretry + mbcnt64(m)] = k;
                        s_tbest[k] = f2u(radius);
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

        // ------------------------------------------------------------------ phase 3: per pixel resolve + temporal
        if (owner) {
            float visibility = 0.0f;
            uint32_t mask = 0;
            for (int i = 0; i < rayCount; ++i) {                                    // :431 in ray order
                const uint32_t k = (uint32_t)i * nvalid + rank;
                const float tbv = u2f(s_tbest[k]);
                const float tmin0 = slice[(size_t)k * 3u].w;
                const float tmax = radius;
                float vis = 1.0f;
                if (tbv < tmax) {                                                   // a hit was committed
                    const bool retried = (retry_mask >> i) & 1u;
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
            // the pixel context is rebuilt rather than kept in registers across the queue loop (it is ~25 VGPRs, and
            // the loop's occupancy matters more than ~500 instructions per pixel): same inputs, same values
            PixelCtx pc2;
            setup_pixel(fp, depth, histCurr, hitMask, px, py, pc2);
            resolve_pixel(fp, depth, histPrev, histCurr, hitMask, px, py, pc2, visibility, mask);
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

hipError_t rtao_trace_launch(vkr_ctx* c, const FrameParams& fp, const uint16_t* depth, const uint16_t* hist_prev,
                             uint16_t* hist_curr, uint32_t* hit_mask, int y0, int y1, hipStream_t st)
{
    if (y1 <= y0) return hipSuccess;
    const uint32_t mul = fp.stripeMul > 1 ? (uint32_t)fp.stripeMul : 1u;
    uint32_t tiles_x = ((uint32_t)fp.W + 7u) / 8u, tiles_y = (((uint32_t)(y1 - y0) + 7u) / 8u + mul - 1u) / mul;
    uint32_t n_tiles = tiles_x * tiles_y;
    const bool streaming = c->trace_streaming && c->wide_count <= (1u << KQ_NODE_BITS);
    if (streaming && K1_TILE_PX == 32) n_tiles *= 2u;            // 8x4 items: two per 8x8 tile row (see k_trace_q)
    SceneDev sc = scene_dev(c);
    hipError_t e;
    if (streaming) {
        // persistent waves: as many blocks as the GPU holds at once, each with its own scratch slice (grow-only)
        const size_t lds = (size_t)fp.rayCount * (size_t)K1_TILE_PX * 4u + (size_t)K1_NQ * 4u + (size_t)KQ_LQ * 6u + (size_t)KQ_AQ * 6u;
        int per_cu = 0;
        if (c->stats_enabled) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_q<true>, K1_BLOCK, lds);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_q<false>, K1_BLOCK, lds);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        uint32_t blocks = (uint32_t)per_cu * (uint32_t)(c->num_cus > 0 ? c->num_cus : 256);
        if (blocks > n_tiles) blocks = n_tiles;
        const size_t need = (size_t)blocks * q_slice_layout(fp.rayCount).total_f4;
        if (c->d_raybuf.n < need) {
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
            if ((e = c->d_raybuf.alloc(need)) != hipSuccess) return e;
        }
        if (!c->d_tile_counter.p && (e = c->d_tile_counter.alloc(1)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(c->d_tile_counter.p, 0, 4, st)) != hipSuccess) return e;
        if (c->stats_enabled)
            k_trace_q<true><<<blocks, K1_BLOCK, lds, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1, tiles_x,
                                                           n_tiles, c->d_raybuf.p, c->d_tile_counter.p, c->d_stats.p);
        else
            k_trace_q<false><<<blocks, K1_BLOCK, lds, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1, tiles_x,
                                                            n_tiles, c->d_raybuf.p, c->d_tile_counter.p, nullptr);
        return hipGetLastError();
    }
    // ray buffer: one slice per wave of the launch, sized for the current rays/px (grow-only, so steady-state
    // frames allocate nothing)
    size_t need = (size_t)n_tiles * ((size_t)fp.rayCount * 64u * 2u + (size_t)fp.rayCount * 16u);
    if (c->d_raybuf.n < need) {
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        if ((e = c->d_raybuf.alloc(need)) != hipSuccess) return e;
    }
    if (c->stats_enabled)
        k_trace<true><<<n_tiles, K1_BLOCK, 0, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1, tiles_x,
                                                   c->d_raybuf.p, c->d_stats.p);
    else
        k_trace<false><<<n_tiles, K1_BLOCK, 0, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1, tiles_x,
                                                    c->d_raybuf.p, nullptr);
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
