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
// K1: trace + temporal (:230-472).  One wave per workgroup, one 8x8 pixel tile per wave (the reference's
// workgroup shape, :6): a tile's cost varies a lot (sky vs foliage), and single-wave workgroups release their CU
// slot the moment their tile is done.  Tiles are dealt round-robin over the XCDs (plain blockIdx order) because
// L1/L2 hit rates are already ~97 % / 99.5 % and balance matters more than L2 affinity.  Three phases per wave:
//   1. per pixel (lane = pixel): everything of main() up to the ray loop, then the rays of :403-430 are
//      GENERATED into the wave's slice of a ray buffer (2 x float4 per ray, compacted over valid pixels);
//   2. per ray (lane = worker): the 64 lanes pull rays from that slice as they become free (the tile's
//      rays/px * 64 rays form the wave's work list), so a lane whose ray ends early -- or whose own pixel is sky --
//      does not idle until the slowest ray of the wave finishes; each lane runs the traversal state machine of
//      traverse.hpp (while-while, LDS short stack) incl. the self-hit retry of traceAoRay (:193-200) and
//      stores visibility (sign bit = "a hit contributed occlusion") in the slice's result area;
//   3. per pixel again: visibility summed in ray order (:431-433) and the temporal accumulation (:440-471).
// The ray slice is written and read by the same wave (workgroup-scope fences = s_waitcnt only: one CU, one L1).
// ---------------------------------------------------------------------------------------------
constexpr int CUR_DONE = (int)0x80000000;

VKR_DI uint32_t ray_slice_float4s(int rayCount) { return (uint32_t)rayCount * 64u * 2u + (uint32_t)rayCount * 16u; }

constexpr int K1_BLOCK = 64;

template <bool STATS>
__global__ __launch_bounds__(K1_BLOCK) void k_trace(SceneDev sc, FrameParams fp, const uint16_t* depthP,
                                                           const uint16_t* histPrevP, uint16_t* histCurr,
                                                           uint32_t* hitMask, int y0, int y1, uint32_t tiles_x,
                                                           uint32_t n_tiles, float4* rayBuf,
                                                           unsigned long long* stats)
{
    __shared__ uint32_t s_stack[2 * VKR_LDS_STACK * K1_BLOCK];
    __shared__ float s_radius[K1_BLOCK];
    uint32_t* stack = s_stack + threadIdx.x;
    constexpr int B = K1_BLOCK;

    const uint32_t tile = blockIdx.x;
    const int lane = threadIdx.x;
    const int px = (int)(tile % tiles_x) * 8 + (lane & 7);
    const int py = y0 + (int)(tile / tiles_x) * 8 + (lane >> 3);
    const int W = fp.W, H = fp.H;
    const bool inside = (px < W && py < y1);                                    // :233-235
    const Img16 depth{depthP, W, H}, histPrev{histPrevP, W, H};
    const size_t o = (size_t)py * W + px;
    const uint32_t debugMode = fp.debugMode;                                    // :250
    const int rayCount = fp.rayCount;                                           // :252
    float4* slice = rayBuf + (size_t)blockIdx.x * ray_slice_float4s(rayCount);
    float* results = (float*)(slice + (size_t)rayCount * 128u);                 // [ray i][pixel lane]

    // ------------------------------------------------------------------ phase 1: per pixel
    bool valid = false;            // this lane's pixel casts rays
    V3 worldPos = v3(0, 0, 0), geomNormal = v3(0, 1, 0), origin = v3(0, 0, 0), tT = v3(0, 0, 0), tB = v3(0, 0, 0);
    float viewZ = 0.0f, bias = 0.0f, footprint = 0.0f, radius = 0.0f, tMin = 0.0f, cs = 1.0f, sn = 0.0f;
    if (inside) {
        if (hitMask) hitMask[o] = 0;
        float u = gclamp(((float)px + 0.5f) / (float)W, 0.0f, 1.0f);            // :237-238
        float v = gclamp(((float)py + 0.5f) / (float)H, 0.0f, 1.0f);
        float linearViewZ = tex_nearest_r(depth, u, v);                         // :239
        if (linearViewZ <= 1e-6f) {
            histCurr[o] = float_to_half(1.0f);                                  // :240-243
        } else {
            V3 viewPos = reconstructViewPosFromLinearZ(fp.invProj, u, v, linearViewZ);  // :247
            viewZ = linearViewZ;
            float wp[4];
            mul_m4_v4(fp.invView, viewPos.x, viewPos.y, viewPos.z, 1.0f, wp);   // :249
            worldPos = V3{wp[0], wp[1], wp[2]};
            const float baseRadius = fp.baseRadius, baseBias = fp.baseBias;     // :253-254

            float texelX = 1.0f / (float)W, texelY = 1.0f / (float)H;           // :257
            float discontinuityAbsZ = gmax(0.02f, baseRadius * 0.25f);          // :260
            const float DISCONTINUITY_REL_Z = 0.015f;

            float uXp = u + texelX, uXm = u - texelX, vYp = v + texelY, vYm = v - texelY;  // :263-266
            float depthXp = tex_nearest_r(depth, uXp, v);                       // :268-271
            float depthXm = tex_nearest_r(depth, uXm, v);
            float depthYp = tex_nearest_r(depth, u, vYp);
            float depthYm = tex_nearest_r(depth, u, vYm);
            depthXp = (depthXp <= 1e-6f) ? linearViewZ : depthXp;               // :272-275
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

            V3 viewPosX = viewPos, viewPosY = viewPos;                          // :302-317
            if (okXp && okXm) viewPosX = (dXp <= dXm) ? viewPosXp : viewPosXm;
            else if (okXp) viewPosX = viewPosXp;
            else if (okXm) viewPosX = viewPosXm;
            if (okYp && okYm) viewPosY = (dYp <= dYm) ? viewPosYp : viewPosYm;
            else if (okYp) viewPosY = viewPosYp;
            else if (okYm) viewPosY = viewPosYm;

            V3 viewDx = viewPosX - viewPos, viewDy = viewPosY - viewPos;        // :319-321
            footprint = gmax(length3(viewDx), length3(viewDy));

            V3 bestViewN = v3(0, 0, 0);                                         // :326-351
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
            if (bestScore < 1e29f) geomNormal = normalize3(mul_m3_v3(fp.invView, bestViewN));     // :353-359
            V3 camPos = V3{fp.camPos[0], fp.camPos[1], fp.camPos[2]};
            if (dot3(geomNormal, camPos - worldPos) < 0.0f) geomNormal = neg(geomNormal);          // :360-362

            float biasFromFootprint = footprint * 0.05f;                        // :365
            float biasMax = gmax(baseBias, gmin(baseRadius * 0.02f, 0.03f));    // :367
            bias = gclamp(gmax(baseBias, biasFromFootprint), baseBias, biasMax);  // :368
            float radiusScale = gclamp(sqrt_(gmax(footprint, 1e-4f) / 0.03f), 1.0f, 1.6f);  // :371
            radius = baseRadius * radiusScale;                                  // :372

            origin = offsetRay(worldPos, geomNormal);                           // :375
            buildTbn(geomNormal, &tT, &tB);                                     // :376
            tMin = gmax(1e-4f, bias * 0.25f);                                   // :377

            if (debugMode == 10u) histCurr[o] = float_to_half(gclamp(bias / 0.05f, 0.0f, 1.0f));            // :381-385
            else if (debugMode == 11u) histCurr[o] = float_to_half(gclamp(footprint / 0.05f, 0.0f, 1.0f));  // :386-390
            else valid = true;
        }
    }
    s_radius[threadIdx.x] = radius;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long vmask = __ballot(valid);
    const uint32_t nvalid = (uint32_t)__popcll(vmask);
    if (nvalid == 0) return;   // wave-uniform: nothing to trace, nothing to resolve
    if (valid) {
        const uint32_t rank = (uint32_t)__popcll(vmask & lt_mask);
        float frameTerm = (rayCount >= 4) ? (fp.frameIndex * 131.0f) : 0.0f;    // :395
        float seedBase = ((float)px * 1.0f + (float)py * 57.0f) + frameTerm;    // :396
        float angle = hash11(seedBase) * 6.28318530718f;                        // :397
        det_sincos(angle, &sn, &cs);                                            // :398-399
        for (int i = 0; i < rayCount; ++i) {                                    // :403-430
            float xix = (float)i / gmax((float)rayCount, 1.0f);                 // hammersley :79-82
            float xiy = radicalInverseVdC((uint32_t)i);
            float ax = xix - 0.5f, ay = xiy - 0.5f;
            float rx = (cs * ax + sn * ay) + 0.5f;                              // rot*(xi-0.5)+0.5, :400,405
            float ry = (-sn * ax + cs * ay) + 0.5f;
            rx = gclamp(rx, 0.0f, 1.0f);                                        // :406
            ry = gclamp(ry, 0.0f, 1.0f);
            float phi = 6.28318530718f * rx;                                    // :408
            float cosTheta = sqrt_(1.0f - ry), sinTheta = sqrt_(ry);            // :409-410
            float sphi, cphi;
            det_sincos(phi, &sphi, &cphi);
            V3 l = V3{cphi * sinTheta, sphi * sinTheta, cosTheta};              // :411
            V3 rd = V3{(tT.x * l.x + tB.x * l.y) + geomNormal.x * l.z, (tT.y * l.x + tB.y * l.y) + geomNormal.y * l.z,
                       (tT.z * l.x + tB.z * l.y) + geomNormal.z * l.z};
            V3 rayDir = normalize3(rd);                                         // :412
            float nd = gclamp(dot3(geomNormal, rayDir), 0.0f, 1.0f);            // :418
            float grazing = 1.0f - nd;
            float slopeScaled = gmax(bias * 0.5f, footprint * 0.02f);           // :421
            float dirOffset = slopeScaled * (0.15f + (1.5f * grazing) * grazing);  // :422
            V3 rayOrigin = origin + rayDir * dirOffset;                         // :423
            float tMinRay = tMin;                                               // :427-429
            tMinRay = gmax(tMinRay, dirOffset * 2.5f);
            tMinRay = gmax(tMinRay, ((radius * 0.03f) * grazing) * grazing);
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
        int cur = CUR_DONE, sp = 0;
        V3 ro = v3(0, 0, 0), rdir = v3(0, 0, 1), id = v3(0, 0, 0);
        float tmin = 0.0f, tmin0 = 0.0f, tmax = 0.0f, tbest = 0.0f, adv = 0.0f;
        uint32_t rid = 0;
        uint32_t spill[2 * (VKR_MAX_STACK - VKR_LDS_STACK)];
        TraceCounters cnt{0, 0, 0, 0};
        uint32_t nrays = 0;

        auto pop = [&]() {
            cur = CUR_DONE;
            const float tb = tbest * 1.000001f;
            while (sp > 0) {
                --sp;
                uint32_t key, ref;
                if (sp < VKR_LDS_STACK) { key = stack[(2 * sp) * B]; ref = stack[(2 * sp + 1) * B]; }
                else { key = spill[2 * (sp - VKR_LDS_STACK)]; ref = spill[2 * (sp - VKR_LDS_STACK) + 1]; }
                if (u2f(key & ~3u) <= tb) { cur = (int)ref; break; }
            }
        };
        auto push = [&](uint32_t key, uint32_t ref) {
            if (sp < VKR_LDS_STACK) { stack[(2 * sp) * B] = key; stack[(2 * sp + 1) * B] = ref; }
            else { spill[2 * (sp - VKR_LDS_STACK)] = key; spill[2 * (sp - VKR_LDS_STACK) + 1] = ref; }
            ++sp;
        };

        for (;;) {
            // ---- a ray whose traversal ended: self-hit retry (:193-200) or result (:187-189,205-212)
            if (active && cur == CUR_DONE) {
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
            // ---- idle lanes pull the next rays of the tile
            const unsigned long long idle = __ballot(!active);
            if (idle != 0ull && next < total) {
                if (!active) {
                    uint32_t k = next + (uint32_t)__popcll(idle & lt_mask);
                    if (k < total) {
                        float4 a = slice[(size_t)k * 2u], b = slice[(size_t)k * 2u + 1u];
                        ro = V3{a.x, a.y, a.z}; tmin = a.w; tmin0 = a.w;
                        rdir = V3{b.x, b.y, b.z}; rid = f2u(b.w);
                        id = V3{__builtin_amdgcn_rcpf(rdir.x), __builtin_amdgcn_rcpf(rdir.y), __builtin_amdgcn_rcpf(rdir.z)};
                        tmax = s_radius[rid & 63u];
                        tbest = tmax; found = false; retried = false; cur = 0; sp = 0; active = true;
                        if (STATS) nrays++;
                    }
                }
                next += (uint32_t)__popcll(idle);
            }
            if (__ballot(active) == 0ull) break;

            // ---- one step per iteration, chosen per wave: lanes holding a node and lanes holding a leaf do not
            // wait for each other's loops to drain; the cheaper triangle step runs as soon as a quarter as many
            // lanes want it as want a node step.
            const int n_node = __popcll(__ballot(cur >= 0));
            const int n_leaf = __popcll(__ballot(cur < 0 && cur != CUR_DONE));
            if (n_leaf == 0 || n_leaf * VKR_LEAF_VOTE < n_node) {
                if (cur >= 0) {
                    const float4* np = sc.nodes + (size_t)cur * 8;
                    float4 mnx = np[0], mny = np[1], mnz = np[2], mxx = np[3], mxy = np[4], mxz = np[5], ch = np[6];
                    if (STATS) { cnt.nodes++; if ((int)(__ffsll((long long)__ballot(1)) - 1) == lane) cnt.wave_node_iters++; }
                    const float tb = tbest * 1.000001f;
                    uint32_t k0 = box_key(mnx.x, mny.x, mnz.x, mxx.x, mxy.x, mxz.x, ro, id, tmin, tb, 0u);
                    uint32_t k1 = box_key(mnx.y, mny.y, mnz.y, mxx.y, mxy.y, mxz.y, ro, id, tmin, tb, 1u);
                    uint32_t k2 = box_key(mnx.z, mny.z, mnz.z, mxx.z, mxy.z, mxz.z, ro, id, tmin, tb, 2u);
                    uint32_t k3 = box_key(mnx.w, mny.w, mnz.w, mxx.w, mxy.w, mxz.w, ro, id, tmin, tb, 3u);
                    cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);  // ascending
                    auto ref_of = [&](uint32_t key) -> uint32_t {
                        uint32_t s = key & 3u;
                        float r = s == 0u ? ch.x : (s == 1u ? ch.y : (s == 2u ? ch.z : ch.w));
                        return f2u(r);
                    };
                    if (k0 == 0xffffffffu) {
                        pop();
                    } else {
                        if (k3 != 0xffffffffu) push(k3, ref_of(k3));
                        if (k2 != 0xffffffffu) push(k2, ref_of(k2));
                        if (k1 != 0xffffffffu) push(k1, ref_of(k1));
                        cur = (int)ref_of(k0);
                    }
                }
            } else if (cur < 0 && cur != CUR_DONE) {
                // ---- leaf: ONE triangle per step (queryAoClosestHit :125-182, closest committed hit)
                uint32_t x = (uint32_t)(~cur);
                uint32_t first = x & 0x0fffffffu, rest = (x >> 28) & 7u;     // rest = triangles after this one
                {
                    const float4* tp = sc.tris + (size_t)first * 3;
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
                    bool hit = (det != 0.0f) && (u >= 0.0f && u <= 1.0f) && (v >= 0.0f && u + v <= 1.0f) &&
                               (t > tmin && t < tbest);
                    if (hit) {
                        uint32_t mw = f2u(a.w);
                        uint32_t cls = mw >> 28;                                // candidate filter :142-173
                        if (cls == 1u) hit = false;
                        else if (cls == 2u) hit = alpha_test(sc, mw & 0x0fffffffu, f2u(b.w), u, v);
                    }
                    if (hit) { tbest = t; found = true; }
                }
                if (rest != 0u) cur = (int)(~((first + 1u) | ((rest - 1u) << 28)));
                else pop();
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
    if (hitMask) hitMask[o] = mask;
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
    uint32_t tiles_x = ((uint32_t)fp.W + 7u) / 8u, tiles_y = ((uint32_t)(y1 - y0) + 7u) / 8u;
    uint32_t n_tiles = tiles_x * tiles_y;
    // ray buffer: one slice per wave of the launch, sized for the current rays/px (grow-only)
    size_t need = (size_t)n_tiles * ((size_t)fp.rayCount * 64u * 2u + (size_t)fp.rayCount * 16u);
    if (c->d_raybuf.n < need) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        e = c->d_raybuf.alloc(need);
        if (e != hipSuccess) return e;
    }
    SceneDev sc = scene_dev(c);
    if (c->stats_enabled)
        k_trace<true><<<n_tiles, K1_BLOCK, 0, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1,
                                                          tiles_x, n_tiles, c->d_raybuf.p, c->d_stats.p);
    else
        k_trace<false><<<n_tiles, K1_BLOCK, 0, st>>>(sc, fp, depth, hist_prev, hist_curr, hit_mask, y0, y1,
                                                           tiles_x, n_tiles, c->d_raybuf.p, nullptr);
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
