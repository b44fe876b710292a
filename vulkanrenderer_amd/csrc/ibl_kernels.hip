// ibl_kernels.hip -- SURVEY 8(f) row 3 / BASELINE configs[1]: the IBL runtime precompute on the device.
//   EquirectToCubemap::convert (EquirectToCubemap.cpp:66-398, equirect_to_cubemap.frag)  -> k_ibl_equirect
//   IblPrecompute::compute      (IblPrecompute.cpp:75-561): irradiance_convolution.frag    -> k_ibl_irradiance
//                                                           prefilter.frag, every mip       -> k_ibl_prefilter
//                                                           brdf_integrate.frag             -> k_ibl_brdf
// The reference rasterises a unit cube six times per map and runs these fragment shaders; here every texel is a
// thread / wave / block of a compute kernel whose localPos is the face point at the pixel centre (the layout the
// reference's capture views, flipped projection and negative-height viewport produce, IblPrecompute.cpp:29-37,85-87,234).
// Texture-sampling bound float work: no MFMA; the environment cube (RGBA32F, 6 MB at 512^2) lives in L2.
// Numeric contract: DESIGN.md section 2, "IBL additions" (atan/asin polynomials, pow5, seamless bilinear cube
// sampling, two-level summation order of irradiance and prefilter).
#include "internal.hpp"

namespace vkr {

constexpr float IBL_PI = 3.14159265359f;               // the shaders' `const float PI`
constexpr float IBL_TWO_PI = 2.0f * 3.14159265359f;
constexpr float IBL_HALF_PI = 0.5f * 3.14159265359f;
constexpr float IBL_DELTA = 0.025f;                    // irradiance_convolution.frag:22
constexpr uint32_t IBL_SAMPLES = 1024u;                // prefilter.frag:54, brdf_integrate.frag:64

// cephes atanf / asinf with explicit fma (same text as oracle/ibl_oracle.cpp: this IS the contract)
VKR_DI float det_atan(float xx)
{
    float x = __builtin_fabsf(xx), y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = fma_(z, 8.05374449538e-2f, -1.38776856032E-1f);
    p = fma_(z, p, 1.99777106478E-1f);
    p = fma_(z, p, -3.33329491539E-1f);
    y = y + fma_(p * z, x, x);
    return xx < 0.0f ? -y : y;
}
VKR_DI float det_atan2(float y, float x)
{
    if (x == 0.0f) return y > 0.0f ? 1.5707963267948966f : (y < 0.0f ? -1.5707963267948966f : 0.0f);
    if (y == 0.0f) return x < 0.0f ? 3.14159265358979323846f : 0.0f;
    float w = x < 0.0f ? (y < 0.0f ? -3.14159265358979323846f : 3.14159265358979323846f) : 0.0f;
    return w + det_atan(y / x);
}
VKR_DI float det_asin(float xx)
{
    float a = __builtin_fabsf(xx), x, z;
    if (a > 1.0f) a = 1.0f;
    if (a < 1.0e-4f) return xx;
    const bool flag = a > 0.5f;
    if (flag) { z = 0.5f * (1.0f - a); x = sqrt_(z); }
    else { x = a; z = x * x; }
    float p = fma_(z, 4.2163199048E-2f, 2.4181311049E-2f);
    p = fma_(z, p, 4.5470025998E-2f);
    p = fma_(z, p, 7.4953002686E-2f);
    p = fma_(z, p, 1.6666752422E-1f);
    z = fma_(p * z, x, x);
    if (flag) { z = z + z; z = 1.5707963267948966f - z; }
    return xx < 0.0f ? -z : z;
}

// ---------------------------------------------------------------- the cube
VKR_DI V3 face_dir(int f, float sc, float tc)
{
    switch (f) {
        case 0: return v3(1.0f, -tc, -sc);
        case 1: return v3(-1.0f, -tc, sc);
        case 2: return v3(sc, 1.0f, tc);
        case 3: return v3(sc, -1.0f, -tc);
        case 4: return v3(sc, -tc, 1.0f);
        default: return v3(-sc, -tc, -1.0f);
    }
}
VKR_DI V3 texel_dir(int f, int x, int y, int S)
{
    const float k = 2.0f / (float)S;
    return face_dir(f, ((float)x + 0.5f) * k - 1.0f, ((float)y + 0.5f) * k - 1.0f);
}

// texel across ONE edge of face f (one of i, j out of range by one), exact integer arithmetic: c = 2i+1-S are the
// texel centres, +-S the edges
VKR_DI size_t across_edge(int S, int f, int i, int j)
{
    int cs = 2 * i + 1 - S, ct = 2 * j + 1 - S;
    cs = cs > S ? S : (cs < -S ? -S : cs);
    ct = ct > S ? S : (ct < -S ? -S : ct);
    int px, py, pz;
    switch (f) {
        case 0: px = S; py = -ct; pz = -cs; break;
        case 1: px = -S; py = -ct; pz = cs; break;
        case 2: px = cs; py = S; pz = ct; break;
        case 3: px = cs; py = -S; pz = -ct; break;
        case 4: px = cs; py = -ct; pz = S; break;
        default: px = -cs; py = -ct; pz = -S; break;
    }
    const int major = f >> 1;
    int g = f;
    if (major != 0 && (px == S || px == -S)) g = px < 0 ? 1 : 0;
    if (major != 1 && (py == S || py == -S)) g = py < 0 ? 3 : 2;
    if (major != 2 && (pz == S || pz == -S)) g = pz < 0 ? 5 : 4;
    int ns, nt;
    switch (g) {
        case 0: ns = -pz; nt = -py; break;
        case 1: ns = pz; nt = -py; break;
        case 2: ns = px; nt = pz; break;
        case 3: ns = px; nt = -pz; break;
        case 4: ns = px; nt = -py; break;
        default: ns = -px; nt = -py; break;
    }
    const int io = ns >= S ? S - 1 : (ns <= -S ? 0 : (ns + S - 1) / 2);
    const int jo = nt >= S ? S - 1 : (nt <= -S ? 0 : (nt + S - 1) / 2);
    return ((size_t)g * S + jo) * S + io;
}

// the four texels of the bilinear footprint [i0, i0+1] x [j0, j0+1] of face f (i0, j0 in [-1, S-1]): seamless -- texels
// beyond an edge come from the neighbouring face, the missing one at a cube corner is the mean of the other three
VKR_DI void cube_footprint(const float4* cube, int S, int f, int i0, int j0, float4 v[4])
{
    int missing = -1;
    if (i0 >= 0 && j0 >= 0 && i0 + 1 < S && j0 + 1 < S) {       // interior: the common case
        const float4* p = cube + ((size_t)f * S + j0) * S + i0;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[S]; v[3] = p[S + 1];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + (k & 1), j = j0 + (k >> 1);
            const bool oi = i < 0 || i >= S, oj = j < 0 || j >= S;
            if (oi && oj) { missing = k; v[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
            else if (oi || oj) v[k] = cube[across_edge(S, f, i, j)];
            else v[k] = cube[((size_t)f * S + j) * S + i];
        }
        if (missing >= 0) {                                      // cube corner: mean of the other three
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k != missing) { acc.x = acc.x + v[k].x; acc.y = acc.y + v[k].y; acc.z = acc.z + v[k].z; }
            const float third = 1.0f / 3.0f;
            const float4 m = make_float4(acc.x * third, acc.y * third, acc.z * third, 0.0f);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k == missing) v[k] = m;
        }
    }
}

// samplerCube, LINEAR, one mip, seamless.  QUADS: `cube` is the footprint table of k_ibl_quads (one aligned 64-byte record
// per footprint: the scattered samples of the convolutions then touch ONE cache line instead of two rows of the image,
// and the edge / corner cases are resolved once per footprint instead of once per sample); same texels, same arithmetic.
template <bool QUADS>
VKR_DI V3 cube_sample(const float4* cube, int S, V3 r)
{
    const float ax = __builtin_fabsf(r.x), ay = __builtin_fabsf(r.y), az = __builtin_fabsf(r.z);
    float sc, tc, ma;
    int f;
    if (az >= ay && az >= ax) { f = r.z < 0.0f ? 5 : 4; sc = r.z < 0.0f ? -r.x : r.x; tc = -r.y; ma = az; }
    else if (ay >= ax) { f = r.y < 0.0f ? 3 : 2; sc = r.x; tc = r.y < 0.0f ? -r.z : r.z; ma = ay; }
    else { f = r.x < 0.0f ? 1 : 0; sc = r.x < 0.0f ? r.z : -r.z; tc = -r.y; ma = ax; }
    const float s = 0.5f * (sc / ma) + 0.5f, t = 0.5f * (tc / ma) + 0.5f;
    const float x = s * (float)S - 0.5f, y = t * (float)S - 0.5f;
    const float fx = __builtin_floorf(x), fy = __builtin_floorf(y), wx = x - fx, wy = y - fy;
    const int i0 = (int)fx, j0 = (int)fy;
    float4 v[4];
    if (QUADS) {
        // |sc|, |tc| <= ma, so x, y lie in [-0.5, S - 0.5] and i0, j0 in [-1, S-1]; clamped all the same (NaN directions)
        const int ii = iclamp(i0 + 1, 0, S), jj = iclamp(j0 + 1, 0, S);
        const float4* q = cube + (((size_t)f * (size_t)(S + 1) + (size_t)jj) * (size_t)(S + 1) + (size_t)ii) * 4u;
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else {
        cube_footprint(cube, S, f, i0, j0, v);
    }
    V3 o;
    { const float top = v[0].x + wx * (v[1].x - v[0].x), bot = v[2].x + wx * (v[3].x - v[2].x); o.x = top + wy * (bot - top); }
    { const float top = v[0].y + wx * (v[1].y - v[0].y), bot = v[2].y + wx * (v[3].y - v[2].y); o.y = top + wy * (bot - top); }
    { const float top = v[0].z + wx * (v[1].z - v[0].z), bot = v[2].z + wx * (v[3].z - v[2].z); o.z = top + wy * (bot - top); }
    return o;
}

// footprint table: record ((f * (S+1) + j0+1) * (S+1) + i0+1) = the four texels cube_footprint(f, i0, j0) returns
__global__ __launch_bounds__(256) void k_ibl_quads(const float4* cube, int S, float4* quads)
{
    const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
    const size_t per = (size_t)(S + 1) * (size_t)(S + 1);
    if (t >= 6u * per) return;
    const int f = (int)(t / per), jj = (int)((t % per) / (size_t)(S + 1)), ii = (int)(t % (size_t)(S + 1));
    float4 v[4];
    cube_footprint(cube, S, f, ii - 1, jj - 1, v);
    float4* q = quads + t * 4u;
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
}

// ---------------------------------------------------------------- equirect_to_cubemap.frag
__global__ __launch_bounds__(256) void k_ibl_equirect(const float4* eq, int W, int H, float4* cube, int S)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 6 * S * S) return;
    const int f = t / (S * S), y = (t / S) % S, x = t % S;
    const V3 v = normalize3(texel_dir(f, x, y, S));                              // :25
    float a = det_atan2(v.z, v.x), b = det_asin(v.y);                            // :14
    a = a * 0.1591f; b = b * 0.3183f;                                            // :15
    a = a + 0.5f; b = b + 0.5f;                                                  // :16
    const float u = a, w = 1.0f - b;                                             // :19
    const float px = u * (float)W - 0.5f, py = w * (float)H - 0.5f;              // LINEAR, U repeat, V clamp to edge
    const float fx = __builtin_floorf(px), fy = __builtin_floorf(py), wx = px - fx, wy = py - fy;
    int i0 = (int)fx, j0 = (int)fy, i1 = i0 + 1, j1 = j0 + 1;
    i0 = ((i0 % W) + W) % W; i1 = ((i1 % W) + W) % W;
    j0 = iclamp(j0, 0, H - 1); j1 = iclamp(j1, 0, H - 1);
    const float4 t00 = eq[(size_t)j0 * W + i0], t10 = eq[(size_t)j0 * W + i1];
    const float4 t01 = eq[(size_t)j1 * W + i0], t11 = eq[(size_t)j1 * W + i1];
    float4 o;
    { const float top = t00.x + wx * (t10.x - t00.x), bot = t01.x + wx * (t11.x - t01.x); o.x = top + wy * (bot - top); }
    { const float top = t00.y + wx * (t10.y - t00.y), bot = t01.y + wx * (t11.y - t01.y); o.y = top + wy * (bot - top); }
    { const float top = t00.z + wx * (t10.z - t00.z), bot = t01.z + wx * (t11.z - t01.z); o.z = top + wy * (bot - top); }
    o.w = 1.0f;                                                                  // :27
    cube[t] = o;
}

// ---------------------------------------------------------------- irradiance_convolution.frag
// tables[0..nphi) = (sin phi, cos phi), tables[256..256+ntheta) = (sin theta, cos theta); phi / theta accumulated by
// repeated float addition exactly like the shader's loop variables (:24-25)
__global__ void k_ibl_irr_tables(float2* tables, int nphi, int ntheta)
{
    if (threadIdx.x == 0) {
        float phi = 0.0f;
        for (int p = 0; p < nphi; ++p) { float s, c; det_sincos(phi, &s, &c); tables[p] = make_float2(s, c); phi += IBL_DELTA; }
    } else if (threadIdx.x == 64) {
        float th = 0.0f;
        for (int t = 0; t < ntheta; ++t) { float s, c; det_sincos(th, &s, &c); tables[256 + t] = make_float2(s, c); th += IBL_DELTA; }
    }
}

// one block per output texel; thread p owns the phi row p (sequential over theta), rows are added in order
template <bool QUADS>
__global__ __launch_bounds__(256) void k_ibl_irradiance(const float4* cube, int S, const float2* tables, int nphi, int ntheta,
                                                        uint16_t* out, int So)
{
    __shared__ float2 s_theta[256];
    __shared__ float s_row[256][3];
    const int texel = blockIdx.x;
    const int f = texel / (So * So), y = (texel / So) % So, x = texel % So;
    if ((int)threadIdx.x < ntheta) s_theta[threadIdx.x] = tables[256 + threadIdx.x];
    __syncthreads();
    const V3 N = normalize3(texel_dir(f, x, y, So));                             // :14
    V3 up = __builtin_fabsf(N.y) < 0.999f ? v3(0.0f, 1.0f, 0.0f) : v3(0.0f, 0.0f, 1.0f);   // :18
    const V3 right = normalize3(cross3(up, N));                                  // :19
    up = cross3(N, right);                                                       // :20
    const int p = threadIdx.x;
    if (p < nphi) {
        const float2 ph = tables[p];
        float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
        for (int t = 0; t < ntheta; ++t) {
            const float st = s_theta[t].x, ct = s_theta[t].y;
            const float tx = st * ph.y, ty = st * ph.x, tz = ct;                 // :26
            const V3 sv = (right * tx + up * ty) + N * tz;                       // :27
            const V3 c = cube_sample<QUADS>(cube, S, sv);                        // :29
            r0 = r0 + (c.x * ct) * st; r1 = r1 + (c.y * ct) * st; r2 = r2 + (c.z * ct) * st;
        }
        s_row[p][0] = r0; s_row[p][1] = r1; s_row[p][2] = r2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float acc = 0.0f;
        for (int q = 0; q < nphi; ++q) acc = acc + s_row[q][threadIdx.x];
        const float nr = (float)(nphi * ntheta);                                 // :30
        out[(size_t)texel * 4 + threadIdx.x] = float_to_half((IBL_PI * acc) * (1.0f / nr));   // :33
    } else if (threadIdx.x == 3) {
        out[(size_t)texel * 4 + 3] = float_to_half(1.0f);
    }
}

// ---------------------------------------------------------------- prefilter.frag
// The tangent-space half vectors of ImportanceSampleGGX (:31-40) depend on (i, roughness) only: one table per mip.
__global__ __launch_bounds__(256) void k_ibl_ggx_table(float roughness, float4* table)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= IBL_SAMPLES) return;
    const float xi_x = (float)i / (float)IBL_SAMPLES;                            // :27
    uint32_t bits = i;                                                           // :17-24
    bits = (bits << 16) | (bits >> 16);
    bits = ((bits & 0x55555555u) << 1) | ((bits & 0xAAAAAAAAu) >> 1);
    bits = ((bits & 0x33333333u) << 2) | ((bits & 0xCCCCCCCCu) >> 2);
    bits = ((bits & 0x0F0F0F0Fu) << 4) | ((bits & 0xF0F0F0F0u) >> 4);
    bits = ((bits & 0x00FF00FFu) << 8) | ((bits & 0xFF00FF00u) >> 8);
    const float xi_y = (float)bits * 2.3283064365386963e-10f;
    const float a = roughness * roughness, a2 = a * a;
    const float phi = IBL_TWO_PI * xi_x;
    const float cosTheta = sqrt_((1.0f - xi_y) / (1.0f + (a2 - 1.0f) * xi_y));
    const float sinTheta = sqrt_(1.0f - cosTheta * cosTheta);
    float sp, cp;
    det_sincos(phi, &sp, &cp);
    table[i] = make_float4(cp * sinTheta, sp * sinTheta, cosTheta, 0.0f);
}

VKR_DI V3 ggx_to_world(float4 h, V3 N)                                           // :42-45
{
    const V3 up = __builtin_fabsf(N.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    const V3 tangent = normalize3(cross3(up, N));
    const V3 bitangent = cross3(N, tangent);
    return normalize3((tangent * h.x + bitangent * h.y) + N * h.z);
}

// one wave per output texel: lane l takes samples l, l+64, ... in order; the 64 partials are added in lane order
template <bool QUADS>
__global__ __launch_bounds__(256) void k_ibl_prefilter(const float4* cube, int S, const float4* table, uint16_t* out, int Sm)
{
    __shared__ float4 s_h[IBL_SAMPLES];
    __shared__ float4 s_part[4][64];
    __shared__ float s_tot[4][4];
    for (uint32_t i = threadIdx.x; i < IBL_SAMPLES; i += 256u) s_h[i] = table[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int texel = blockIdx.x * 4 + wave;
    const int total = 6 * Sm * Sm;
    const bool live = texel < total;
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, w = 0.0f;
    V3 N = v3(0.0f, 0.0f, 1.0f);
    if (live) {
        const int f = texel / (Sm * Sm), y = (texel / Sm) % Sm, x = texel % Sm;
        N = normalize3(texel_dir(f, x, y, Sm));                                  // :50
        for (uint32_t i = lane; i < IBL_SAMPLES; i += 64u) {                     // :58
            const V3 Hh = ggx_to_world(s_h[i], N);                               // :60
            const V3 L = normalize3(Hh * (2.0f * dot3(N, Hh)) - N);              // :61 (V = R = N)
            const float NdotL = gmax(dot3(N, L), 0.0f);                          // :63
            if (NdotL > 0.0f) {
                const V3 c = cube_sample<QUADS>(cube, S, L);
                c0 = c0 + c.x * NdotL; c1 = c1 + c.y * NdotL; c2 = c2 + c.z * NdotL;   // :65
                w = w + NdotL;                                                   // :66
            }
        }
    }
    s_part[wave][lane] = make_float4(c0, c1, c2, w);
    __syncthreads();
    if (lane < 4) {                       // lanes 0..2: colour channels, lane 3: total weight
        const float* base = &s_part[wave][0].x;
        float acc = base[lane];
        for (int l = 1; l < 64; ++l) acc = acc + base[4 * l + lane];
        s_tot[wave][lane] = acc;
    }
    __syncthreads();
    if (live && lane < 4) {
        const float tw = s_tot[wave][3];
        float r = 1.0f;                                                          // :72
        if (lane < 3) { const float tc = s_tot[wave][lane]; r = tw > 0.0f ? tc / tw : tc; }   // :70
        out[(size_t)texel * 4 + lane] = float_to_half(r);
    }
}

// ---------------------------------------------------------------- brdf_integrate.frag
VKR_DI float geometry_schlick_ggx(float NdotV, float roughness)                  // :41-45
{
    const float r = roughness + 1.0f;
    const float k = (r * r) / 8.0f;
    return NdotV / (NdotV * (1.0f - k) + k);
}
// one thread per texel, the shader's own running sums (bit-identical to the sequential restatement)
__global__ __launch_bounds__(256) void k_ibl_brdf(uint16_t* out, int S)
{
    __shared__ float4 s_xi[IBL_SAMPLES];       // (Xi.x, Xi.y, sin phi, cos phi): texel-independent
    for (uint32_t i = threadIdx.x; i < IBL_SAMPLES; i += 256u) {
        uint32_t bits = i;
        bits = (bits << 16) | (bits >> 16);
        bits = ((bits & 0x55555555u) << 1) | ((bits & 0xAAAAAAAAu) >> 1);
        bits = ((bits & 0x33333333u) << 2) | ((bits & 0xCCCCCCCCu) >> 2);
        bits = ((bits & 0x0F0F0F0Fu) << 4) | ((bits & 0xF0F0F0F0u) >> 4);
        bits = ((bits & 0x00FF00FFu) << 8) | ((bits & 0xFF00FF00u) >> 8);
        const float xi_x = (float)i / (float)IBL_SAMPLES;
        float sp, cp;
        det_sincos(IBL_TWO_PI * xi_x, &sp, &cp);
        s_xi[i] = make_float4(xi_x, (float)bits * 2.3283064365386963e-10f, sp, cp);
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= S * S) return;
    const int x = t % S, y = t / S;
    const float NdotV = ((float)x + 0.5f) / (float)S, roughness = ((float)y + 0.5f) / (float)S;   // outUV, :89
    const V3 V = v3(sqrt_(1.0f - NdotV * NdotV), 0.0f, NdotV);                   // :54-57
    const V3 N = v3(0.0f, 0.0f, 1.0f);
    const float a = roughness * roughness, a2 = a * a;
    // the tangent frame of ImportanceSampleGGX for N = +Z: |N.z| >= 0.999 -> up = +X (:35-37)
    const V3 tangent = normalize3(cross3(v3(1.0f, 0.0f, 0.0f), N));
    const V3 bitangent = cross3(N, tangent);
    float A = 0.0f, B = 0.0f;
    for (uint32_t i = 0; i < IBL_SAMPLES; ++i) {
        const float4 xi = s_xi[i];
        const float cosTheta = sqrt_((1.0f - xi.y) / (1.0f + (a2 - 1.0f) * xi.y));
        const float sinTheta = sqrt_(1.0f - cosTheta * cosTheta);
        const float hx = xi.w * sinTheta, hy = xi.z * sinTheta, hz = cosTheta;
        const V3 Hh = normalize3((tangent * hx + bitangent * hy) + N * hz);
        const float VH = dot3(V, Hh);
        const V3 L = normalize3(Hh * (2.0f * VH) - V);                           // :68
        const float NdotL = gmax(L.z, 0.0f), NdotH = gmax(Hh.z, 0.0f), VdotH = gmax(VH, 0.0f);
        if (NdotL > 0.0f) {
            const float G = geometry_schlick_ggx(NdotL, roughness) * geometry_schlick_ggx(NdotV, roughness);
            const float G_Vis = (G * VdotH) / (NdotH * NdotV);                   // :76
            const float m = 1.0f - VdotH, m2 = m * m;
            const float Fc = (m2 * m2) * m;                                      // :77
            A = A + (1.0f - Fc) * G_Vis;                                         // :78
            B = B + Fc * G_Vis;                                                  // :79
        }
    }
    out[(size_t)t * 2 + 0] = float_to_half(A / (float)IBL_SAMPLES);             // :82-83
    out[(size_t)t * 2 + 1] = float_to_half(B / (float)IBL_SAMPLES);
}

}  // namespace vkr

// ---------------------------------------------------------------- C ABI (include/vkr_rt.h, "IBL precompute")
using namespace vkr;

namespace {
int ibl_fail(vkr_ctx* c, int code, const char* msg) { if (c) c->err = msg; return code; }
int ibl_hip(vkr_ctx* c, hipError_t e, const char* what)
{
    if (e == hipSuccess) return VKR_OK;
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? VKR_ERR_OOM : VKR_ERR_HIP;
}
bool pow2(uint32_t v) { return v && !(v & (v - 1)); }
int ibl_scratch(vkr_ctx* c)
{
    if (c->d_ibl_tables.p) return VKR_OK;
    if (int rc = ibl_hip(c, c->d_ibl_tables.alloc(512 * 2 + IBL_SAMPLES * 4), "hipMalloc(ibl tables)")) return rc;
    return VKR_OK;
}
// Footprint table of the environment cube for the convolution kernels (64 B per footprint: 101 MB at 512^2); cubes above
// 2048^2 (1.6 GB of table) are sampled directly.  Returns the pointer the kernels sample from and whether it is the table.
constexpr uint32_t IBL_QUADS_MAX = 2048u;
int ibl_quads(vkr_ctx* c, const float* env_cube, uint32_t S, hipStream_t st, const float4** src, bool* quads)
{
    *src = (const float4*)env_cube; *quads = false;
    if (S > IBL_QUADS_MAX || !c->ibl_quads) return VKR_OK;
    const size_t n = 6u * (size_t)(S + 1u) * (size_t)(S + 1u);
    if (c->d_ibl_quads.n < n * 4u) {
        if (int rc = ibl_hip(c, hipStreamSynchronize(st), "hipStreamSynchronize")) return rc;
        if (int rc = ibl_hip(c, c->d_ibl_quads.alloc(n * 4u), "hipMalloc(ibl footprint table)")) return rc;
    }
    k_ibl_quads<<<(uint32_t)((n + 255u) / 256u), 256, 0, st>>>((const float4*)env_cube, (int)S, c->d_ibl_quads.p);
    if (int rc = ibl_hip(c, hipGetLastError(), "k_ibl_quads")) return rc;
    *src = c->d_ibl_quads.p; *quads = true;
    return VKR_OK;
}
}  // namespace

extern "C" {

uint32_t vkr_ibl_prefilter_mip_count(uint32_t size)
{
    uint32_t m = 0;
    while (m < 32 && (size >> m)) ++m;   // floor(log2(size)) + 1 (IblPrecompute.cpp:68-71)
    return m;
}

size_t vkr_ibl_prefilter_texels(uint32_t size)
{
    size_t n = 0;
    for (uint32_t m = 0; m < vkr_ibl_prefilter_mip_count(size); ++m) { const size_t s = (size >> m) ? (size >> m) : 1; n += 6 * s * s; }
    return n;
}

int vkr_ibl_equirect_to_cube(vkr_ctx* c, const float* equirect, uint32_t width, uint32_t height, float* env_cube,
                             uint32_t cube_size, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!equirect || !env_cube) return ibl_fail(c, VKR_ERR_INVALID_ARG, "equirect and env_cube must be non-NULL");
    if (width == 0 || height == 0 || width > 32768u || height > 32768u) return ibl_fail(c, VKR_ERR_INVALID_ARG, "bad equirect extent");
    if (!pow2(cube_size) || cube_size > 8192u) return ibl_fail(c, VKR_ERR_INVALID_ARG, "cube_size must be a power of two <= 8192");
    if (int rc = ibl_hip(c, hipSetDevice(c->device), "hipSetDevice")) return rc;
    const uint32_t n = 6u * cube_size * cube_size;
    k_ibl_equirect<<<(n + 255u) / 256u, 256, 0, (hipStream_t)stream>>>((const float4*)equirect, (int)width, (int)height,
                                                                      (float4*)env_cube, (int)cube_size);
    return ibl_hip(c, hipGetLastError(), "k_ibl_equirect");
}

int vkr_ibl_irradiance(vkr_ctx* c, const float* env_cube, uint32_t env_size, uint16_t* irradiance, uint32_t size, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!env_cube || !irradiance) return ibl_fail(c, VKR_ERR_INVALID_ARG, "env_cube and irradiance must be non-NULL");
    if (!pow2(env_size) || env_size > 8192u || !pow2(size) || size > 2048u) return ibl_fail(c, VKR_ERR_INVALID_ARG, "sizes must be powers of two");
    if (int rc = ibl_hip(c, hipSetDevice(c->device), "hipSetDevice")) return rc;
    if (int rc = ibl_scratch(c)) return rc;
    int nphi = 0, ntheta = 0;                                      // trip counts of the float loops (:24-25)
    for (float phi = 0.0f; phi < IBL_TWO_PI; phi += IBL_DELTA) ++nphi;
    for (float th = 0.0f; th < IBL_HALF_PI; th += IBL_DELTA) ++ntheta;
    if (nphi > 256 || ntheta > 256) return ibl_fail(c, VKR_ERR_LIMIT, "irradiance sample grid exceeds the kernel's tables");
    hipStream_t st = (hipStream_t)stream;
    float2* tables = (float2*)c->d_ibl_tables.p;
    k_ibl_irr_tables<<<1, 128, 0, st>>>(tables, nphi, ntheta);
    const float4* src; bool quads;
    if (int rc = ibl_quads(c, env_cube, env_size, st, &src, &quads)) return rc;
    if (quads) k_ibl_irradiance<true><<<6u * size * size, 256, 0, st>>>(src, (int)env_size, tables, nphi, ntheta, irradiance, (int)size);
    else k_ibl_irradiance<false><<<6u * size * size, 256, 0, st>>>(src, (int)env_size, tables, nphi, ntheta, irradiance, (int)size);
    return ibl_hip(c, hipGetLastError(), "k_ibl_irradiance");
}

int vkr_ibl_prefilter(vkr_ctx* c, const float* env_cube, uint32_t env_size, uint16_t* prefilter, uint32_t size, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!env_cube || !prefilter) return ibl_fail(c, VKR_ERR_INVALID_ARG, "env_cube and prefilter must be non-NULL");
    if (!pow2(env_size) || env_size > 8192u || !pow2(size) || size > 4096u) return ibl_fail(c, VKR_ERR_INVALID_ARG, "sizes must be powers of two");
    if (int rc = ibl_hip(c, hipSetDevice(c->device), "hipSetDevice")) return rc;
    if (int rc = ibl_scratch(c)) return rc;
    hipStream_t st = (hipStream_t)stream;
    float4* table = (float4*)(c->d_ibl_tables.p + 512 * 2);
    const uint32_t mips = vkr_ibl_prefilter_mip_count(size);
    const float4* src; bool quads;
    if (int rc = ibl_quads(c, env_cube, env_size, st, &src, &quads)) return rc;
    size_t base = 0;
    for (uint32_t mip = 0; mip < mips; ++mip) {                    // IblPrecompute.cpp:376-470
        const uint32_t sm = (size >> mip) ? (size >> mip) : 1u;
        const float roughness = (float)mip / (float)(mips - 1 > 1 ? mips - 1 : 1);   // :383
        k_ibl_ggx_table<<<IBL_SAMPLES / 256, 256, 0, st>>>(roughness, table);
        const uint32_t texels = 6u * sm * sm;
        if (quads) k_ibl_prefilter<true><<<(texels + 3u) / 4u, 256, 0, st>>>(src, (int)env_size, table, prefilter + base, (int)sm);
        else k_ibl_prefilter<false><<<(texels + 3u) / 4u, 256, 0, st>>>(src, (int)env_size, table, prefilter + base, (int)sm);
        base += (size_t)texels * 4;
    }
    return ibl_hip(c, hipGetLastError(), "k_ibl_prefilter");
}

int vkr_ibl_brdf_lut(vkr_ctx* c, uint16_t* lut, uint32_t size, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!lut) return ibl_fail(c, VKR_ERR_INVALID_ARG, "lut is NULL");
    if (size == 0 || size > 8192u) return ibl_fail(c, VKR_ERR_INVALID_ARG, "bad LUT size");
    if (int rc = ibl_hip(c, hipSetDevice(c->device), "hipSetDevice")) return rc;
    k_ibl_brdf<<<(size * size + 255u) / 256u, 256, 0, (hipStream_t)stream>>>(lut, (int)size);
    return ibl_hip(c, hipGetLastError(), "k_ibl_brdf");
}

int vkr_ibl_compute(vkr_ctx* c, const float* env_cube, uint32_t env_size, uint16_t* irradiance, uint32_t irradiance_size,
                    uint16_t* prefilter, uint32_t prefilter_size, uint16_t* brdf_lut, uint32_t brdf_lut_size, void* stream)
{
    if (int rc = vkr_ibl_irradiance(c, env_cube, env_size, irradiance, irradiance_size, stream)) return rc;
    if (int rc = vkr_ibl_prefilter(c, env_cube, env_size, prefilter, prefilter_size, stream)) return rc;
    return vkr_ibl_brdf_lut(c, brdf_lut, brdf_lut_size, stream);
}

}  // extern "C"
