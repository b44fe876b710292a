// ibl_oracle.cpp -- CPU ORACLE of the IBL runtime precompute (test infrastructure, NOT product code).
//
// SURVEY 8(f) row 3 / BASELINE configs[1].  A scalar float32 restatement of
//   assets/shaders/FragShaders/equirect_to_cubemap.frag   (EquirectToCubemap::convert, EquirectToCubemap.cpp:66-398)
//   assets/shaders/FragShaders/irradiance_convolution.frag (IblPrecompute.cpp:107-283)
//   assets/shaders/FragShaders/prefilter.frag              (IblPrecompute.cpp:285-470, roughness = mip / (mips - 1), :383)
//   assets/shaders/FragShaders/brdf_integrate.frag         (IblPrecompute.cpp:472-546)
// used only as the checker by tests/ and bench.py's cpu_baseline leg.
//
// PARITY UNPINNED: the reference has no tests or golden images for these maps and cannot be run here.  Pinned by
// known-answer tests (tests/test_ibl_oracle.py: constant environments, analytic BRDF-LUT limits, face orientation,
// seam continuity) and by its correspondence to the shader text cited at each function.
//
// Where GLSL/Vulkan leave the result to the implementation this file fixes one definition ("Numeric
// contract", IBL additions):
//   - the rasterised cube: fragment (x, y) of face f gets localPos = the face's point at the pixel centre,
//     sc = (x + 0.5) * (2 / S) - 1, tc likewise, in the Vulkan face layout (CAPTURE_VIEWS + flipped projection +
//     negative-height viewport of IblPrecompute.cpp:29-37,85-87,234 reduce to exactly that layout)
//   - atan(y, x), asin: cephes atanf/asinf polynomials with explicit fma; sin/cos/sqrt/normalize as in rtao_oracle.cpp
//   - pow(x, 5.0) = ((x*x)*(x*x))*x
//   - samplerCube, LINEAR, one mip (EquirectToCubemap.cpp:385-395): Vulkan face selection (z wins ties, then y),
//     full-float bilinear in lerp form, seamless edges (the texel across an edge comes from the adjacent face), the
//     missing texel at a cube corner = mean of the other three of the footprint
//   - sampler2D equirect: LINEAR, U repeat, V clamp-to-edge (Renderer.cpp:113-115), one mip
//   - RGBA16F / RG16F stores round to nearest even
//   - summation order: order = 0 is the shaders' single running sum; order = 1 is the product's two-level sum
//     (irradiance: one partial per phi row, rows added in order; prefilter: 64 interleaved partials added in order),
//     a documented deviation whose effect the tests bound (<= 1 fp16 ulp).
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace {

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline float gmax(float a, float b) { return (a < b) ? b : a; }
inline float dot3(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline V3 cross3(V3 a, V3 b) { return V3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))}; }
inline V3 normalize3(V3 a) { return a * (1.0f / sqrtf(dot3(a, a))); }
inline uint16_t float_to_half(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }

const float PI_F = 3.14159265359f;           // the shaders' `const float PI`
const float TWO_PI_F = 2.0f * 3.14159265359f;
const float HALF_PI_F = 0.5f * 3.14159265359f;

inline void det_sincos(float x, float* s_out, float* c_out)   // identical to rtao_oracle.cpp
{
    const double TWO_OVER_PI = 0.63661977236758134308, PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    double xd = (double)x, kd = nearbyint(xd * TWO_OVER_PI);
    double rd = xd - kd * PIO2_HI;
    rd = rd - kd * PIO2_LO;
    float r = (float)rd;
    int q = (int)((long long)kd & 3LL);
    float z = r * r;
    float ps = fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    ps = fmaf(z, ps, -1.6666654611E-1f);
    float sr = fmaf(r * z, ps, r);
    float pc = fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    pc = fmaf(z, pc, 4.166664568298827E-002f);
    float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    float s, c;
    switch (q) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
    *s_out = s; *c_out = c;
}

// cephes atanf: reduce to |x| <= tan(pi/8), degree-4 minimax in x^2
inline float det_atan(float xx)
{
    float x = fabsf(xx), y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = fmaf(z, 8.05374449538e-2f, -1.38776856032E-1f);
    p = fmaf(z, p, 1.99777106478E-1f);
    p = fmaf(z, p, -3.33329491539E-1f);
    y = y + fmaf(p * z, x, x);
    return xx < 0.0f ? -y : y;
}
// GLSL atan(y, x)
inline float det_atan2(float y, float x)
{
    if (x == 0.0f) return y > 0.0f ? 1.5707963267948966f : (y < 0.0f ? -1.5707963267948966f : 0.0f);
    if (y == 0.0f) return x < 0.0f ? 3.14159265358979323846f : 0.0f;
    float w = x < 0.0f ? (y < 0.0f ? -3.14159265358979323846f : 3.14159265358979323846f) : 0.0f;
    return w + det_atan(y / x);
}
// cephes asinf
inline float det_asin(float xx)
{
    float a = fabsf(xx), x, z;
    if (a > 1.0f) a = 1.0f;            // |v.y| of a normalised vector can exceed 1 by an ulp
    if (a < 1.0e-4f) return xx;
    bool flag = a > 0.5f;
    if (flag) { z = 0.5f * (1.0f - a); x = sqrtf(z); }
    else { x = a; z = x * x; }
    float p = fmaf(z, 4.2163199048E-2f, 2.4181311049E-2f);
    p = fmaf(z, p, 4.5470025998E-2f);
    p = fmaf(z, p, 7.4953002686E-2f);
    p = fmaf(z, p, 1.6666752422E-1f);
    z = fmaf(p * z, x, x);
    if (flag) { z = z + z; z = 1.5707963267948966f - z; }
    return xx < 0.0f ? -z : z;
}

// ---- the cube -----------------------------------------------------------------------------------------------
// localPos of fragment (x, y) on face f (Vulkan layout; see the header)
inline V3 face_dir(int f, float sc, float tc)
{
    switch (f) {
        case 0: return V3{1.0f, -tc, -sc};
        case 1: return V3{-1.0f, -tc, sc};
        case 2: return V3{sc, 1.0f, tc};
        case 3: return V3{sc, -1.0f, -tc};
        case 4: return V3{sc, -tc, 1.0f};
        default: return V3{-sc, -tc, -1.0f};
    }
}
inline V3 texel_dir(int f, int x, int y, int S)
{
    float k = 2.0f / (float)S;
    return face_dir(f, ((float)x + 0.5f) * k - 1.0f, ((float)y + 0.5f) * k - 1.0f);
}

struct Cube { const float* p; int S; };
inline const float* cube_texel(const Cube& c, int f, int i, int j) { return c.p + (((size_t)f * c.S + j) * c.S + i) * 4; }

// texel (i, j) of face f where ONE of i, j is out of [0, S) by one: the texel across that edge.  Integer
// coordinates c = 2*i + 1 - S (odd, texel centres), +-S = the face's edges.
inline void across_edge(int S, int f, int i, int j, int* fo, int* io, int* jo)
{
    int cs = 2 * i + 1 - S, ct = 2 * j + 1 - S;
    cs = cs > S ? S : (cs < -S ? -S : cs);
    ct = ct > S ? S : (ct < -S ? -S : ct);
    int P[3];
    switch (f) {   // face_dir scaled by S
        case 0: P[0] = S; P[1] = -ct; P[2] = -cs; break;
        case 1: P[0] = -S; P[1] = -ct; P[2] = cs; break;
        case 2: P[0] = cs; P[1] = S; P[2] = ct; break;
        case 3: P[0] = cs; P[1] = -S; P[2] = -ct; break;
        case 4: P[0] = cs; P[1] = -ct; P[2] = S; break;
        default: P[0] = -cs; P[1] = -ct; P[2] = -S; break;
    }
    int major = f >> 1, g = f;
    for (int a = 0; a < 3; ++a)
        if (a != major && (P[a] == S || P[a] == -S)) g = 2 * a + (P[a] < 0 ? 1 : 0);
    int ns, nt;
    switch (g) {   // inverse of face_dir
        case 0: ns = -P[2]; nt = -P[1]; break;
        case 1: ns = P[2]; nt = -P[1]; break;
        case 2: ns = P[0]; nt = P[2]; break;
        case 3: ns = P[0]; nt = -P[2]; break;
        case 4: ns = P[0]; nt = -P[1]; break;
        default: ns = -P[0]; nt = -P[1]; break;
    }
    *fo = g;
    *io = ns >= S ? S - 1 : (ns <= -S ? 0 : (ns + S - 1) / 2);
    *jo = nt >= S ? S - 1 : (nt <= -S ? 0 : (nt + S - 1) / 2);
}

inline void cube_sample(const Cube& c, V3 r, float out[3])
{
    const int S = c.S;
    float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z), sc, tc, ma;
    int f;
    if (az >= ay && az >= ax) { f = r.z < 0.0f ? 5 : 4; sc = r.z < 0.0f ? -r.x : r.x; tc = -r.y; ma = az; }
    else if (ay >= ax) { f = r.y < 0.0f ? 3 : 2; sc = r.x; tc = r.y < 0.0f ? -r.z : r.z; ma = ay; }
    else { f = r.x < 0.0f ? 1 : 0; sc = r.x < 0.0f ? r.z : -r.z; tc = -r.y; ma = ax; }
    float s = 0.5f * (sc / ma) + 0.5f, t = 0.5f * (tc / ma) + 0.5f;
    float x = s * (float)S - 0.5f, y = t * (float)S - 0.5f;
    float fx = floorf(x), fy = floorf(y), wx = x - fx, wy = y - fy;
    int i0 = (int)fx, j0 = (int)fy;
    const float* tx[4];
    int missing = -1;
    for (int k = 0; k < 4; ++k) {
        int i = i0 + (k & 1), j = j0 + (k >> 1);
        bool oi = i < 0 || i >= S, oj = j < 0 || j >= S;
        if (oi && oj) { missing = k; tx[k] = nullptr; }
        else if (oi || oj) { int g, gi, gj; across_edge(S, f, i, j, &g, &gi, &gj); tx[k] = cube_texel(c, g, gi, gj); }
        else tx[k] = cube_texel(c, f, i, j);
    }
    for (int ch = 0; ch < 3; ++ch) {
        float v[4];
        for (int k = 0; k < 4; ++k) v[k] = tx[k] ? tx[k][ch] : 0.0f;
        if (missing >= 0) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) if (k != missing) acc = acc + v[k];
            v[missing] = acc * (1.0f / 3.0f);
        }
        float top = v[0] + wx * (v[1] - v[0]);
        float bot = v[2] + wx * (v[3] - v[2]);
        out[ch] = top + wy * (bot - top);
    }
}

// ---- equirect_to_cubemap.frag ---------------------------------------------------------------------------------
inline void equirect_sample(const float* eq, int W, int H, float u, float v, float out[3])
{
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float fx = floorf(x), fy = floorf(y), wx = x - fx, wy = y - fy;
    int i0 = (int)fx, j0 = (int)fy;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = ((i0 % W) + W) % W; i1 = ((i1 % W) + W) % W;                    // U: repeat
    j0 = j0 < 0 ? 0 : (j0 > H - 1 ? H - 1 : j0); j1 = j1 < 0 ? 0 : (j1 > H - 1 ? H - 1 : j1);   // V: clamp to edge
    for (int ch = 0; ch < 3; ++ch) {
        float t00 = eq[((size_t)j0 * W + i0) * 4 + ch], t10 = eq[((size_t)j0 * W + i1) * 4 + ch];
        float t01 = eq[((size_t)j1 * W + i0) * 4 + ch], t11 = eq[((size_t)j1 * W + i1) * 4 + ch];
        float top = t00 + wx * (t10 - t00), bot = t01 + wx * (t11 - t01);
        out[ch] = top + wy * (bot - top);
    }
}
inline void spherical_uv(V3 v, float* u, float* w)      // equirect_to_cubemap.frag:12-21
{
    float a = det_atan2(v.z, v.x), b = det_asin(v.y);
    a = a * 0.1591f; b = b * 0.3183f;                     // :15
    a = a + 0.5f; b = b + 0.5f;                           // :16
    *u = a; *w = 1.0f - b;                                // :19
}

// ---- prefilter.frag / brdf_integrate.frag helpers -------------------------------------------------------------
inline float radical_inverse(uint32_t bits)             // prefilter.frag:17-24
{
    bits = (bits << 16) | (bits >> 16);
    bits = ((bits & 0x55555555u) << 1) | ((bits & 0xAAAAAAAAu) >> 1);
    bits = ((bits & 0x33333333u) << 2) | ((bits & 0xCCCCCCCCu) >> 2);
    bits = ((bits & 0x0F0F0F0Fu) << 4) | ((bits & 0xF0F0F0F0u) >> 4);
    bits = ((bits & 0x00FF00FFu) << 8) | ((bits & 0xFF00FF00u) >> 8);
    return (float)bits * 2.3283064365386963e-10f;
}
inline V3 importance_sample_ggx(float xi_x, float xi_y, V3 N, float roughness)   // prefilter.frag:30-46
{
    float a = roughness * roughness, a2 = a * a;
    float phi = TWO_PI_F * xi_x;
    float cosTheta = sqrtf((1.0f - xi_y) / (1.0f + (a2 - 1.0f) * xi_y));
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float sp, cp;
    det_sincos(phi, &sp, &cp);
    V3 H{cp * sinTheta, sp * sinTheta, cosTheta};
    V3 up = fabsf(N.z) < 0.999f ? V3{0.0f, 0.0f, 1.0f} : V3{1.0f, 0.0f, 0.0f};
    V3 tangent = normalize3(cross3(up, N));
    V3 bitangent = cross3(N, tangent);
    return normalize3((tangent * H.x + bitangent * H.y) + N * H.z);
}
inline float geometry_schlick_ggx(float NdotV, float roughness)   // brdf_integrate.frag:41-45
{
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    return NdotV / (NdotV * (1.0f - k) + k);
}

}  // namespace

extern "C" {

float orc_atan2(float y, float x) { return det_atan2(y, x); }
float orc_asin(float x) { return det_asin(x); }
void orc_ibl_texel_dir(int f, int x, int y, int S, float out[3]) { V3 d = texel_dir(f, x, y, S); out[0] = d.x; out[1] = d.y; out[2] = d.z; }
void orc_ibl_sample_cube(const float* cube, int S, const float dir[3], float out[3]) { cube_sample(Cube{cube, S}, V3{dir[0], dir[1], dir[2]}, out); }
void orc_ibl_sample_equirect(const float* eq, int W, int H, const float dir[3], float out[3])
{
    float u, v;
    spherical_uv(normalize3(V3{dir[0], dir[1], dir[2]}), &u, &v);
    equirect_sample(eq, W, H, u, v, out);
}

// EquirectToCubemap::convert: RGBA32F equirect [H][W][4] -> RGBA32F cube [6][S][S][4]
void orc_ibl_equirect_to_cube(const float* eq, int W, int H, float* cube, int S)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int f = 0; f < 6; ++f)
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x) {
                float u, v, c[3];
                spherical_uv(normalize3(texel_dir(f, x, y, S)), &u, &v);     // :25
                equirect_sample(eq, W, H, u, v, c);                            // :26
                float* o = cube + (((size_t)f * S + y) * S + x) * 4;
                o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = 1.0f;             // :27
            }
}

// irradiance_convolution.frag -> RGBA16F [6][So][So][4]
void orc_ibl_irradiance(const float* cube, int S, uint16_t* out, int So, int order)
{
    // the loop variables of :24-25, accumulated exactly as the float loops do
    std::vector<float> phis, thetas;
    for (float phi = 0.0f; phi < TWO_PI_F; phi += 0.025f) phis.push_back(phi);
    for (float th = 0.0f; th < HALF_PI_F; th += 0.025f) thetas.push_back(th);
    std::vector<float> sp(phis.size()), cp(phis.size()), st(thetas.size()), ct(thetas.size());
    for (size_t i = 0; i < phis.size(); ++i) det_sincos(phis[i], &sp[i], &cp[i]);
    for (size_t i = 0; i < thetas.size(); ++i) det_sincos(thetas[i], &st[i], &ct[i]);
    const float nrSamples = (float)(phis.size() * thetas.size());              // :30 (exact in float)
    Cube c{cube, S};
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int f = 0; f < 6; ++f)
        for (int y = 0; y < So; ++y)
            for (int x = 0; x < So; ++x) {
                V3 N = normalize3(texel_dir(f, x, y, So));                       // :14
                V3 up = fabsf(N.y) < 0.999f ? V3{0.0f, 1.0f, 0.0f} : V3{0.0f, 0.0f, 1.0f};   // :18
                V3 right = normalize3(cross3(up, N));                            // :19
                up = cross3(N, right);                                           // :20
                float irr[3] = {0.0f, 0.0f, 0.0f};
                for (size_t p = 0; p < phis.size(); ++p) {
                    float row[3] = {0.0f, 0.0f, 0.0f};
                    float* acc = order ? row : irr;
                    for (size_t t = 0; t < thetas.size(); ++t) {
                        V3 ts{st[t] * cp[p], st[t] * sp[p], ct[t]};              // :26
                        V3 sv = (right * ts.x + up * ts.y) + N * ts.z;           // :27
                        float tex[3];
                        cube_sample(c, sv, tex);                                 // :29
                        for (int ch = 0; ch < 3; ++ch) acc[ch] = acc[ch] + (tex[ch] * ct[t]) * st[t];
                    }
                    if (order) for (int ch = 0; ch < 3; ++ch) irr[ch] = irr[ch] + row[ch];
                }
                uint16_t* o = out + (((size_t)f * So + y) * So + x) * 4;
                for (int ch = 0; ch < 3; ++ch) o[ch] = float_to_half((PI_F * irr[ch]) * (1.0f / nrSamples));   // :33
                o[3] = float_to_half(1.0f);
            }
}

// prefilter.frag over all mips; output mip-major: mip m = [6][So>>m][So>>m][4], roughness = m / (mips - 1)
void orc_ibl_prefilter(const float* cube, int S, uint16_t* out, int So, int order)
{
    int mips = 1;
    while ((So >> mips) >= 1) ++mips;                                            // floor(log2(So)) + 1, IblPrecompute.cpp:68-71
    Cube c{cube, S};
    size_t base = 0;
    for (int mip = 0; mip < mips; ++mip) {
        int Sm = So >> mip;
        if (Sm < 1) Sm = 1;
        float roughness = (float)mip / (float)(mips - 1 > 1 ? mips - 1 : 1);     // :383
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
        for (int f = 0; f < 6; ++f)
            for (int y = 0; y < Sm; ++y)
                for (int x = 0; x < Sm; ++x) {
                    V3 N = normalize3(texel_dir(f, x, y, Sm));                   // :50
                    V3 V = N;
                    float col[64][3], wgt[64];
                    const int lanes = order ? 64 : 1;
                    for (int l = 0; l < lanes; ++l) { col[l][0] = col[l][1] = col[l][2] = 0.0f; wgt[l] = 0.0f; }
                    for (uint32_t i = 0; i < 1024u; ++i) {                        // :58
                        int l = order ? (int)(i & 63u) : 0;
                        V3 H = importance_sample_ggx((float)i / 1024.0f, radical_inverse(i), N, roughness);
                        V3 L = normalize3(H * (2.0f * dot3(V, H)) - V);          // :61
                        float NdotL = gmax(dot3(N, L), 0.0f);                    // :63
                        if (NdotL > 0.0f) {
                            float tex[3];
                            cube_sample(c, L, tex);
                            for (int ch = 0; ch < 3; ++ch) col[l][ch] = col[l][ch] + tex[ch] * NdotL;   // :65
                            wgt[l] = wgt[l] + NdotL;                              // :66
                        }
                    }
                    float tc[3] = {col[0][0], col[0][1], col[0][2]}, tw = wgt[0];
                    for (int l = 1; l < lanes; ++l) { for (int ch = 0; ch < 3; ++ch) tc[ch] = tc[ch] + col[l][ch]; tw = tw + wgt[l]; }
                    uint16_t* o = out + base + (((size_t)f * Sm + y) * Sm + x) * 4;
                    for (int ch = 0; ch < 3; ++ch) o[ch] = float_to_half(tw > 0.0f ? tc[ch] / tw : tc[ch]);   // :70
                    o[3] = float_to_half(1.0f);
                }
        base += (size_t)6 * Sm * Sm * 4;
    }
}

// brdf_integrate.frag -> RG16F [S][S][2]; column = NdotV, row = roughness (brdf_quad.vert, IblPrecompute.cpp:50-55,517)
void orc_ibl_brdf_lut(uint16_t* out, int S)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < S; ++y)
        for (int x = 0; x < S; ++x) {
            float NdotV = ((float)x + 0.5f) / (float)S, roughness = ((float)y + 0.5f) / (float)S;
            V3 V{sqrtf(1.0f - NdotV * NdotV), 0.0f, NdotV};                     // :54-57
            V3 N{0.0f, 0.0f, 1.0f};
            float A = 0.0f, B = 0.0f;
            for (uint32_t i = 0; i < 1024u; ++i) {
                V3 H = importance_sample_ggx((float)i / 1024.0f, radical_inverse(i), N, roughness);
                float VH = dot3(V, H);
                V3 L = normalize3(H * (2.0f * VH) - V);                          // :68
                float NdotL = gmax(L.z, 0.0f), NdotH = gmax(H.z, 0.0f), VdotH = gmax(VH, 0.0f);
                if (NdotL > 0.0f) {
                    float G = geometry_schlick_ggx(NdotL, roughness) * geometry_schlick_ggx(NdotV, roughness);   // :47-51
                    float G_Vis = (G * VdotH) / (NdotH * NdotV);                 // :76
                    float m = 1.0f - VdotH, m2 = m * m;
                    float Fc = (m2 * m2) * m;                                    // :77 pow(., 5.0)
                    A = A + (1.0f - Fc) * G_Vis;                                 // :78
                    B = B + Fc * G_Vis;                                          // :79
                }
            }
            out[((size_t)y * S + x) * 2 + 0] = float_to_half(A / 1024.0f);      // :82
            out[((size_t)y * S + x) * 2 + 1] = float_to_half(B / 1024.0f);
        }
}

}  // extern "C"
