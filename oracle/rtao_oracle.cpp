// rtao_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// A scalar float32 restatement of the reference's RTAO / shadow-ray path, used only as the checker by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in vulkanrenderer_amd/ may
// import, link or call this file.
//
// PARITY UNPINNED: the reference (chenjunjia-java/VulkanRenderer) has no tests, no golden vectors and
// cannot be built or run here (Windows-only CMake, Vulkan SDK, rayQueryEXT inside the driver; SURVEY.md
// 8c).  This restatement is therefore pinned only by hand-derived known-answer cases (tests/test_oracle_*.py)
// and by its line-by-line correspondence to the shader sources cited at every function.
//
// What is restated (paths relative to the reference root):
//   assets/shaders/CompShaders/rtao_trace_half.comp   (K1: trace + temporal)
//   assets/shaders/CompShaders/rtao_atrous.comp       (K2: A-Trous)
//   assets/shaders/CompShaders/rtao_upsample.comp     (K3: upsample / copy)
//   assets/shaders/FragShaders/pbr.frag:169-260       (shadow rays)
//   app/src/Rendering/pass/RtaoComputePass.cpp:503-571 (dispatch order, step/iteration, final index)
//
// Where GLSL leaves precision to the driver (sin/cos/exp/pow/inversesqrt, texture filtering, the BVH
// itself) this file FIXES one definition, documented in DESIGN.md "Numeric contract":
//   - sin/cos: double-precision Cody-Waite reduction by pi/2 + float minimax polynomials (cephes sinf/cosf)
//   - exp(x<=0): cephes expf scheme with explicit fmaf; exp(x < -87) = 0
//   - pow(x,3) = (x*x)*x ; pow(x,32) = five squarings ; inversesqrt(x) = 1/sqrt(x) (both correctly rounded)
//   - dot/cross use explicit fmaf chains (dot3/cross3 below); everything else is unfused (-ffp-contract=off)
//   - "TerminateOnFirstHit" is resolved as CLOSEST committed hit (a legal outcome of any-order traversal)
//   - ray/triangle: Moller-Trumbore, no culling, t strictly inside (tMin, tMax)
//   - LINEAR sampling = full-float bilinear, clamp-to-edge; NEAREST = floor(uv*size) clamped
// The CPU BVH here (median split, <=4 triangles per leaf) is deliberately NOT the product's LBVH: results
// must agree through any correct acceleration structure.  A brute-force mode exists for small scenes.
//
// Build: oracle/Makefile  (g++ -O2 -ffp-contract=off -mavx2 -mfma -mf16c -fopenmp)

#include <immintrin.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <atomic>
#include <map>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// numeric contract
// ------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
struct V2 { float x, y; };

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }

inline float gmax(float a, float b) { return (a < b) ? b : a; }   // GLSL max: y if x<y else x
inline float gmin(float a, float b) { return (b < a) ? b : a; }   // GLSL min: y if y<x else x
inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
inline int iclamp(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

inline float dot3(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline V3 cross3(V3 a, V3 b)
{
    return V3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
inline float inversesqrt(float x) { return 1.0f / sqrtf(x); }
inline V3 normalize3(V3 a) { return a * inversesqrt(dot3(a, a)); }
inline float length3(V3 a) { return sqrtf(dot3(a, a)); }

inline float half_to_float(uint16_t h) { return _cvtsh_ss(h); }
inline uint16_t float_to_half(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }

inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// sin/cos: k = rint(x * 2/pi) and r = x - k*pi/2 in double (two-term pi/2), polynomials in float.
inline void det_sincos(float x, float* s_out, float* c_out)
{
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_LO = 6.12323399573676603587e-17;
    double xd = (double)x;
    double kd = nearbyint(xd * TWO_OVER_PI);
    double rd = xd - kd * PIO2_HI;
    rd = rd - kd * PIO2_LO;
    float r = (float)rd;
    int q = (int)((long long)kd & 3LL);
    float z = r * r;
    // cephes sinf / cosf minimax on [-pi/4, pi/4]
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
    *s_out = s;
    *c_out = c;
}
inline float det_sin(float x) { float s, c; det_sincos(x, &s, &c); return s; }

// exp for x <= 0 (all call sites have non-positive arguments); exp(x < -87) = 0, NaN -> NaN.
inline float det_exp(float x)
{
    if (x < -87.0f) return 0.0f;
    float k = nearbyintf(x * 1.44269504088896341f);
    float r = fmaf(k, -0.693359375f, x);
    r = fmaf(k, 2.12194440e-4f, r);
    float p = fmaf(r, 1.9875691500E-4f, 1.3981999507E-3f);
    p = fmaf(r, p, 8.3334519073E-3f);
    p = fmaf(r, p, 4.1665795894E-2f);
    p = fmaf(r, p, 1.6666665459E-1f);
    p = fmaf(r, p, 5.0000001201E-1f);
    p = fmaf(r * r, p, r) + 1.0f;
    int ki = (int)k;
    return p * u2f((uint32_t)(ki + 127) << 23);
}
inline float pow3(float x) { return (x * x) * x; }
inline float pow32(float x) { x = x * x; x = x * x; x = x * x; x = x * x; x = x * x; return x; }
inline float smoothstep(float e0, float e1, float x)
{
    float t = gclamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}
inline float mixf(float a, float b, float w) { return a * (1.0f - w) + b * w; }
inline float fractf(float x) { return x - floorf(x); }

// column-major mat4 (GLM): m[c*4+r]
inline void mul_m4_v4(const float* m, float x, float y, float z, float w, float out[4])
{
    for (int r = 0; r < 4; ++r) out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}
inline V3 mul_m3_v3(const float* m, V3 v)  // upper-left 3x3 of a column-major mat4
{
    return V3{(m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
              (m[2] * v.x + m[6] * v.y) + m[10] * v.z};
}

// inverse(mat4): cofactor expansion in double, rounded once to float (GLSL leaves precision open).
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

// ------------------------------------------------------------------------------------------------
// scene + acceleration structure (the oracle's own; stands in for the driver's BLAS/TLAS)
// ------------------------------------------------------------------------------------------------
struct Tri {
    V3 v0, e1, e2;
    uint32_t mesh, prim, flat;
    uint32_t opaque;  // BLAS geometry opaque flag of the mesh (RayTracingContext.cpp:226-227)
    uint32_t inst;    // instance it came from (normal matrix of the G-buffer producer)
};

struct BNode {
    float bmin[3], bmax[3];
    int32_t left, right;  // internal: children; leaf: left = -(first+1), right = count
};

struct AlphaTex {
    std::vector<uint8_t> a;
    uint32_t w = 0, h = 0;
};

struct Scene {
    // per mesh (FrameManager.cpp:1264-1304)
    std::vector<uint32_t> lut_material, lut_index_offset;  // InstanceLUTEntry
    std::vector<uint32_t> mesh_vertex_offset, mesh_index_count, mesh_opaque;
    std::vector<float> positions;  // merged xyz
    std::vector<uint32_t> indices; // merged, GLOBAL vertex ids (voff + idx)
    std::vector<V2> uvs;           // merged
    std::vector<float> normals;    // merged xyz vertex normals (Vertex.h offset 12); has_normals per mesh
    std::vector<uint32_t> mesh_has_normals;
    std::vector<float> inst_xf;    // current 3x4 row-major transforms
    float material_params[256 * 4];
    AlphaTex tex[256];
    std::vector<uint32_t> inst_mesh;
    std::vector<Tri> tris;
    std::vector<uint32_t> order;   // triangle permutation used by the BVH leaves
    std::vector<BNode> nodes;
    bool brute = false;
    std::atomic<uint64_t> n_rays{0}, n_nodes{0}, n_tris{0};
};

struct BuildCtx {
    Scene* s;
    std::vector<V3> cmin, cmax, cent;
};

static void tri_bounds(const Tri& t, V3* lo, V3* hi)
{
    V3 a = t.v0, b = t.v0 + t.e1, c = t.v0 + t.e2;
    *lo = V3{std::min(a.x, std::min(b.x, c.x)), std::min(a.y, std::min(b.y, c.y)), std::min(a.z, std::min(b.z, c.z))};
    *hi = V3{std::max(a.x, std::max(b.x, c.x)), std::max(a.y, std::max(b.y, c.y)), std::max(a.z, std::max(b.z, c.z))};
}

static int build_node(BuildCtx& bc, uint32_t first, uint32_t count)
{
    Scene& s = *bc.s;
    int idx = (int)s.nodes.size();
    s.nodes.push_back(BNode{});
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    float clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
    for (uint32_t i = first; i < first + count; ++i) {
        uint32_t t = s.order[i];
        const float* a = &bc.cmin[t].x; const float* b = &bc.cmax[t].x; const float* c = &bc.cent[t].x;
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], a[k]); hi[k] = std::max(hi[k], b[k]);
            clo[k] = std::min(clo[k], c[k]); chi[k] = std::max(chi[k], c[k]);
        }
    }
    // conservative padding so that a triangle hit is never culled by rounding in the slab test
    for (int k = 0; k < 3; ++k) {
        float pad = 1e-5f * std::max(fabsf(lo[k]), fabsf(hi[k])) + 1e-6f;
        s.nodes[idx].bmin[k] = lo[k] - pad;
        s.nodes[idx].bmax[k] = hi[k] + pad;
    }
    if (count <= 4) {
        s.nodes[idx].left = -(int32_t)(first + 1);
        s.nodes[idx].right = (int32_t)count;
        return idx;
    }
    int axis = 0;
    float ext = chi[0] - clo[0];
    if (chi[1] - clo[1] > ext) { axis = 1; ext = chi[1] - clo[1]; }
    if (chi[2] - clo[2] > ext) { axis = 2; }
    uint32_t mid = first + count / 2;
    std::nth_element(s.order.begin() + first, s.order.begin() + mid, s.order.begin() + first + count,
                     [&](uint32_t a, uint32_t b) {
                         float ca = (&bc.cent[a].x)[axis], cb = (&bc.cent[b].x)[axis];
                         return ca < cb || (ca == cb && a < b);
                     });
    int l = build_node(bc, first, mid - first);
    int r = build_node(bc, mid, first + count - mid);
    s.nodes[idx].left = l;
    s.nodes[idx].right = r;
    return idx;
}

static void build_bvh(Scene& s)
{
    s.nodes.clear();
    size_t n = s.tris.size();
    s.order.resize(n);
    for (size_t i = 0; i < n; ++i) s.order[i] = (uint32_t)i;
    if (n == 0) return;
    BuildCtx bc;
    bc.s = &s;
    bc.cmin.resize(n); bc.cmax.resize(n); bc.cent.resize(n);
    for (size_t i = 0; i < n; ++i) {
        tri_bounds(s.tris[i], &bc.cmin[i], &bc.cmax[i]);
        bc.cent[i] = (bc.cmin[i] + bc.cmax[i]) * 0.5f;
    }
    s.nodes.reserve(n);
    build_node(bc, 0, (uint32_t)n);
}

// Instance transform: 3x4 row-major (RayTracingContext.cpp:28-44) applied to mesh-local positions.
inline V3 xform(const float* m, V3 p)
{
    return V3{((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
              ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]};
}

static void flatten(Scene& s, const float* inst_xf)
{
    // one world-space triangle per (instance, primitive); empty meshes contribute nothing
    // (RayTracingContext.cpp:206-208,230-233).  flat id = running index in instance-major order.
    s.tris.clear();
    s.inst_xf.assign(inst_xf, inst_xf + 12 * s.inst_mesh.size());
    uint32_t flat = 0;
    for (size_t i = 0; i < s.inst_mesh.size(); ++i) {
        uint32_t m = s.inst_mesh[i];
        const float* xf = inst_xf + 12 * i;
        uint32_t nprim = s.mesh_index_count[m] / 3;
        uint32_t ioff = s.lut_index_offset[m];
        for (uint32_t p = 0; p < nprim; ++p) {
            V3 v[3];
            for (int k = 0; k < 3; ++k) {
                uint32_t gi = s.indices[ioff + 3 * p + k];
                v[k] = xform(xf, V3{s.positions[3 * gi], s.positions[3 * gi + 1], s.positions[3 * gi + 2]});
            }
            Tri t;
            t.v0 = v[0]; t.e1 = v[1] - v[0]; t.e2 = v[2] - v[0];
            t.mesh = m; t.prim = p; t.flat = flat++; t.opaque = s.mesh_opaque[m]; t.inst = (uint32_t)i;
            s.tris.push_back(t);
        }
    }
}

// baseColor alpha, LOD 0, bilinear, REPEAT wrap (glTF default sampler). rtao_trace_half.comp:170.
static float sample_alpha(const AlphaTex& t, V2 uv)
{
    if (t.w == 0 || t.h == 0) return 1.0f;
    float x = uv.x * (float)t.w - 0.5f;
    float y = uv.y * (float)t.h - 0.5f;
    if (!(fabsf(x) < 1.0e9f) || !(fabsf(y) < 1.0e9f)) return 1.0f;
    float fx = floorf(x), fy = floorf(y);
    float ax = x - fx, ay = y - fy;
    int W = (int)t.w, H = (int)t.h;
    int ix0 = (int)fx % W; if (ix0 < 0) ix0 += W;
    int iy0 = (int)fy % H; if (iy0 < 0) iy0 += H;
    int ix1 = ix0 + 1; if (ix1 == W) ix1 = 0;
    int iy1 = iy0 + 1; if (iy1 == H) iy1 = 0;
    const float s = 1.0f / 255.0f;
    float t00 = (float)t.a[(size_t)iy0 * W + ix0] * s, t10 = (float)t.a[(size_t)iy0 * W + ix1] * s;
    float t01 = (float)t.a[(size_t)iy1 * W + ix0] * s, t11 = (float)t.a[(size_t)iy1 * W + ix1] * s;
    float top = t00 + ax * (t10 - t00);
    float bot = t01 + ax * (t11 - t01);
    return top + ay * (bot - top);
}

enum QueryMode { Q_AO = 0, Q_SHADOW = 1 };

// Contract-departure switches (tests/test_contract_departures.py only; 0 = the contract the GPU kernels implement).
//   1  A-Trous: the three per-tap quotients as IEEE divisions (rtao_atrous.comp:99,103 spell `/`) instead of reciprocal multiplies
//   2  A-Trous: every fetch through its sampler at the COMPUTED uv (rtao_atrous.comp:55-56,80-89: LINEAR for the AO image,
//      NEAREST for depth / normal at clamp(uv + offset)) instead of an exact texel fetch
//   4  queryAoClosestHit: gl_RayFlagsTerminateOnFirstHitEXT taken literally -- the first ACCEPTED candidate in traversal order
//      ends the query (rtao_trace_half.comp:131-139,176-181), instead of the closest committed hit
//   8  with 4: the other traversal order (children / triangles visited in reverse), a second legal driver behaviour
static int g_variant = 0;

struct Hit { float t; uint32_t tri; float u, v; };

// Candidate filter of queryAoClosestHit (rtao_trace_half.comp:141-175).
static bool ao_confirm(const Scene& s, const Tri& tr, float bu, float bv)
{
    if (tr.opaque) return true;                       // opaque geometry: auto-committed by the driver
    uint32_t mat = s.lut_material[tr.mesh];           // :144-146
    uint32_t ioff = s.lut_index_offset[tr.mesh];
    float cutoff = s.material_params[4 * mat + 0];    // :148-150
    float mode = s.material_params[4 * mat + 1];
    if (mode < 0.5f) return true;                     // :153-156
    if (mode > 1.5f) return false;                    // :158-160 BLEND: non-occluding
    float w0 = (1.0f - bu) - bv;                      // :165
    uint32_t i0 = s.indices[ioff + tr.prim * 3 + 0];  // :166-168
    uint32_t i1 = s.indices[ioff + tr.prim * 3 + 1];
    uint32_t i2 = s.indices[ioff + tr.prim * 3 + 2];
    V2 a = s.uvs[i0], b = s.uvs[i1], c = s.uvs[i2];
    V2 uv = V2{(a.x * w0 + b.x * bu) + c.x * bv, (a.y * w0 + b.y * bu) + c.y * bv};  // :169
    float alpha = sample_alpha(s.tex[mat], uv);       // :170
    return alpha >= cutoff;                           // :171
}

// Moller-Trumbore, no culling; accepts t strictly inside (tmin, tbest) or a tie with a lower flat id
// (ties only matter for the G-buffer's choice of triangle; t itself is unaffected).
static inline bool tri_test(const Tri& tr, V3 o, V3 d, float tmin, float tbest, uint32_t best_flat, bool tie,
                            float* t_out, float* u_out, float* v_out)
{
    V3 pvec = cross3(d, tr.e2);
    float det = dot3(tr.e1, pvec);
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    V3 tvec = o - tr.v0;
    float u = dot3(tvec, pvec) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    V3 qvec = cross3(tvec, tr.e1);
    float v = dot3(d, qvec) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float t = dot3(tr.e2, qvec) * inv;
    if (!(t > tmin)) return false;
    if (!(t < tbest || (tie && t == tbest && tr.flat < best_flat))) return false;
    *t_out = t; *u_out = u; *v_out = v;
    return true;
}

// centre_dir (MSAA prepass only): the fragment shader runs once per pixel per primitive, so the MASK discard of a
// candidate is decided with the barycentrics of the PIXEL-CENTRE ray on the triangle's plane (extrapolated when the
// centre is not covered), not with those of the sample ray.
static bool query(Scene& s, V3 o, V3 d, float tmin, float tmax, QueryMode mode, bool tie, Hit* hit, const V3* centre_dir = nullptr)
{
    uint64_t c_nodes = 0, c_tris = 0;
    float tbest = tmax;
    uint32_t best_flat = 0xffffffffu, best_tri = 0xffffffffu;
    float bu = 0, bv = 0;
    bool found = false;
    const bool first_found = (g_variant & 4) != 0 && !tie && centre_dir == nullptr;   // AO queries only (not the G-buffer producer)
    const bool reverse = first_found && (g_variant & 8) != 0;
    auto test = [&](uint32_t ti) -> bool {
        const Tri& tr = s.tris[ti];
        float t, u, v;
        ++c_tris;
        if (!tri_test(tr, o, d, tmin, tbest, best_flat, tie, &t, &u, &v)) return false;
        if (mode == Q_AO) {
            float cu = u, cv = v;
            if (centre_dir && !tr.opaque) {
                V3 pc = cross3(*centre_dir, tr.e2);
                float detc = dot3(tr.e1, pc);
                if (detc == 0.0f) return false;            // plane edge-on to the centre ray: no fragment value exists
                float invc = 1.0f / detc;
                V3 tvec = o - tr.v0;
                cu = dot3(tvec, pc) * invc;
                cv = dot3(*centre_dir, cross3(tvec, tr.e1)) * invc;
            }
            if (!ao_confirm(s, tr, cu, cv)) return false;
        }
        tbest = t; best_flat = tr.flat; best_tri = ti; bu = u; bv = v; found = true;
        return mode == Q_SHADOW || (first_found && mode == Q_AO);  // any-hit terminates
    };
    if (s.brute || s.nodes.empty()) {
        const uint32_t nt = (uint32_t)s.tris.size();
        for (uint32_t i = 0; i < nt; ++i)
            if (test(reverse ? nt - 1u - i : i)) break;
    } else {
        float idx = 1.0f / d.x, idy = 1.0f / d.y, idz = 1.0f / d.z;
        int stack[128];
        int sp = 0;
        stack[sp++] = 0;
        bool done = false;
        while (sp > 0 && !done) {
            const BNode& n = s.nodes[stack[--sp]];
            ++c_nodes;
            float t0 = (n.bmin[0] - o.x) * idx, t1 = (n.bmax[0] - o.x) * idx;
            float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
            t0 = (n.bmin[1] - o.y) * idy; t1 = (n.bmax[1] - o.y) * idy;
            tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
            t0 = (n.bmin[2] - o.z) * idz; t1 = (n.bmax[2] - o.z) * idz;
            tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
            tn = fmaxf(tn, tmin); tf = fminf(tf, tbest);
            if (!(tn * 0.999999f <= tf * 1.000001f)) continue;
            if (n.left < 0) {
                uint32_t first = (uint32_t)(-n.left - 1);
                for (uint32_t i = 0; i < (uint32_t)n.right; ++i)
                    if (test(s.order[first + (reverse ? (uint32_t)n.right - 1u - i : i)])) { done = true; break; }
            } else if (reverse) {
                stack[sp++] = n.right;
                stack[sp++] = n.left;
            } else {
                stack[sp++] = n.left;
                stack[sp++] = n.right;
            }
        }
    }
    s.n_rays.fetch_add(1, std::memory_order_relaxed);
    s.n_nodes.fetch_add(c_nodes, std::memory_order_relaxed);
    s.n_tris.fetch_add(c_tris, std::memory_order_relaxed);
    if (hit) { hit->t = tbest; hit->tri = best_tri; hit->u = bu; hit->v = bv; }
    return found;
}

// ------------------------------------------------------------------------------------------------
// samplers
// ------------------------------------------------------------------------------------------------
struct Img16 { const uint16_t* p; int w, h; };

// NEAREST, clamp-to-edge (FrameManager.cpp:890-899,961-970)
inline float tex_nearest_r(const Img16& im, float u, float v)
{
    int x = iclamp((int)floorf(u * (float)im.w), 0, im.w - 1);
    int y = iclamp((int)floorf(v * (float)im.h), 0, im.h - 1);
    return half_to_float(im.p[(size_t)y * im.w + x]);
}
// LINEAR, clamp-to-edge (FrameManager.cpp:1004-1011)
inline float tex_linear_r(const Img16& im, float u, float v)
{
    float x = u * (float)im.w - 0.5f, y = v * (float)im.h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = x - fx, ay = y - fy;
    int x0 = iclamp((int)fx, 0, im.w - 1), x1 = iclamp((int)fx + 1, 0, im.w - 1);
    int y0 = iclamp((int)fy, 0, im.h - 1), y1 = iclamp((int)fy + 1, 0, im.h - 1);
    float t00 = half_to_float(im.p[(size_t)y0 * im.w + x0]), t10 = half_to_float(im.p[(size_t)y0 * im.w + x1]);
    float t01 = half_to_float(im.p[(size_t)y1 * im.w + x0]), t11 = half_to_float(im.p[(size_t)y1 * im.w + x1]);
    float top = t00 + ax * (t10 - t00);
    float bot = t01 + ax * (t11 - t01);
    return top + ay * (bot - top);
}

// ------------------------------------------------------------------------------------------------
// rtao_trace_half.comp
// ------------------------------------------------------------------------------------------------
// :69-77
inline float radicalInverseVdC(uint32_t bits)
{
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)bits * 2.3283064365386963e-10f;
}
// :79-82
inline V2 hammersley(uint32_t i, uint32_t n) { return V2{(float)i / gmax((float)n, 1.0f), radicalInverseVdC(i)}; }
// :64-67
inline float hash11(float x) { return fractf(det_sin(x) * 43758.5453123f); }

// :84-90 ; returns columns t, b, n
inline void buildTbn(V3 n, V3* t, V3* b)
{
    V3 up = (fabsf(n.y) < 0.999f) ? v3(0, 1, 0) : v3(1, 0, 0);
    *t = normalize3(cross3(up, n));
    *b = cross3(n, *t);
}

// :92-100
inline V3 reconstructViewPosFromLinearZ(const float* invProj, float u, float v, float linearViewZ)
{
    float fv[4];
    mul_m4_v4(invProj, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 1.0f, 1.0f, fv);
    float w = gmax(fv[3], 1e-6f);
    V3 dir = normalize3(V3{fv[0] / w, fv[1] / w, fv[2] / w});
    float t = linearViewZ / gmax(-dir.z, 1e-6f);
    return dir * t;
}

// :112-123
inline V3 offsetRay(V3 p, V3 n)
{
    const float ORIGIN_EPS = 1.0f / 32.0f, FLOAT_SCALE = 1.0f / 65536.0f, INT_SCALE = 256.0f;
    int ox = (int)(n.x * INT_SCALE), oy = (int)(n.y * INT_SCALE), oz = (int)(n.z * INT_SCALE);
    float px = u2f((uint32_t)((int32_t)f2u(p.x) + ((p.x < 0.0f) ? -ox : ox)));
    float py = u2f((uint32_t)((int32_t)f2u(p.y) + ((p.y < 0.0f) ? -oy : oy)));
    float pz = u2f((uint32_t)((int32_t)f2u(p.z) + ((p.z < 0.0f) ? -oz : oz)));
    return V3{(fabsf(p.x) < ORIGIN_EPS) ? (p.x + FLOAT_SCALE * n.x) : px,
              (fabsf(p.y) < ORIGIN_EPS) ? (p.y + FLOAT_SCALE * n.y) : py,
              (fabsf(p.z) < ORIGIN_EPS) ? (p.z + FLOAT_SCALE * n.z) : pz};
}

// :125-182
inline bool queryAoClosestHit(Scene& s, V3 origin, V3 dir, float tMin, float tMax, float* tHit)
{
    Hit h;
    if (!query(s, origin, dir, tMin, tMax, Q_AO, false, &h)) { *tHit = tMax; return false; }
    *tHit = h.t;
    return true;
}

// :184-213 ; *occluded reports the per-ray hit/miss tap (1 = a hit contributed occlusion)
inline float traceAoRay(Scene& s, V3 origin, V3 dir, float tMin, float tMax, int* occluded)
{
    *occluded = 0;
    float t = tMax;
    if (!queryAoClosestHit(s, origin, dir, tMin, tMax, &t)) return 1.0f;
    if (t <= tMin * 1.5f) {
        float retryAdvance = gmax(tMin * 2.0f, 2e-4f);
        float retryT = tMax;
        if (!queryAoClosestHit(s, origin + dir * retryAdvance, dir, gmax(tMin, 1e-4f), tMax, &retryT)) return 1.0f;
        t = retryT + retryAdvance;
    }
    if (t <= tMin * 1.5f) return 1.0f;
    *occluded = 1;
    float x = gclamp((t - tMin) / gmax(tMax - tMin, 1e-3f), 0.0f, 1.0f);
    return pow3(x);
}

// :215-228
inline bool computePrevUv(const float* prevViewProj, V3 worldPos, V2* prevUv)
{
    float c[4];
    mul_m4_v4(prevViewProj, worldPos.x, worldPos.y, worldPos.z, 1.0f, c);
    if (c[3] <= 1e-5f) { *prevUv = V2{0, 0}; return false; }
    float nx = c[0] / c[3], ny = c[1] / c[3];
    *prevUv = V2{nx * 0.5f + 0.5f, ny * 0.5f + 0.5f};
    if (prevUv->x < 0.0f || prevUv->y < 0.0f || prevUv->x > 1.0f || prevUv->y > 1.0f) return false;
    return true;
}

struct Frame {
    int W, H;
    const float* ubo;
    float invProj[16], invView[16];
};

// :230-472, one pixel
static void trace_pixel(Scene& s, const Frame& f, const Img16& depth, const Img16& histPrev, uint16_t* histCurr,
                        uint32_t* hitMask, int px, int py)
{
    const float* ubo = f.ubo;
    const int W = f.W, H = f.H;
    size_t o = (size_t)py * W + px;
    if (hitMask) hitMask[o] = 0;
    float u = gclamp(((float)px + 0.5f) / (float)W, 0.0f, 1.0f);   // :237-238
    float v = gclamp(((float)py + 0.5f) / (float)H, 0.0f, 1.0f);
    float linearViewZ = tex_nearest_r(depth, u, v);               // :239
    if (linearViewZ <= 1e-6f) { histCurr[o] = float_to_half(1.0f); return; }  // :240-243

    V3 viewPos = reconstructViewPosFromLinearZ(f.invProj, u, v, linearViewZ);  // :247
    float viewZ = linearViewZ;
    float wp[4];
    mul_m4_v4(f.invView, viewPos.x, viewPos.y, viewPos.z, 1.0f, wp);          // :249
    V3 worldPos = V3{wp[0], wp[1], wp[2]};
    uint32_t debugMode = (uint32_t)(ubo[92 + 3] + 0.5f);                       // :250

    int rayCount = iclamp((int)(ubo[96 + 1] + 0.5f), 1, 16);                   // :252
    float baseRadius = gmax(ubo[96 + 2], 0.01f);                               // :253
    float baseBias = gmax(ubo[96 + 3], 0.0005f);                               // :254

    float texelX = 1.0f / (float)depth.w, texelY = 1.0f / (float)depth.h;      // :257
    float discontinuityAbsZ = gmax(0.02f, baseRadius * 0.25f);                 // :260
    const float DISCONTINUITY_REL_Z = 0.015f;

    float uXp = u + texelX, uXm = u - texelX, vYp = v + texelY, vYm = v - texelY;  // :263-266
    float depthXp = tex_nearest_r(depth, uXp, v);                              // :268-271
    float depthXm = tex_nearest_r(depth, uXm, v);
    float depthYp = tex_nearest_r(depth, u, vYp);
    float depthYm = tex_nearest_r(depth, u, vYm);
    depthXp = (depthXp <= 1e-6f) ? linearViewZ : depthXp;                      // :272-275
    depthXm = (depthXm <= 1e-6f) ? linearViewZ : depthXm;
    depthYp = (depthYp <= 1e-6f) ? linearViewZ : depthYp;
    depthYm = (depthYm <= 1e-6f) ? linearViewZ : depthYm;

    V3 viewPosXp = reconstructViewPosFromLinearZ(f.invProj, uXp, v, depthXp);  // :277-280
    V3 viewPosXm = reconstructViewPosFromLinearZ(f.invProj, uXm, v, depthXm);
    V3 viewPosYp = reconstructViewPosFromLinearZ(f.invProj, u, vYp, depthYp);
    V3 viewPosYm = reconstructViewPosFromLinearZ(f.invProj, u, vYm, depthYm);

    float zXp = -viewPosXp.z, zXm = -viewPosXm.z, zYp = -viewPosYp.z, zYm = -viewPosYm.z;  // :282-285
    float dXp = fabsf(zXp - viewZ), dXm = fabsf(zXm - viewZ);                  // :287-290
    float dYp = fabsf(zYp - viewZ), dYm = fabsf(zYm - viewZ);
    float zden = gmax(viewZ, 1e-3f);
    float rXp = dXp / zden, rXm = dXm / zden, rYp = dYp / zden, rYm = dYm / zden;  // :292-295
    bool okXp = (rXp <= DISCONTINUITY_REL_Z) && (dXp <= discontinuityAbsZ);    // :297-300
    bool okXm = (rXm <= DISCONTINUITY_REL_Z) && (dXm <= discontinuityAbsZ);
    bool okYp = (rYp <= DISCONTINUITY_REL_Z) && (dYp <= discontinuityAbsZ);
    bool okYm = (rYm <= DISCONTINUITY_REL_Z) && (dYm <= discontinuityAbsZ);

    V3 viewPosX = viewPos, viewPosY = viewPos;                                 // :302-317
    if (okXp && okXm) viewPosX = (dXp <= dXm) ? viewPosXp : viewPosXm;
    else if (okXp) viewPosX = viewPosXp;
    else if (okXm) viewPosX = viewPosXm;
    if (okYp && okYm) viewPosY = (dYp <= dYm) ? viewPosYp : viewPosYm;
    else if (okYp) viewPosY = viewPosYp;
    else if (okYm) viewPosY = viewPosYm;

    V3 viewDx = viewPosX - viewPos, viewDy = viewPosY - viewPos;               // :319-321
    float footprint = gmax(length3(viewDx), length3(viewDy));

    V3 bestViewN = v3(0, 0, 0);                                                // :326-351
    float bestScore = 1e30f;
    auto consider = [&](bool ok, V3 a, V3 b, float sc) {
        if (!ok) return;
        V3 n = cross3(a, b);
        float n2 = dot3(n, n);
        if (n2 > 1e-14f && sc < bestScore) { bestScore = sc; bestViewN = n * inversesqrt(n2); }
    };
    consider(okXp && okYp, viewPosYp - viewPos, viewPosXp - viewPos, dXp + dYp);   // :328-333
    consider(okXp && okYm, viewPosXp - viewPos, viewPosYm - viewPos, dXp + dYm);   // :334-339
    consider(okXm && okYp, viewPosYp - viewPos, viewPosXm - viewPos, dXm + dYp);   // :340-345
    consider(okXm && okYm, viewPosXm - viewPos, viewPosYm - viewPos, dXm + dYm);   // :346-351

    V3 geomNormal = v3(0, 1, 0);                                               // :353-359
    if (bestScore < 1e29f) geomNormal = normalize3(mul_m3_v3(f.invView, bestViewN));
    V3 camPos = V3{ubo[84], ubo[85], ubo[86]};
    if (dot3(geomNormal, camPos - worldPos) < 0.0f) geomNormal = neg(geomNormal);  // :360-362

    float biasFromFootprint = footprint * 0.05f;                               // :365
    float biasMax = gmax(baseBias, gmin(baseRadius * 0.02f, 0.03f));           // :367
    float bias = gclamp(gmax(baseBias, biasFromFootprint), baseBias, biasMax); // :368
    float radiusScale = gclamp(sqrtf(gmax(footprint, 1e-4f) / 0.03f), 1.0f, 1.6f);  // :371
    float radius = baseRadius * radiusScale;                                   // :372

    V3 origin = offsetRay(worldPos, geomNormal);                               // :375
    V3 tT, tB;
    buildTbn(geomNormal, &tT, &tB);                                            // :376
    float tMin = gmax(1e-4f, bias * 0.25f);                                    // :377

    if (debugMode == 10u) { histCurr[o] = float_to_half(gclamp(bias / 0.05f, 0.0f, 1.0f)); return; }       // :381-385
    if (debugMode == 11u) { histCurr[o] = float_to_half(gclamp(footprint / 0.05f, 0.0f, 1.0f)); return; }  // :386-390

    float frameIndex = ubo[100 + 3];
    float frameTerm = (rayCount >= 4) ? (frameIndex * 131.0f) : 0.0f;          // :395
    float seedBase = ((float)px * 1.0f + (float)py * 57.0f) + frameTerm;       // :396
    float angle = hash11(seedBase) * 6.28318530718f;                           // :397
    float cs, sn;
    det_sincos(angle, &sn, &cs);                                               // :398-399

    float visibility = 0.0f;
    uint32_t mask = 0;
    for (int i = 0; i < rayCount; ++i) {                                       // :403-432
        V2 xi = hammersley((uint32_t)i, (uint32_t)rayCount);
        float ax = xi.x - 0.5f, ay = xi.y - 0.5f;
        // rot = mat2(cs, -sn, sn, cs): columns (cs,-sn),(sn,cs)  :400,405
        float rx = (cs * ax + sn * ay) + 0.5f;
        float ry = (-sn * ax + cs * ay) + 0.5f;
        rx = gclamp(rx, 0.0f, 1.0f);                                           // :406
        ry = gclamp(ry, 0.0f, 1.0f);
        float phi = 6.28318530718f * rx;                                       // :408
        float cosTheta = sqrtf(1.0f - ry), sinTheta = sqrtf(ry);               // :409-410
        float sphi, cphi;
        det_sincos(phi, &sphi, &cphi);
        V3 l = V3{cphi * sinTheta, sphi * sinTheta, cosTheta};                 // :411
        V3 rd = V3{(tT.x * l.x + tB.x * l.y) + geomNormal.x * l.z, (tT.y * l.x + tB.y * l.y) + geomNormal.y * l.z,
                   (tT.z * l.x + tB.z * l.y) + geomNormal.z * l.z};
        V3 rayDir = normalize3(rd);                                            // :412
        float nd = gclamp(dot3(geomNormal, rayDir), 0.0f, 1.0f);               // :418
        float grazing = 1.0f - nd;
        float slopeScaled = gmax(bias * 0.5f, footprint * 0.02f);              // :421
        float dirOffset = slopeScaled * (0.15f + (1.5f * grazing) * grazing);  // :422
        V3 rayOrigin = origin + rayDir * dirOffset;                            // :423
        float tMinRay = tMin;                                                  // :427-429
        tMinRay = gmax(tMinRay, dirOffset * 2.5f);
        tMinRay = gmax(tMinRay, ((radius * 0.03f) * grazing) * grazing);
        int occ = 0;
        visibility += traceAoRay(s, rayOrigin, rayDir, tMinRay, radius, &occ); // :431
        mask |= (uint32_t)occ << i;
    }
    if (hitMask) hitMask[o] = mask;
    float current = visibility / (float)rayCount;                              // :433
    if (debugMode == 12u) { histCurr[o] = float_to_half(gclamp(current, 0.0f, 1.0f)); return; }  // :435-438

    float hu = ((float)px + 0.5f) / (float)W, hv = ((float)py + 0.5f) / (float)H;  // :440
    V2 prevUv = V2{hu, hv};
    bool hasPrev = computePrevUv(ubo + 104, worldPos, &prevUv);                // :442
    float su = hasPrev ? prevUv.x : hu, sv = hasPrev ? prevUv.y : hv;
    float history = tex_linear_r(histPrev, su, sv);                            // :443
    float prevViewZ = tex_nearest_r(depth, su, sv);                            // :444
    if (prevViewZ <= 1e-6f) prevViewZ = viewZ;                                 // :445-447
    float depthDiff = fabsf(prevViewZ - viewZ) / gmax(viewZ, 1e-3f);           // :449
    float temporalAlpha = gclamp(ubo[100 + 1], 0.0f, 0.99f);                   // :450
    float baseThreshold = gmax(ubo[100 + 2], 0.0f);                            // :452
    float noiseTerm = 0.6f / sqrtf((float)rayCount);                           // :453
    float disocclusionThreshold = gclamp(baseThreshold + noiseTerm, 0.0f, 1.5f);  // :454
    float variance = gmax(current * (1.0f - current), 0.0f) / (float)rayCount; // :457
    float sigma = sqrtf(variance);                                             // :458
    float clampK = 2.5f;
    float historyClamped = gclamp(history, current - clampK * sigma, current + clampK * sigma);  // :460
    float delta = fabsf(current - historyClamped);                             // :462
    float rejectDelta = smoothstep(disocclusionThreshold, disocclusionThreshold * 2.0f + 1e-5f, delta);  // :463
    float rejectDepth = smoothstep(0.02f, 0.15f, depthDiff);                   // :466
    float reject = gmax(rejectDelta, rejectDepth);                             // :467
    float historyWeight = (frameIndex < 0.5f) ? 0.0f : (temporalAlpha * (1.0f - reject));  // :468
    float accum = mixf(current, historyClamped, historyWeight);                // :469
    histCurr[o] = float_to_half(accum);                                        // :471
}

// ------------------------------------------------------------------------------------------------
// rtao_atrous.comp
// ------------------------------------------------------------------------------------------------
// :38-46
inline V3 decodeNormalSafe(const uint16_t* normal, int W, int x, int y)
{
    const uint16_t* p = normal + ((size_t)y * W + x) * 4;
    V3 n = V3{half_to_float(p[0]) * 2.0f - 1.0f, half_to_float(p[1]) * 2.0f - 1.0f, half_to_float(p[2]) * 2.0f - 1.0f};
    float l2 = dot3(n, n);
    if (n.x != n.x || n.y != n.y || n.z != n.z || l2 < 1e-6f) return v3(0, 1, 0);
    return n * inversesqrt(l2);
}

// :48-120, one pixel.  All samplers hit exact texel centres: clamp(uv+offset,0,1) == clamp-to-edge texel.
static void atrous_pixel(const float* ubo, int W, int H, int stepIn, const uint16_t* src, const uint16_t* depth,
                         const uint16_t* normal, uint16_t* dst, int px, int py)
{
    size_t o = (size_t)py * W + px;
    float center = half_to_float(src[o]);                                      // :62-69
    float centerZ = gmax(half_to_float(depth[o]), 1e-3f);                      // :70-71
    V3 centerN = decodeNormalSafe(normal, W, px, py);                          // :72
    float sum = 0.0f, weightSum = 0.0f;
    int step = stepIn < 1 ? 1 : stepIn;                                        // :76
    float radius = gmax(ubo[96 + 2], 0.01f);                                   // :100
    float sigmaAbs = gmax(0.01f, radius * 0.05f);                              // :101
    const float sigmaRel = 0.0025f;                                            // :102
    const float invSigmaAbs = 1.0f / sigmaAbs, invSigmaRel = 1.0f / sigmaRel;
    float stepScale = 1.0f + 0.25f * (float)step;                              // :104
    const float invZden = 1.0f / gmax(centerZ, 1e-3f);
    const bool v_div = (g_variant & 1) != 0, v_samp = (g_variant & 2) != 0;
    const Img16 srcImg{src, W, H}, depthImg{depth, W, H};
    const float invSizeX = 1.0f / (float)W, invSizeY = 1.0f / (float)H;       // :55
    const float uvx = ((float)px + 0.5f) * invSizeX, uvy = ((float)py + 0.5f) * invSizeY;   // :56
    if (v_samp) {                                                              // centre through the samplers at uv (:62-72)
        center = tex_linear_r(srcImg, uvx, uvy);
        centerZ = gmax(tex_nearest_r(depthImg, uvx, uvy), 1e-3f);
        centerN = decodeNormalSafe(normal, W, iclamp((int)floorf(uvx * (float)W), 0, W - 1), iclamp((int)floorf(uvy * (float)H), 0, H - 1));
    }
    for (int y = -2; y <= 2; ++y) {                                            // :78-112
        for (int x = -2; x <= 2; ++x) {
            int tx = iclamp(px + x * step, 0, W - 1), ty = iclamp(py + y * step, 0, H - 1);  // :80-81
            size_t to = (size_t)ty * W + tx;
            float sv = half_to_float(src[to]);                                 // :82-89
            if (v_samp) {
                const float tu = gclamp(uvx + ((float)x * (float)step) * invSizeX, 0.0f, 1.0f);   // :80-81
                const float tv = gclamp(uvy + ((float)y * (float)step) * invSizeY, 0.0f, 1.0f);
                sv = tex_linear_r(srcImg, tu, tv);
                tx = iclamp((int)floorf(tu * (float)W), 0, W - 1);             // NEAREST texel of depth / normal at tapUv
                ty = iclamp((int)floorf(tv * (float)H), 0, H - 1);
                to = (size_t)ty * W + tx;
            }
            float spatial = det_exp(-0.5f * (float)(x * x + y * y));           // :90
            float range = det_exp(-fabsf(sv - center) * 8.0f);                 // :91
            float tapZ = gmax(half_to_float(depth[to]), 1e-3f);                // :92-93
            V3 tapN = decodeNormalSafe(normal, W, tx, ty);                     // :94
            float dz = fabsf(tapZ - centerZ);                                  // :98
            // GLSL `/` may be a reciprocal multiply (2.5 ULP); the contract takes that form for the three
            // per-tap quotients, with the reciprocals correctly rounded
            float dzRel = v_div ? dz / gmax(centerZ, 1e-3f) : dz * invZden;    // :99
            float depthNorm = v_div ? gmax(dz / sigmaAbs, dzRel / sigmaRel) : gmax(dz * invSigmaAbs, dzRel * invSigmaRel);   // :103
            depthNorm *= stepScale;                                            // :104
            float depthWeight = det_exp(-depthNorm * depthNorm);               // :105
            float normalWeight = pow32(gclamp(dot3(centerN, tapN), 0.0f, 1.0f));  // :107
            float w = ((spatial * range) * depthWeight) * normalWeight;        // :108
            sum += sv * w;                                                     // :109
            weightSum += w;                                                    // :110
        }
    }
    float filtered = (weightSum > 0.0f) ? (sum / weightSum) : center;          // :114
    dst[o] = float_to_half(filtered);                                          // :115-119
}

// ------------------------------------------------------------------------------------------------
// pbr.frag shadow rays
// ------------------------------------------------------------------------------------------------
// :199-215
inline float TraceShadowRayDirectional(Scene& s, V3 o, V3 d)
{
    return query(s, o, d, 0.001f, 500.0f, Q_SHADOW, false, nullptr) ? 0.0f : 1.0f;
}

static const float kPoisson[16][2] = {  // :244-249 (data)
    {-0.613f, -0.358f}, {0.582f, -0.396f}, {-0.451f, 0.552f}, {0.208f, 0.623f},
    {-0.263f, -0.826f}, {-0.883f, 0.217f}, {0.759f, 0.274f},  {0.352f, -0.721f},
    {-0.713f, -0.045f}, {0.039f, -0.956f}, {-0.189f, 0.291f}, {0.595f, 0.645f},
    {-0.515f, 0.718f},  {0.941f, -0.135f}, {-0.137f, -0.535f}, {0.169f, 0.892f}};

// :217-260
static float ComputeShadowVisibilityDirectional(Scene& s, V3 worldPos, V3 geomNormal, V3 L, float sunAngularRadius,
                                                float sampleCountF)
{
    float dirLen2 = dot3(L, L);
    if (dirLen2 < 1e-8f) return 1.0f;
    V3 rayDir = normalize3(L);
    float bias = 0.002f;
    V3 rayOrigin = worldPos + geomNormal * bias;
    int sampleCount = iclamp((int)(sampleCountF + 0.5f), 1, 16);
    if (sunAngularRadius <= 0.0f || sampleCount <= 1) {
        float vis = TraceShadowRayDirectional(s, rayOrigin, rayDir);
        return vis > 0.5f ? 1.0f : 0.2f;
    }
    V3 up = (fabsf(rayDir.y) < 0.999f) ? v3(0, 1, 0) : v3(0, 0, 1);
    V3 T = normalize3(cross3(up, rayDir));
    V3 B = cross3(rayDir, T);
    float visSum = 0.0f;
    for (int i = 0; i < sampleCount; ++i) {
        float ux = kPoisson[i][0], uy = kPoisson[i][1];
        V3 offset = V3{sunAngularRadius * (ux * T.x + uy * B.x), sunAngularRadius * (ux * T.y + uy * B.y),
                       sunAngularRadius * (ux * T.z + uy * B.z)};
        V3 sampleDir = normalize3(rayDir + offset);
        visSum += TraceShadowRayDirectional(s, rayOrigin, sampleDir);
    }
    float vis = visSum / (float)sampleCount;
    return mixf(0.0f, 1.0f, vis);
}

// :169-196
static float ComputeShadowVisibility(Scene& s, V3 worldPos, V3 geomNormal, V3 lightPos)
{
    V3 toLight = lightPos - worldPos;
    float lightDistance = length3(toLight);
    if (lightDistance <= 0.0001f) return 1.0f;
    V3 rayDir = V3{toLight.x / lightDistance, toLight.y / lightDistance, toLight.z / lightDistance};
    float bias = gmax(0.002f, 0.0005f * lightDistance);
    V3 rayOrigin = worldPos + geomNormal * bias;
    float tMin = 0.001f;
    float tMax = gmax(lightDistance - bias, tMin);
    return query(s, rayOrigin, rayDir, tMin, tMax, Q_SHADOW, false, nullptr) ? 0.0f : 1.0f;
}

}  // namespace

// ================================================================================================
// C API (ctypes).  Struct layouts mirror include/vkr_rt.h so tests can reuse one set of ctypes classes,
// but this file does not include that header: the oracle shares no code with the product.
// ================================================================================================
extern "C" {

struct orc_mesh_desc {
    const void* positions; uint32_t position_stride; uint32_t vertex_count;
    const uint32_t* indices; uint32_t index_count;
    const void* uvs; uint32_t uv_stride;
    int32_t material_index; uint32_t opaque;
    const void* normals; uint32_t normal_stride;
};
struct orc_instance { uint32_t mesh_index; float transform[12]; };
struct orc_alpha_texture { const uint8_t* alpha; uint32_t width, height; };
struct orc_scene_desc {
    const orc_mesh_desc* meshes; uint32_t mesh_count;
    const orc_instance* instances; uint32_t instance_count;
    const float* material_params;
    const orc_alpha_texture* alpha_textures;
};

void* orc_scene_create(const orc_scene_desc* d, int brute_force)
{
    Scene* s = new Scene();
    s->brute = brute_force != 0;
    uint32_t voff = 0, ioff = 0;
    for (uint32_t m = 0; m < d->mesh_count; ++m) {
        const orc_mesh_desc& md = d->meshes[m];
        int mi = md.material_index;
        s->lut_material.push_back((mi >= 0 && mi < 256) ? (uint32_t)mi : 0u);  // FrameManager.cpp:1280-1283
        s->lut_index_offset.push_back(ioff);
        s->mesh_vertex_offset.push_back(voff);
        s->mesh_index_count.push_back(md.index_count);
        s->mesh_opaque.push_back(md.opaque ? 1u : 0u);
        for (uint32_t v = 0; v < md.vertex_count; ++v) {
            const float* p = (const float*)((const char*)md.positions + (size_t)v * md.position_stride);
            s->positions.push_back(p[0]); s->positions.push_back(p[1]); s->positions.push_back(p[2]);
            if (md.uvs) {
                const float* t = (const float*)((const char*)md.uvs + (size_t)v * md.uv_stride);
                s->uvs.push_back(V2{t[0], t[1]});
            } else {
                s->uvs.push_back(V2{0, 0});
            }
            if (md.normals) {
                const float* n = (const float*)((const char*)md.normals + (size_t)v * md.normal_stride);
                s->normals.push_back(n[0]); s->normals.push_back(n[1]); s->normals.push_back(n[2]);
            } else {
                s->normals.push_back(0.0f); s->normals.push_back(1.0f); s->normals.push_back(0.0f);
            }
        }
        s->mesh_has_normals.push_back(md.normals ? 1u : 0u);
        for (uint32_t i = 0; i < md.index_count; ++i) s->indices.push_back(voff + md.indices[i]);  // :1296-1304
        voff += md.vertex_count;
        ioff += md.index_count;
    }
    for (int i = 0; i < 256; ++i) {
        s->material_params[4 * i + 0] = d->material_params ? d->material_params[4 * i + 0] : 0.5f;
        s->material_params[4 * i + 1] = d->material_params ? d->material_params[4 * i + 1] : 0.0f;
        s->material_params[4 * i + 2] = 0; s->material_params[4 * i + 3] = 0;
        if (d->alpha_textures && d->alpha_textures[i].alpha) {
            s->tex[i].w = d->alpha_textures[i].width; s->tex[i].h = d->alpha_textures[i].height;
            s->tex[i].a.assign(d->alpha_textures[i].alpha,
                               d->alpha_textures[i].alpha + (size_t)s->tex[i].w * s->tex[i].h);
        }
    }
    std::vector<float> xf;
    for (uint32_t i = 0; i < d->instance_count; ++i) {
        s->inst_mesh.push_back(d->instances[i].mesh_index);
        xf.insert(xf.end(), d->instances[i].transform, d->instances[i].transform + 12);
    }
    flatten(*s, xf.data());
    if (!s->brute) build_bvh(*s);
    return s;
}

// RayTracingContext::updateTopLevelAS semantics: same instances, new transforms.
int orc_scene_set_instances(void* sp, const orc_instance* inst, uint32_t count)
{
    Scene* s = (Scene*)sp;
    if (count != s->inst_mesh.size()) return 1;
    std::vector<float> xf;
    for (uint32_t i = 0; i < count; ++i) {
        if (inst[i].mesh_index != s->inst_mesh[i]) return 1;
        xf.insert(xf.end(), inst[i].transform, inst[i].transform + 12);
    }
    flatten(*s, xf.data());
    if (!s->brute) build_bvh(*s);
    return 0;
}

void orc_scene_destroy(void* s) { delete (Scene*)s; }
uint32_t orc_scene_triangle_count(void* s) { return (uint32_t)((Scene*)s)->tris.size(); }

void orc_stats(void* sp, uint64_t out[3], int reset)
{
    Scene* s = (Scene*)sp;
    out[0] = s->n_rays.load(); out[1] = s->n_nodes.load(); out[2] = s->n_tris.load();
    if (reset) { s->n_rays = 0; s->n_nodes = 0; s->n_tris = 0; }
}

void orc_set_variant(int v) { g_variant = v; }
int orc_get_variant() { return g_variant; }

void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// K1 (RtaoComputePass::dispatchTrace, RtaoComputePass.cpp:503-517) on rows [y0,y1).
void orc_rtao_trace(void* sp, const float* ubo, int W, int H, const uint16_t* depth, const uint16_t* histPrev,
                    uint16_t* histCurr, uint32_t* hitMask, int y0, int y1)
{
    Scene& s = *(Scene*)sp;
    Frame f;
    f.W = W; f.H = H; f.ubo = ubo;
    inverse_m4(ubo + 32, f.invProj);  // :245
    inverse_m4(ubo + 16, f.invView);  // :246
    Img16 d{depth, W, H}, hp{histPrev, W, H};
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = y0; y < y1; ++y)
        for (int x = 0; x < W; ++x) trace_pixel(s, f, d, hp, histCurr, hitMask, x, y);
}

// K2, one iteration (RtaoComputePass::dispatchAtrous loop body, RtaoComputePass.cpp:530-537).
void orc_rtao_atrous(const float* ubo, int W, int H, int step, const uint16_t* src, const uint16_t* depth,
                     const uint16_t* normal, uint16_t* dst, int y0, int y1)
{
#pragma omp parallel for schedule(static)
    for (int y = y0; y < y1; ++y)
        for (int x = 0; x < W; ++x) atrous_pixel(ubo, W, H, step, src, depth, normal, dst, x, y);
}

// K3 (rtao_upsample.comp:35-58)
void orc_rtao_upsample(const float* ubo, int W, int H, const uint16_t* src, const uint16_t* normal, uint16_t* dst)
{
    uint32_t debugMode = (uint32_t)(ubo[92 + 3] + 0.5f);  // :46
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t o = (size_t)y * W + x;
            if (debugMode == 9u) {  // :45-52
                const uint16_t* p = normal + o * 4;
                V3 n = normalize3(V3{half_to_float(p[0]) * 2.0f - 1.0f, half_to_float(p[1]) * 2.0f - 1.0f,
                                     half_to_float(p[2]) * 2.0f - 1.0f});
                dst[o] = float_to_half(gclamp(n.y * 0.5f + 0.5f, 0.0f, 1.0f));
            } else {
                dst[o] = src[o];  // :55-57
            }
        }
}

// Whole pass (RtaoComputePass::render, RtaoComputePass.cpp:102-156).  hist[2], ping[2] are caller-owned
// W*H R16F images; curr = frame-in-flight index (FrameManager.cpp:1155-1165).
void orc_rtao_execute(void* sp, const float* ubo, int W, int H, uint32_t iterations, const uint16_t* depth,
                      const uint16_t* normal, uint16_t* hist0, uint16_t* hist1, uint16_t* ping0, uint16_t* ping1,
                      uint32_t curr, uint16_t* rtao_full, uint32_t* hitMask)
{
    uint16_t* hist[2] = {hist0, hist1};
    uint16_t* ping[2] = {ping0, ping1};
    orc_rtao_trace(sp, ubo, W, H, depth, hist[curr ^ 1u], hist[curr], hitMask, 0, H);
    uint32_t n = iterations < 1 ? 1 : iterations;  // :529
    for (uint32_t i = 0; i < n; ++i) {
        const uint16_t* src = (i == 0) ? hist[curr] : ping[(i - 1) & 1u];  // rtao_atrous.comp:57-69
        orc_rtao_atrous(ubo, W, H, 1 << i, src, depth, normal, ping[i & 1u], 0, H);
    }
    orc_rtao_upsample(ubo, W, H, ping[(n - 1) & 1u], normal, rtao_full);  // RtaoComputePass.cpp:560-561
}

void orc_shadow_sun(void* sp, const float* worldPos, const float* geomNormal, const uint8_t* valid, uint32_t count,
                    const float L[3], float radius, float samples, float* out)
{
    Scene& s = *(Scene*)sp;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)count; ++i) {
        if (valid && !valid[i]) { out[i] = 1.0f; continue; }
        out[i] = ComputeShadowVisibilityDirectional(s, V3{worldPos[3 * i], worldPos[3 * i + 1], worldPos[3 * i + 2]},
                                                    V3{geomNormal[3 * i], geomNormal[3 * i + 1], geomNormal[3 * i + 2]},
                                                    V3{L[0], L[1], L[2]}, radius, samples);
    }
}

void orc_shadow_point(void* sp, const float* worldPos, const float* geomNormal, const uint8_t* valid, uint32_t count,
                      const float P[3], float* out)
{
    Scene& s = *(Scene*)sp;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)count; ++i) {
        if (valid && !valid[i]) { out[i] = 1.0f; continue; }
        out[i] = ComputeShadowVisibility(s, V3{worldPos[3 * i], worldPos[3 * i + 1], worldPos[3 * i + 2]},
                                         V3{geomNormal[3 * i], geomNormal[3 * i + 1], geomNormal[3 * i + 2]},
                                         V3{P[0], P[1], P[2]});
    }
}

// G-buffer stand-in for DepthPrepass (depth_prepass.vert:42-52, depth_only.frag:20-35): one primary ray per
// pixel centre; linear view-Z = -(view*worldPos).z -> R16F; normal = geometric normal*0.5+0.5 -> RGBA16F.
// transpose(inverse(mat3(model))) up to a positive factor (depth_prepass.vert:47; the result is normalised anyway):
// cofactors in double from the instance's 3x4 row-major transform, sign of the determinant applied, one rounding.
static void normal_matrix(const float* xf, float* N)
{
    const double m[9] = {xf[0], xf[1], xf[2], xf[4], xf[5], xf[6], xf[8], xf[9], xf[10]};
    double c[9];
    c[0] = m[4] * m[8] - m[5] * m[7]; c[1] = m[5] * m[6] - m[3] * m[8]; c[2] = m[3] * m[7] - m[4] * m[6];
    c[3] = m[2] * m[7] - m[1] * m[8]; c[4] = m[0] * m[8] - m[2] * m[6]; c[5] = m[1] * m[6] - m[0] * m[7];
    c[6] = m[1] * m[5] - m[2] * m[4]; c[7] = m[2] * m[3] - m[0] * m[5]; c[8] = m[0] * m[4] - m[1] * m[3];
    const double det = (m[0] * c[0] + m[1] * c[1]) + m[2] * c[2];
    const double sg = det < 0.0 ? -1.0 : 1.0;
    for (int k = 0; k < 9; ++k) N[k] = (float)(c[k] * sg);
}
static V3 world_normal(const float* N, V3 n)
{
    return normalize3(V3{(N[0] * n.x + N[1] * n.y) + N[2] * n.z, (N[3] * n.x + N[4] * n.y) + N[5] * n.z,
                         (N[6] * n.x + N[7] * n.y) + N[8] * n.z});
}

void orc_gbuffer(void* sp, const float* ubo, int W, int H, uint16_t* depth, uint16_t* normal, float* worldPosOut,
                 float* ngOut, uint8_t* validOut)
{
    Scene& s = *(Scene*)sp;
    float invProj[16], invView[16];
    inverse_m4(ubo + 32, invProj);
    inverse_m4(ubo + 16, invView);
    V3 cam = V3{ubo[84], ubo[85], ubo[86]};
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t o = (size_t)y * W + x;
            float u = ((float)x + 0.5f) / (float)W, v = ((float)y + 0.5f) / (float)H;
            float fv[4];
            mul_m4_v4(invProj, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 1.0f, 1.0f, fv);
            V3 dv = normalize3(V3{fv[0] / fv[3], fv[1] / fv[3], fv[2] / fv[3]});
            V3 d = normalize3(mul_m3_v3(invView, dv));
            Hit h;
            bool hit = query(s, cam, d, 0.0f, 1.0e30f, Q_AO, true, &h);
            float z = 0.0f;
            V3 n = v3(0, 0, 1), wp = v3(0, 0, 0), ng = v3(0, 1, 0);
            if (hit) {
                wp = cam + d * h.t;
                float vp[4];
                mul_m4_v4(ubo + 16, wp.x, wp.y, wp.z, 1.0f, vp);
                z = -vp[2];
                const Tri& tr = s.tris[h.tri];
                const V3 nface = normalize3(cross3(tr.e1, tr.e2));
                n = nface;
                if (s.mesh_has_normals[tr.mesh]) {
                    // depth_prepass.vert:47-48 per vertex, rasteriser interpolation, depth_only.frag:31
                    float N[9];
                    normal_matrix(&s.inst_xf[12 * (size_t)tr.inst], N);
                    const uint32_t ioff = s.lut_index_offset[tr.mesh] + 3 * tr.prim;
                    V3 vn[3];
                    for (int k = 0; k < 3; ++k) {
                        const uint32_t gi = s.indices[ioff + k];
                        vn[k] = world_normal(N, V3{s.normals[3 * gi], s.normals[3 * gi + 1], s.normals[3 * gi + 2]});
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
                uint16_t* p = normal + o * 4;
                if (valid) {
                    p[0] = float_to_half(n.x * 0.5f + 0.5f); p[1] = float_to_half(n.y * 0.5f + 0.5f);
                    p[2] = float_to_half(n.z * 0.5f + 0.5f); p[3] = float_to_half(1.0f);
                } else {  // clear value (0.5,0.5,1,1), DepthPrepass.cpp:36
                    p[0] = float_to_half(0.5f); p[1] = float_to_half(0.5f); p[2] = float_to_half(1.0f); p[3] = float_to_half(1.0f);
                }
            }
            if (worldPosOut) { worldPosOut[3 * o] = wp.x; worldPosOut[3 * o + 1] = wp.y; worldPosOut[3 * o + 2] = wp.z; }
            if (ngOut) { ngOut[3 * o] = ng.x; ngOut[3 * o + 1] = ng.y; ngOut[3 * o + 2] = ng.z; }
            if (validOut) validOut[o] = valid ? 1 : 0;
        }
}

// DepthPrepass with MSAA (DepthPrepass.cpp:31-65: both colour attachments resolve with eAverage; the pipeline has no
// sample shading, DepthPrepassPipeline.cpp:96).  The reference takes the device's maximum sample count, the driver's
// sample positions and its resolve arithmetic; this restatement fixes: `samples` in {1,2,4,8} at the Vulkan standard
// sample locations; coverage and depth test per sample (a ray through the sample position, closest non-discarded
// primitive); the fragment's outputs evaluated ONCE per primitive at the pixel centre (attributes extrapolated on the
// triangle's plane when the centre is not covered); each sample stores them as fp16; the resolve is the float32 mean
// in sample order, rounded to fp16.  Uncovered samples hold the clear values (depth 0, normal (0.5,0.5,1,1)).
// world position / Ng / valid keep the pixel-centre semantics of orc_gbuffer (they stand for pbr.frag's varyings).
static const float kSamplePos[4][8][2] = {
    {{0.5f, 0.5f}},
    {{0.75f, 0.75f}, {0.25f, 0.25f}},
    {{0.375f, 0.125f}, {0.875f, 0.375f}, {0.125f, 0.625f}, {0.625f, 0.875f}},
    {{0.5625f, 0.3125f}, {0.4375f, 0.6875f}, {0.8125f, 0.5625f}, {0.3125f, 0.1875f},
     {0.1875f, 0.8125f}, {0.0625f, 0.4375f}, {0.6875f, 0.9375f}, {0.9375f, 0.0625f}}};

static V3 primary_dir(const float* invProj, const float* invView, float fx, float fy, int W, int H)
{
    float u = fx / (float)W, v = fy / (float)H;
    float fv[4];
    mul_m4_v4(invProj, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 1.0f, 1.0f, fv);
    V3 dv = normalize3(V3{fv[0] / fv[3], fv[1] / fv[3], fv[2] / fv[3]});
    return normalize3(mul_m3_v3(invView, dv));
}

// the fragment outputs of triangle `ti` for the pixel whose centre ray is (cam, dc): linear view Z and encoded normal
static void fragment_outputs(const Scene& s, const float* view, V3 cam, V3 dc, uint32_t ti, float t_s, float u_s, float v_s,
                             V3 d_s, float* z_out, V3* n_out)
{
    const Tri& tr = s.tris[ti];
    float t = t_s, u = u_s, v = v_s;
    V3 dir = d_s;
    V3 pc = cross3(dc, tr.e2);
    float detc = dot3(tr.e1, pc);
    if (detc != 0.0f) {                                   // centre ray on the triangle's plane (else: the sample's own hit)
        float invc = 1.0f / detc;
        V3 tvec = cam - tr.v0;
        V3 qvec = cross3(tvec, tr.e1);
        u = dot3(tvec, pc) * invc;
        v = dot3(dc, qvec) * invc;
        t = dot3(tr.e2, qvec) * invc;
        dir = dc;
    }
    V3 wp = cam + dir * t;
    float vp[4];
    mul_m4_v4(view, wp.x, wp.y, wp.z, 1.0f, vp);
    *z_out = -vp[2];
    V3 n = normalize3(cross3(tr.e1, tr.e2));
    if (s.mesh_has_normals[tr.mesh]) {
        float N[9];
        normal_matrix(&s.inst_xf[12 * (size_t)tr.inst], N);
        const uint32_t ioff = s.lut_index_offset[tr.mesh] + 3 * tr.prim;
        V3 vn[3];
        for (int k = 0; k < 3; ++k) {
            const uint32_t gi = s.indices[ioff + k];
            vn[k] = world_normal(N, V3{s.normals[3 * gi], s.normals[3 * gi + 1], s.normals[3 * gi + 2]});
        }
        const float w0 = (1.0f - u) - v;
        n = normalize3(V3{(vn[0].x * w0 + vn[1].x * u) + vn[2].x * v, (vn[0].y * w0 + vn[1].y * u) + vn[2].y * v,
                          (vn[0].z * w0 + vn[1].z * u) + vn[2].z * v});
    }
    *n_out = n;
}

void orc_gbuffer_msaa(void* sp, const float* ubo, int W, int H, int samples, uint16_t* depth, uint16_t* normal,
                      float* worldPosOut, float* ngOut, uint8_t* validOut)
{
    Scene& s = *(Scene*)sp;
    const int set = samples == 8 ? 3 : (samples == 4 ? 2 : (samples == 2 ? 1 : 0));
    const int S = 1 << set;
    float invProj[16], invView[16];
    inverse_m4(ubo + 32, invProj);
    inverse_m4(ubo + 16, invView);
    V3 cam = V3{ubo[84], ubo[85], ubo[86]};
    // centre-semantics outputs first (and the S == 1 images)
    orc_gbuffer(sp, ubo, W, H, S == 1 ? depth : nullptr, S == 1 ? normal : nullptr, worldPosOut, ngOut, S == 1 ? validOut : nullptr);
    if (S == 1) return;
    const float invS = 1.0f / (float)S;
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t o = (size_t)y * W + x;
            V3 dc = primary_dir(invProj, invView, (float)x + 0.5f, (float)y + 0.5f, W, H);
            float zsum = 0.0f, nsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int k = 0; k < S; ++k) {
                V3 ds = primary_dir(invProj, invView, (float)x + kSamplePos[set][k][0], (float)y + kSamplePos[set][k][1], W, H);
                Hit h;
                uint16_t zh = 0, nh[4] = {float_to_half(0.5f), float_to_half(0.5f), float_to_half(1.0f), float_to_half(1.0f)};
                if (query(s, cam, ds, 0.0f, 1.0e30f, Q_AO, true, &h, &dc)) {
                    float z;
                    V3 n;
                    fragment_outputs(s, ubo + 16, cam, dc, h.tri, h.t, h.u, h.v, ds, &z, &n);
                    zh = float_to_half(z);
                    nh[0] = float_to_half(n.x * 0.5f + 0.5f); nh[1] = float_to_half(n.y * 0.5f + 0.5f);
                    nh[2] = float_to_half(n.z * 0.5f + 0.5f); nh[3] = float_to_half(1.0f);
                }
                zsum = zsum + half_to_float(zh);
                for (int c = 0; c < 4; ++c) nsum[c] = nsum[c] + half_to_float(nh[c]);
            }
            uint16_t zr = float_to_half(zsum * invS);
            if (depth) depth[o] = zr;
            if (normal) for (int c = 0; c < 4; ++c) normal[o * 4 + c] = float_to_half(nsum[c] * invS);
            if (validOut) validOut[o] = half_to_float(zr) > 1e-6f ? 1 : 0;
        }
}

// ---- unit taps for known-answer tests -----------------------------------------------------------
// Renderer::rebuildRayTracingInstances (Renderer.cpp:503-527): DFS over the node hierarchy, worldFromNode =
// parentWorld * local (GLM scalar mat4 product: column j = ((A0*b0 + A1*b1) + A2*b2) + A3*b3, every product and sum
// rounded), one instance per (node, mesh index) in DFS order, then toVkTransformMatrix (RayTracingContext.cpp:28-44):
// 3x4 row-major of the column-major mat4.  nodes: parent index (-1 = root, parents first), local[16], instance range.
struct orc_node { int32_t parent; float local[16]; uint32_t first_instance, instance_count; };
static void glm_mul_m4(const float* A, const float* B, float* R)
{
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r) {
            float acc = A[0 * 4 + r] * B[j * 4 + 0];
            acc = acc + A[1 * 4 + r] * B[j * 4 + 1];
            acc = acc + A[2 * 4 + r] * B[j * 4 + 2];
            acc = acc + A[3 * 4 + r] * B[j * 4 + 3];
            R[j * 4 + r] = acc;
        }
}
void orc_instances_from_nodes(const orc_node* nodes, uint32_t node_count, const float* model, float* out_xf)
{
    std::vector<float> world((size_t)node_count * 16);
    for (uint32_t n = 0; n < node_count; ++n) {      // parents precede children: the DFS recurrence in array order
        const float* parentWorld = nodes[n].parent < 0 ? model : &world[(size_t)nodes[n].parent * 16];
        glm_mul_m4(parentWorld, nodes[n].local, &world[(size_t)n * 16]);
        const float* m = &world[(size_t)n * 16];
        for (uint32_t k = 0; k < nodes[n].instance_count; ++k) {
            float* o = out_xf + (size_t)(nodes[n].first_instance + k) * 12;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) o[r * 4 + c] = m[c * 4 + r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GPU-driven indirect build.  FrameManager::buildSharedDrawSlots (FrameManager.cpp:81-116) and
// FrameManager::prepareSharedOpaqueIndirect (:118-175) restated with the same containers (std::stable_sort,
// std::map<pair<bool,uint32_t>, vector>), plus the documented next step of docs/GPU_DRIVEN_RENDERING_DESIGN.md --
// a frustum test of the mesh's transformed AABB (Frustum.h:18-60, BoundingBox.h:14-39) -- as an optional filter.
// ---------------------------------------------------------------------------------------------------------------
struct orc_draw_slot { uint32_t node_linear_index, mesh_index, mat_index, double_sided; };
struct orc_mesh_draw_info { uint32_t vertex_offset, first_index, index_count; };      // GlobalMeshBuffer.h:12-16
struct orc_draw_cmd { uint32_t index_count, instance_count, first_index; int32_t vertex_offset; uint32_t first_instance; };
struct orc_bucket_span { uint32_t double_sided, mat_index, first_command, draw_count; };

// the sort of buildSharedDrawSlots (:108-115)
void orc_sort_draw_slots(orc_draw_slot* slots, uint32_t count)
{
    std::stable_sort(slots, slots + count, [](const orc_draw_slot& a, const orc_draw_slot& b) {
        if ((a.double_sided != 0) != (b.double_sided != 0)) return (a.double_sided != 0) < (b.double_sided != 0);
        if (a.mat_index != b.mat_index) return a.mat_index < b.mat_index;
        return a.mesh_index < b.mesh_index;
    });
}

// Frustum::extractPlanes (Frustum.h:46-60); m = GLM column-major view-projection
void orc_frustum_planes(const float* m, float* planes /* 6 x 4 */)
{
    auto M = [&](int c, int r) { return m[c * 4 + r]; };
    const int row[6] = {0, 0, 1, 1, 2, 2};
    const float sg[6] = {1.0f, -1.0f, 1.0f, -1.0f, 1.0f, -1.0f};
    for (int p = 0; p < 6; ++p) {
        float* o = planes + 4 * p;
        for (int c = 0; c < 4; ++c) o[c] = (sg[p] > 0.0f) ? M(c, 3) + M(c, row[p]) : M(c, 3) - M(c, row[p]);
        float len = sqrtf((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]);      // glm::length(vec3)
        if (len > 1e-6f) { o[0] /= len; o[1] /= len; o[2] /= len; o[3] /= len; }
    }
}

// BoundingBox::Transform (BoundingBox.h:14-39) then Frustum::Intersects (Frustum.h:18-44)
static bool box_in_frustum(const float* world, const float* bmin, const float* bmax, const float* planes)
{
    float nmin[3] = {1e30f, 1e30f, 1e30f}, nmax[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = 0; i < 8; ++i) {
        const float cx = (i & 1) ? bmax[0] : bmin[0], cy = (i & 2) ? bmax[1] : bmin[1], cz = (i & 4) ? bmax[2] : bmin[2];
        for (int r = 0; r < 3; ++r) {
            // glm mat4 * vec4: (m[0]*x + m[1]*y) + (m[2]*z + m[3]*w), w = 1
            const float p = (world[0 * 4 + r] * cx + world[1 * 4 + r] * cy) + (world[2 * 4 + r] * cz + world[3 * 4 + r] * 1.0f);
            nmin[r] = p < nmin[r] ? p : nmin[r];            // glm::min(newMin, p): (y < x) ? y : x
            nmax[r] = nmax[r] < p ? p : nmax[r];            // glm::max(newMax, p): (x < y) ? y : x
        }
    }
    for (int k = 0; k < 6; ++k) {
        const float* pl = planes + 4 * k;
        const float vx = pl[0] >= 0.0f ? nmax[0] : nmin[0], vy = pl[1] >= 0.0f ? nmax[1] : nmin[1], vz = pl[2] >= 0.0f ? nmax[2] : nmin[2];
        if (((pl[0] * vx + pl[1] * vy) + pl[2] * vz) + pl[3] < 0.0f) return false;
    }
    return true;
}

// prepareSharedOpaqueIndirect (:118-175).  nodes: linear nodes (parent linear index or -1, local matrix); slots already
// sorted; view_proj / mesh_bounds (6 floats per mesh) may be NULL (= the reference as it is: no culling).
// Outputs sized max_draws (draw_data 16 floats each, commands) and span_capacity; returns the draw count,
// *span_count = number of bucket spans.
uint32_t orc_prepare_indirect(const int32_t* node_parent, const float* node_local, uint32_t node_count,
                              const orc_draw_slot* slots, uint32_t slot_count, const orc_mesh_draw_info* infos,
                              uint32_t mesh_count, const float* model, uint32_t max_draws_in, const float* view_proj,
                              const float* mesh_bounds, float* draw_data, orc_draw_cmd* commands, orc_bucket_span* spans,
                              uint32_t span_capacity, uint32_t* span_count)
{
    const uint32_t maxDraws = max_draws_in < 1u ? 1u : max_draws_in;            // FrameManager.cpp:24
    std::vector<float> world((size_t)node_count * 16);
    for (uint32_t n = 0; n < node_count; ++n) {                                  // :121-126
        const float* parentWorld = node_parent[n] >= 0 ? &world[(size_t)node_parent[n] * 16] : model;
        glm_mul_m4(parentWorld, node_local + (size_t)n * 16, &world[(size_t)n * 16]);
    }
    float planes[24];
    if (view_proj) orc_frustum_planes(view_proj, planes);
    std::map<std::pair<bool, uint32_t>, std::vector<orc_draw_cmd>> buckets;      // :135-136
    uint32_t drawId = 0;
    for (uint32_t i = 0; i < slot_count; ++i) {                                  // :138-152
        const orc_draw_slot& slot = slots[i];
        if (drawId >= maxDraws) break;
        if (slot.node_linear_index >= node_count || slot.mesh_index >= mesh_count) continue;
        if (view_proj && mesh_bounds &&
            !box_in_frustum(&world[(size_t)slot.node_linear_index * 16], mesh_bounds + 6 * (size_t)slot.mesh_index,
                            mesh_bounds + 6 * (size_t)slot.mesh_index + 3, planes))
            continue;
        memcpy(draw_data + (size_t)drawId * 16, &world[(size_t)slot.node_linear_index * 16], 64);
        const orc_mesh_draw_info& info = infos[slot.mesh_index];
        orc_draw_cmd cmd;
        cmd.index_count = info.index_count; cmd.instance_count = 1; cmd.first_index = info.first_index;
        cmd.vertex_offset = (int32_t)info.vertex_offset; cmd.first_instance = drawId;
        buckets[{slot.double_sided != 0, slot.mat_index}].push_back(cmd);
        ++drawId;
    }
    size_t indirectOffset = 0;
    uint32_t ns = 0;
    for (const auto& kv : buckets) {                                             // :154-172
        const auto& cmds = kv.second;
        if (cmds.empty()) continue;
        const size_t canCopy = (indirectOffset < maxDraws) ? std::min(cmds.size(), (size_t)maxDraws - indirectOffset) : 0u;
        if (canCopy > 0) memcpy(commands + indirectOffset, cmds.data(), canCopy * sizeof(orc_draw_cmd));
        orc_bucket_span span;
        span.double_sided = kv.first.first ? 1u : 0u; span.mat_index = kv.first.second;
        span.first_command = (uint32_t)indirectOffset;
        const uint32_t available = (span.first_command < maxDraws) ? (maxDraws - span.first_command) : 0u;
        span.draw_count = (uint32_t)std::min(cmds.size(), (size_t)available);
        if (span.draw_count > 0) { if (ns < span_capacity) spans[ns] = span; ++ns; }
        indirectOffset += cmds.size();
    }
    *span_count = ns;
    return drawId;                                                               // :173
}

float orc_radical_inverse(uint32_t i) { return radicalInverseVdC(i); }
void orc_hammersley(uint32_t i, uint32_t n, float out[2]) { V2 h = hammersley(i, n); out[0] = h.x; out[1] = h.y; }
void orc_offset_ray(const float p[3], const float n[3], float out[3])
{
    V3 r = offsetRay(V3{p[0], p[1], p[2]}, V3{n[0], n[1], n[2]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_build_tbn(const float n[3], float t[3], float b[3])
{
    V3 T, B;
    buildTbn(V3{n[0], n[1], n[2]}, &T, &B);
    t[0] = T.x; t[1] = T.y; t[2] = T.z; b[0] = B.x; b[1] = B.y; b[2] = B.z;
}
float orc_sin(float x) { return det_sin(x); }
float orc_cos(float x) { float s, c; det_sincos(x, &s, &c); return c; }
float orc_exp(float x) { return det_exp(x); }
float orc_hash11(float x) { return hash11(x); }
float orc_smoothstep(float a, float b, float x) { return smoothstep(a, b, x); }
void orc_inverse_m4(const float* m, float* out) { inverse_m4(m, out); }
float orc_trace_ao_ray(void* sp, const float o[3], const float d[3], float tMin, float tMax, int* occluded)
{
    return traceAoRay(*(Scene*)sp, V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, tMin, tMax, occluded);
}
int orc_query_ao(void* sp, const float o[3], const float d[3], float tMin, float tMax, float* tHit)
{
    return queryAoClosestHit(*(Scene*)sp, V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, tMin, tMax, tHit) ? 1 : 0;
}
float orc_shadow_ray(void* sp, const float o[3], const float d[3])
{
    return TraceShadowRayDirectional(*(Scene*)sp, V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]});
}
float orc_tex_linear(const uint16_t* img, int W, int H, float u, float v) { return tex_linear_r(Img16{img, W, H}, u, v); }
float orc_tex_nearest(const uint16_t* img, int W, int H, float u, float v) { return tex_nearest_r(Img16{img, W, H}, u, v); }

}  // extern "C"
