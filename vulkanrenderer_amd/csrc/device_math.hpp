// device_math.hpp -- the numeric contract of the path, device side (gfx950).
//
// GLSL leaves sin/cos/exp/pow/inversesqrt precision and texture-filter arithmetic to the driver.  This
// implementation fixes one definition for each (DESIGN.md "Numeric contract") so results are reproducible
// bit for bit across launches, band splits and GPUs.  The whole translation unit is compiled with
// -ffp-contract=off: a multiply-add is fused only where it is spelled fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vkr {

struct V3 { float x, y, z; };
struct V2 { float x, y; };

#define VKR_DI __device__ __forceinline__

VKR_DI V3 v3(float x, float y, float z) { return V3{x, y, z}; }
VKR_DI V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
VKR_DI V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
VKR_DI V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
VKR_DI V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }

// GLSL min/max/clamp semantics (max: y if x<y else x; min: y if y<x else x), NaN-propagation included.
VKR_DI float gmax(float a, float b) { return (a < b) ? b : a; }
VKR_DI float gmin(float a, float b) { return (b < a) ? b : a; }
VKR_DI float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
VKR_DI int iclamp(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

VKR_DI float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VKR_DI float dot3(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
VKR_DI V3 cross3(V3 a, V3 b)
{
    return V3{fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))};
}
// sqrt and divide are the correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
VKR_DI float sqrt_(float x) { return __builtin_sqrtf(x); }
VKR_DI float inversesqrt(float x) { return 1.0f / sqrt_(x); }
VKR_DI V3 normalize3(V3 a) { return a * inversesqrt(dot3(a, a)); }
VKR_DI float length3(V3 a) { return sqrt_(dot3(a, a)); }

VKR_DI float half_to_float(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// fp32 -> fp16, round to nearest even.  The empty asm pins the fp32 value: without it the compiler may fold a preceding
// fp32 multiply into v_fma_mixlo_f16, which rounds the exact product once -- a different result from "round to fp32,
// then to fp16" on ties (seen in k_ibl_irradiance: 1 texel component in ~10^4).
VKR_DI uint16_t float_to_half(float f)
{
    asm volatile("" : "+v"(f));
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}

VKR_DI uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
VKR_DI float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// sin/cos: range reduction by pi/2 in double (two-term Cody-Waite), cephes minimax polynomials in float.
VKR_DI void det_sincos(float x, float* s_out, float* c_out)
{
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_LO = 6.12323399573676603587e-17;
    double xd = (double)x;
    double kd = __builtin_rint(xd * TWO_OVER_PI);
    double rd = xd - kd * PIO2_HI;
    rd = rd - kd * PIO2_LO;
    float r = (float)rd;
    int q = (int)((long long)kd & 3LL);
    float z = r * r;
    float ps = fma_(z, -1.9515295891E-4f, 8.3321608736E-3f);
    ps = fma_(z, ps, -1.6666654611E-1f);
    float sr = fma_(r * z, ps, r);
    float pc = fma_(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    pc = fma_(z, pc, 4.166664568298827E-002f);
    float cr = fma_(z * z, pc, fma_(z, -0.5f, 1.0f));
    float s = (q & 1) ? cr : sr;
    float c = (q & 1) ? sr : cr;
    // q: 0 -> (s,c)  1 -> (c,-s)  2 -> (-s,-c)  3 -> (-c,s)
    s = (q & 2) ? -s : s;
    c = ((q + 1) & 2) ? -c : c;
    *s_out = s;
    *c_out = c;
}
VKR_DI float det_sin(float x) { float s, c; det_sincos(x, &s, &c); return s; }

// exp for non-positive arguments; exp(x < -87) = 0.
VKR_DI float det_exp(float x)
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
    int ki = (int)k;
    float res = p * u2f((uint32_t)(ki + 127) << 23);
    return (x < -87.0f) ? 0.0f : res;
}
VKR_DI float pow3(float x) { return (x * x) * x; }
VKR_DI float pow32(float x) { x = x * x; x = x * x; x = x * x; x = x * x; x = x * x; return x; }
VKR_DI float smoothstep_(float e0, float e1, float x)
{
    float t = gclamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}
VKR_DI float mixf(float a, float b, float w) { return a * (1.0f - w) + b * w; }
VKR_DI float fractf(float x) { return x - __builtin_floorf(x); }

// column-major mat4 (GLM): m[c*4+r]
VKR_DI void mul_m4_v4(const float* m, float x, float y, float z, float w, float out[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}
VKR_DI V3 mul_m3_v3(const float* m, V3 v)
{
    return V3{(m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
              (m[2] * v.x + m[6] * v.y) + m[10] * v.z};
}

}  // namespace vkr
