// rcp_exact.hip -- exhaustive check on gfx950: for which float inputs does v_rcp_f32 + one fma Newton step give the
// correctly rounded reciprocal, i.e. the bits of the compiler's IEEE `1.0f / x` (v_div_scale / v_rcp / 4 x v_fma / v_div_fmas /
// v_div_fixup, ~11 instructions)?  Prints the mismatch count per biased exponent of x.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/calib/rcp_exact.hip -o tools/calib/rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k(unsigned long long* bad_by_exp, unsigned int hi)
{
    const unsigned int bits = (hi << 24) | (blockIdx.x * 256u + threadIdx.x);
    const float x = __builtin_bit_cast(float, bits);
    const float want = 1.0f / x;
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    const float got = __builtin_fmaf(e, r, r);
    const unsigned int w = __builtin_bit_cast(unsigned int, want), g = __builtin_bit_cast(unsigned int, got);
    const bool nan_both = (want != want) && (got != got);
    if (w != g && !nan_both) atomicAdd(&bad_by_exp[(bits >> 23) & 0xffu], 1ull);
}

int main()
{
    unsigned long long* d; CK(hipMalloc(&d, 256 * 8)); CK(hipMemset(d, 0, 256 * 8));
    for (unsigned int hi = 0; hi < 256; ++hi) k<<<65536, 256>>>(d, hi);
    CK(hipDeviceSynchronize());
    unsigned long long h[256]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long total = 0; int lo = -1, hiexp = -1;
    printf("{\"mismatches_by_biased_exponent\": {");
    bool first = true;
    for (int e = 0; e < 256; ++e) if (h[e]) { printf("%s\"%d\": %llu", first ? "" : ", ", e, h[e]); first = false; total += h[e]; }
    for (int e = 1; e < 255; ++e) if (!h[e]) { if (lo < 0) lo = e; hiexp = e; }
    printf("}, \"total\": %llu, \"clean_exponents_between\": [%d, %d]}\n", total, lo, hiexp);
    return 0;
}
