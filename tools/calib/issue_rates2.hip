// issue_rates2.hip -- second calibration set: which VALU opcodes issue at the fast (fma) rate on gfx950.
// Same method as issue_rates.hip (16 independent instructions per loop iteration, 65536 iterations, 8 waves per SIMD
// requested); reports chip-wide wave-instructions per ns from the kernel's wall time.
// build: hipcc --offload-arch=gfx950 -O2 tools/calib/issue_rates2.hip -o tools/calib/issue_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int LOOPS = 32768;

#define R8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                     OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
#define KERNEL(NAME, OP, TAIL)                                                                                        \
    __global__ __launch_bounds__(64) void NAME(float* sink, int loops)                                                \
    {                                                                                                                 \
        float a0 = threadIdx.x * 1.0f, a1 = 1.5f, a2 = 2.5f, a3 = 3.5f, a4 = 4.5f, a5 = 5.5f, a6 = 6.5f, a7 = 7.5f;   \
        float b = 1.0000001f, c = 1e-9f;                                                                              \
        for (int i = 0; i < loops; ++i)                                                                               \
            asm volatile(R8(OP, TAIL) R8(OP, TAIL)                                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(b), "v"(c));                                                                           \
        float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                              \
        if (r == 123.456f) sink[0] = r;                                                                               \
    }
KERNEL(k_add_f32, "v_add_f32", ", %8")
KERNEL(k_mul_f32, "v_mul_f32", ", %8")
KERNEL(k_sub_f32, "v_sub_f32", ", %8")
KERNEL(k_fmac_f32, "v_fmac_f32", ", %8")
KERNEL(k_max_f32, "v_max_f32", ", %8")
KERNEL(k_min_f32, "v_min_f32", ", %8")
KERNEL(k_min3_f32, "v_min3_f32", ", %8, %9")
KERNEL(k_med3_f32, "v_med3_f32", ", %8, %9")
KERNEL(k_add_u32, "v_add_u32", ", %8")
KERNEL(k_and_b32, "v_and_b32", ", %8")
KERNEL(k_or_b32, "v_or_b32", ", %8")
KERNEL(k_lshlrev_b32, "v_lshlrev_b32", ", 3")
KERNEL(k_lshl_add_u32, "v_lshl_add_u32", ", 2, %8")
KERNEL(k_bfe_u32, "v_bfe_u32", ", 8, 8")
KERNEL(k_mov_b32, "v_mov_b32", "")
KERNEL(k_perm_b32, "v_perm_b32", ", %8, %9")
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32", "")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24", ", %8, %9")
KERNEL(k_add3_u32, "v_add3_u32", ", %8, %9")
KERNEL(k_fma_f32, "v_fma_f32", ", %8, %9")
KERNEL(k_mul_legacy, "v_mul_legacy_f32", ", %8")
KERNEL(k_cvt_pk_f16, "v_cvt_pkrtz_f16_f32", ", %8")
KERNEL(k_fma_mix, "v_fma_mix_f32", ", %8, %9")
KERNEL(k_exp_f32, "v_exp_f32", "")
// round 3: the opcodes of K2's software exp that the first sets did not cover (literal-constant fmas, ldexp, rndne, cvt to int)
KERNEL(k_fmaak_f32, "v_fmaak_f32", ", %8, 0x3f000001")
KERNEL(k_fmamk_f32, "v_fmamk_f32", ", 0x3f000001, %8")
KERNEL(k_ldexp_f32, "v_ldexp_f32", ", 0")
KERNEL(k_rndne_f32, "v_rndne_f32", "")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32", "")

typedef void (*kfn)(float*, int);
static void run(const char* name, kfn k, float* sink)
{
    const int blocks = 1024 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, sink, LOOPS);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, sink, LOOPS);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_ns = (double)blocks * 16.0 * LOOPS / (ms * 1e6);
    printf("{\"kind\": \"%s\", \"blocks\": %d, \"wall_us\": %.1f, \"wall_inst_per_ns_chip\": %.1f, \"cyc_per_inst_per_simd_at_2p4ghz\": %.2f}\n",
           name, blocks, ms * 1e3, per_ns, 1024.0 * 2.4 / per_ns);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}
#define RUN(N) run(#N, N, sink)
int main()
{
    float* sink;
    CK(hipMalloc(&sink, 16));
    RUN(k_fma_f32); RUN(k_add_f32); RUN(k_mul_f32); RUN(k_sub_f32); RUN(k_fmac_f32); RUN(k_max_f32); RUN(k_min_f32);
    RUN(k_min3_f32); RUN(k_med3_f32); RUN(k_add_u32); RUN(k_and_b32); RUN(k_or_b32); RUN(k_lshlrev_b32); RUN(k_lshl_add_u32);
    RUN(k_bfe_u32); RUN(k_mov_b32); RUN(k_perm_b32); RUN(k_cvt_f32_u32); RUN(k_mad_u32_u24); RUN(k_add3_u32);
    RUN(k_mul_legacy); RUN(k_cvt_pk_f16); RUN(k_fma_mix); RUN(k_exp_f32);
    RUN(k_fmaak_f32); RUN(k_fmamk_f32); RUN(k_ldexp_f32); RUN(k_rndne_f32); RUN(k_cvt_i32_f32);
    return 0;
}
