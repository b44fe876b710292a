// issue_rates.hip -- calibration of per-SIMD instruction issue rates on gfx950 (MI355X).
//
// Each wave runs LOOPS iterations of a block of 16 independent instructions of one kind and stamps s_memtime around the
// loop.  Launched as 1024 SIMDs x W waves (one 64-thread workgroup per wave), W = 1, 2, 4, 6, 8.  Reported per kind and W:
//   cyc/inst/wave  = median over waves of (stamp delta) / (16 * LOOPS)            -- what one wave sees
//   inst/cyc/SIMD  = W / (cyc/inst/wave)                                            -- aggregate issue rate of a SIMD
// and the same from the kernel's wall time (HIP events) at the clock GRBM reports, as a cross-check.
// Used by DESIGN.md section 4 ("issue-rate calibration") and bench.py's K1 roofline (bound: valu).
//
// build: hipcc --offload-arch=gfx950 -O2 tools/calib/issue_rates.hip -o tools/calib/issue_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int LOOPS = 65536;

enum Kind { K_FMA = 0, K_PKFMA, K_CVTUB, K_MAX3, K_CNDMASK, K_CMP, K_MBCNT, K_RCP, K_SALU_ADD, K_SALU_AND64, K_SALU_BCNT,
            K_MIX_VS, K_MIX_V2S, K_LDS_RD, K_LDS_WR, K_READLANE, K_WRITELANE, K_FMA_DEP, K_COUNT };
static const char* kind_name[K_COUNT] = {
    "v_fma_f32", "v_pk_fma_f32", "v_cvt_f32_ubyte1", "v_max3_f32", "v_cndmask_b32", "v_cmp_le_f32(->sgpr pair)",
    "v_mbcnt_lo+hi (pairs)", "v_rcp_f32", "s_add_u32", "s_and_b64", "s_bcnt1_i32_b64", "mix 1 v_fma : 1 s_add",
    "mix 2 v_fma : 1 s_add", "ds_read_b32", "ds_write_b32", "v_readlane_b32", "v_writelane_b32", "v_fma_f32 dependent chain"};

template <int KIND>
__global__ __launch_bounds__(64) void k_issue(unsigned long long* stamps, float* sink, int loops)
{
    __shared__ float lds[256];
    float a0 = threadIdx.x * 1.0f, a1 = 1.5f, a2 = 2.5f, a3 = 3.5f, a4 = 4.5f, a5 = 5.5f, a6 = 6.5f, a7 = 7.5f;
    float b = 1.0000001f, c = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    unsigned u0 = threadIdx.x * 77u, u1 = 3u;
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    unsigned long long m0 = 0x5555555555555555ull, m1 = 0x3333333333333333ull;
    lds[threadIdx.x] = a0;
    unsigned laddr = threadIdx.x * 4u;
    __syncthreads();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < loops; ++i) {
        if (KIND == K_FMA) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == K_FMA_DEP) {
            asm volatile(
                "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                : "+v"(a0) : "v"(b), "v"(c));
        } else if (KIND == K_PKFMA) {
            asm volatile(
                "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        } else if (KIND == K_CVTUB) {
            asm volatile(
                "v_cvt_f32_ubyte1 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte1 %2, %8\n v_cvt_f32_ubyte1 %3, %8\n"
                "v_cvt_f32_ubyte1 %4, %8\n v_cvt_f32_ubyte1 %5, %8\n v_cvt_f32_ubyte1 %6, %8\n v_cvt_f32_ubyte1 %7, %8\n"
                "v_cvt_f32_ubyte2 %0, %8\n v_cvt_f32_ubyte2 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte2 %3, %8\n"
                "v_cvt_f32_ubyte2 %4, %8\n v_cvt_f32_ubyte2 %5, %8\n v_cvt_f32_ubyte2 %6, %8\n v_cvt_f32_ubyte2 %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0));
        } else if (KIND == K_MAX3) {
            asm volatile(
                "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == K_CNDMASK) {
            asm volatile(
                "v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                "v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m0));
        } else if (KIND == K_CMP) {
            asm volatile(
                "v_cmp_le_f32 %0, %2, %3\n v_cmp_le_f32 %1, %3, %2\n v_cmp_le_f32 %0, %2, %4\n v_cmp_le_f32 %1, %4, %2\n"
                "v_cmp_le_f32 %0, %2, %3\n v_cmp_le_f32 %1, %3, %2\n v_cmp_le_f32 %0, %2, %4\n v_cmp_le_f32 %1, %4, %2\n"
                "v_cmp_le_f32 %0, %2, %3\n v_cmp_le_f32 %1, %3, %2\n v_cmp_le_f32 %0, %2, %4\n v_cmp_le_f32 %1, %4, %2\n"
                "v_cmp_le_f32 %0, %2, %3\n v_cmp_le_f32 %1, %3, %2\n v_cmp_le_f32 %0, %2, %4\n v_cmp_le_f32 %1, %4, %2\n"
                : "+s"(m0), "+s"(m1) : "v"(a0), "v"(b), "v"(c));
        } else if (KIND == K_MBCNT) {
            asm volatile(
                "v_mbcnt_lo_u32_b32 %0, %4, 0\n v_mbcnt_hi_u32_b32 %0, %5, %0\n v_mbcnt_lo_u32_b32 %1, %4, 0\n v_mbcnt_hi_u32_b32 %1, %5, %1\n"
                "v_mbcnt_lo_u32_b32 %2, %4, 0\n v_mbcnt_hi_u32_b32 %2, %5, %2\n v_mbcnt_lo_u32_b32 %3, %4, 0\n v_mbcnt_hi_u32_b32 %3, %5, %3\n"
                "v_mbcnt_lo_u32_b32 %0, %4, 0\n v_mbcnt_hi_u32_b32 %0, %5, %0\n v_mbcnt_lo_u32_b32 %1, %4, 0\n v_mbcnt_hi_u32_b32 %1, %5, %1\n"
                "v_mbcnt_lo_u32_b32 %2, %4, 0\n v_mbcnt_hi_u32_b32 %2, %5, %2\n v_mbcnt_lo_u32_b32 %3, %4, 0\n v_mbcnt_hi_u32_b32 %3, %5, %3\n"
                : "+v"(u0), "+v"(u1), "+v"(s1), "+v"(s2) : "s"((unsigned)m0), "s"((unsigned)(m0 >> 32)));
        } else if (KIND == K_RCP) {
            asm volatile(
                "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == K_SALU_ADD) {
            asm volatile(
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (KIND == K_SALU_AND64) {
            asm volatile(
                "s_and_b64 %0, %0, %1\n s_andn2_b64 %1, %1, %0\n s_or_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0\n"
                "s_and_b64 %0, %0, %1\n s_andn2_b64 %1, %1, %0\n s_or_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0\n"
                "s_and_b64 %0, %0, %1\n s_andn2_b64 %1, %1, %0\n s_or_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0\n"
                "s_and_b64 %0, %0, %1\n s_andn2_b64 %1, %1, %0\n s_or_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0\n"
                : "+s"(m0), "+s"(m1) : : "scc");
        } else if (KIND == K_SALU_BCNT) {
            asm volatile(
                "s_bcnt1_i32_b64 %0, %4\n s_bcnt1_i32_b64 %1, %5\n s_bcnt1_i32_b64 %2, %4\n s_bcnt1_i32_b64 %3, %5\n"
                "s_bcnt1_i32_b64 %0, %4\n s_bcnt1_i32_b64 %1, %5\n s_bcnt1_i32_b64 %2, %4\n s_bcnt1_i32_b64 %3, %5\n"
                "s_bcnt1_i32_b64 %0, %4\n s_bcnt1_i32_b64 %1, %5\n s_bcnt1_i32_b64 %2, %4\n s_bcnt1_i32_b64 %3, %5\n"
                "s_bcnt1_i32_b64 %0, %4\n s_bcnt1_i32_b64 %1, %5\n s_bcnt1_i32_b64 %2, %4\n s_bcnt1_i32_b64 %3, %5\n"
                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(m0), "s"(m1) : "scc");
        } else if (KIND == K_MIX_VS) {
            asm volatile(
                "v_fma_f32 %0, %0, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %10, %11\n s_add_u32 %9, %9, 1\n"
                "v_fma_f32 %2, %2, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %10, %11\n s_add_u32 %9, %9, 1\n"
                "v_fma_f32 %4, %4, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %10, %11\n s_add_u32 %9, %9, 1\n"
                "v_fma_f32 %6, %6, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %7, %7, %10, %11\n s_add_u32 %9, %9, 1\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(b), "v"(c) : "scc");
        } else if (KIND == K_MIX_V2S) {
            asm volatile(
                "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %2, %2, %10, %11\n"
                "v_fma_f32 %3, %3, %10, %11\n s_add_u32 %9, %9, 1\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n"
                "s_add_u32 %8, %8, 1\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n s_add_u32 %9, %9, 1\n"
                "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %2, %2, %10, %11\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(b), "v"(c) : "scc");
        } else if (KIND == K_LDS_RD) {
            asm volatile(
                "ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                "ds_read_b32 %4, %8\n ds_read_b32 %5, %8 offset:256\n ds_read_b32 %6, %8 offset:512\n ds_read_b32 %7, %8 offset:768\n"
                "ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                "ds_read_b32 %4, %8\n ds_read_b32 %5, %8 offset:256\n ds_read_b32 %6, %8 offset:512\n ds_read_b32 %7, %8 offset:768\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(laddr) : "memory");
        } else if (KIND == K_LDS_WR) {
            asm volatile(
                "ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n"
                "ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n"
                "ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n"
                "ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n"
                "s_waitcnt lgkmcnt(0)\n"
                : : "v"(laddr), "v"(a0) : "memory");
        } else if (KIND == K_READLANE) {
            asm volatile(
                "v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n"
                "v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n"
                "v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n"
                "v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n"
                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(u0));
        } else if (KIND == K_WRITELANE) {
            asm volatile(
                "v_writelane_b32 %0, %4, 3\n v_writelane_b32 %1, %5, 5\n v_writelane_b32 %2, %4, 7\n v_writelane_b32 %3, %5, 9\n"
                "v_writelane_b32 %0, %4, 4\n v_writelane_b32 %1, %5, 6\n v_writelane_b32 %2, %4, 8\n v_writelane_b32 %3, %5, 10\n"
                "v_writelane_b32 %0, %4, 3\n v_writelane_b32 %1, %5, 5\n v_writelane_b32 %2, %4, 7\n v_writelane_b32 %3, %5, 9\n"
                "v_writelane_b32 %0, %4, 4\n v_writelane_b32 %1, %5, 6\n v_writelane_b32 %2, %4, 8\n v_writelane_b32 %3, %5, 10\n"
                : "+v"(u0), "+v"(u1), "+v"(s2), "+v"(s3) : "s"(s0), "s"(s1));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 + u1 + s0 + s1 + s2 + s3) +
              (float)(m0 + m1);
    if (r == 123.456f) sink[0] = r;
}

template <int KIND>
static void run_kind(unsigned long long* d_stamps, float* d_sink, int n_simd)
{
    const int ws[] = {1, 2, 4, 6, 8};
    for (int w : ws) {
        const int blocks = n_simd * w;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        k_issue<KIND><<<blocks, 64>>>(d_stamps, d_sink, LOOPS);   // warm
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k_issue<KIND><<<blocks, 64>>>(d_stamps, d_sink, LOOPS);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> raw(2 * (size_t)blocks), st(blocks), rt(blocks);
        CK(hipMemcpy(raw.data(), d_stamps, raw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int i = 0; i < blocks; ++i) { st[i] = raw[2 * i]; rt[i] = raw[2 * i + 1]; }
        std::sort(st.begin(), st.end());
        std::sort(rt.begin(), rt.end());
        const double med = (double)st[blocks / 2];
        const double per = med / (16.0 * LOOPS);
        const double ghz = med / (double)rt[blocks / 2] * 0.1;   // s_memrealtime ticks at 100 MHz
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"cyc_per_inst_per_wave\": %.3f, \"inst_per_cyc_per_simd\": %.4f, "
               "\"wall_us\": %.1f, \"wall_inst_per_ns_chip\": %.2f, \"memtime_ghz\": %.3f}\n",
               kind_name[KIND], w, per, (double)w / per, ms * 1e3, (double)blocks * 16.0 * LOOPS / (ms * 1e6), ghz);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int n_simd = p.multiProcessorCount * 4;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"note\": \"s_memtime ticks are shader cycles per MI355X_MICROARCH.md\"}\n",
           p.name, p.multiProcessorCount, p.clockRate);
    unsigned long long* d_stamps;
    float* d_sink;
    CK(hipMalloc(&d_stamps, (size_t)n_simd * 16 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_sink, 16));
    run_kind<K_FMA>(d_stamps, d_sink, n_simd);
    run_kind<K_FMA_DEP>(d_stamps, d_sink, n_simd);
    run_kind<K_PKFMA>(d_stamps, d_sink, n_simd);
    run_kind<K_CVTUB>(d_stamps, d_sink, n_simd);
    run_kind<K_MAX3>(d_stamps, d_sink, n_simd);
    run_kind<K_CNDMASK>(d_stamps, d_sink, n_simd);
    run_kind<K_CMP>(d_stamps, d_sink, n_simd);
    run_kind<K_MBCNT>(d_stamps, d_sink, n_simd);
    run_kind<K_RCP>(d_stamps, d_sink, n_simd);
    run_kind<K_SALU_ADD>(d_stamps, d_sink, n_simd);
    run_kind<K_SALU_AND64>(d_stamps, d_sink, n_simd);
    run_kind<K_SALU_BCNT>(d_stamps, d_sink, n_simd);
    run_kind<K_MIX_VS>(d_stamps, d_sink, n_simd);
    run_kind<K_MIX_V2S>(d_stamps, d_sink, n_simd);
    run_kind<K_LDS_RD>(d_stamps, d_sink, n_simd);
    run_kind<K_LDS_WR>(d_stamps, d_sink, n_simd);
    run_kind<K_READLANE>(d_stamps, d_sink, n_simd);
    run_kind<K_WRITELANE>(d_stamps, d_sink, n_simd);
    return 0;
}
