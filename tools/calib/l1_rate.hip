// l1_rate.hip -- L1-hit return rate per CU on gfx950 for (a) global_load_dwordx4, (b) global_load_dword,
// (c) buffer_load_format_xyzw from packed bytes (4 B fetched, 16 B returned per lane).  Each wave re-reads a 4 KB window
// (L1 resident), 8 independent loads per loop iteration, 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O2 tools/calib/l1_rate.hip -o tools/calib/l1_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
constexpr unsigned W3_U8X4_USCALED = 4u | (5u << 3) | (6u << 6) | (7u << 9) | (2u << 12) | (10u << 15);
constexpr unsigned W3_RAW = 4u | (5u << 3) | (6u << 6) | (7u << 9) | (4u << 15);

template <int MODE>
__global__ __launch_bounds__(64) void k(const unsigned* p, float* out, int iters)
{
    unsigned long long a = (unsigned long long)p;
    i4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    rs.z = 1 << 24;
    rs.w = (int)(MODE == 2 ? W3_U8X4_USCALED : W3_RAW);
    // scattered within a 4 KB window per wave (like node fetches: each lane its own 64-byte record)
    unsigned off = ((threadIdx.x * 37u) & 63u) * 64u + (blockIdx.x & 7u) * 4096u;
    f4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        f4 v0, v1, v2, v3, v4, v5, v6, v7;
        if (MODE == 0) {
            asm volatile(
                "buffer_load_dwordx4 %0, %8, %9, 0 offen\n buffer_load_dwordx4 %1, %8, %9, 0 offen offset:16\n"
                "buffer_load_dwordx4 %2, %8, %9, 0 offen offset:32\n buffer_load_dwordx4 %3, %8, %9, 0 offen offset:48\n"
                "buffer_load_dwordx4 %4, %8, %9, 0 offen offset:64\n buffer_load_dwordx4 %5, %8, %9, 0 offen offset:80\n"
                "buffer_load_dwordx4 %6, %8, %9, 0 offen offset:96\n buffer_load_dwordx4 %7, %8, %9, 0 offen offset:112\n"
                "s_waitcnt vmcnt(0)"
                : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(off), "s"(rs) : "memory");
        } else if (MODE == 1) {
            float s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile(
                "buffer_load_dword %0, %8, %9, 0 offen\n buffer_load_dword %1, %8, %9, 0 offen offset:4\n"
                "buffer_load_dword %2, %8, %9, 0 offen offset:8\n buffer_load_dword %3, %8, %9, 0 offen offset:12\n"
                "buffer_load_dword %4, %8, %9, 0 offen offset:16\n buffer_load_dword %5, %8, %9, 0 offen offset:20\n"
                "buffer_load_dword %6, %8, %9, 0 offen offset:24\n buffer_load_dword %7, %8, %9, 0 offen offset:28\n"
                "s_waitcnt vmcnt(0)"
                : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(s4), "=&v"(s5), "=&v"(s6), "=&v"(s7) : "v"(off), "s"(rs) : "memory");
            v0 = f4{s0, s1, s2, s3}; v1 = f4{s4, s5, s6, s7}; v2 = v3 = v4 = v5 = v6 = v7 = f4{0, 0, 0, 0};
        } else {
            asm volatile(
                "buffer_load_format_xyzw %0, %8, %9, 0 offen\n buffer_load_format_xyzw %1, %8, %9, 0 offen offset:4\n"
                "buffer_load_format_xyzw %2, %8, %9, 0 offen offset:8\n buffer_load_format_xyzw %3, %8, %9, 0 offen offset:12\n"
                "buffer_load_format_xyzw %4, %8, %9, 0 offen offset:16\n buffer_load_format_xyzw %5, %8, %9, 0 offen offset:20\n"
                "buffer_load_format_xyzw %6, %8, %9, 0 offen offset:24\n buffer_load_format_xyzw %7, %8, %9, 0 offen offset:28\n"
                "s_waitcnt vmcnt(0)"
                : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(off), "s"(rs) : "memory");
        }
        acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
        off ^= 64u;
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

int main()
{
    unsigned* d; float* o;
    CK(hipMalloc(&d, 1 << 20)); CK(hipMalloc(&o, 16));
    CK(hipMemset(d, 0x11, 1 << 20));
    const int blocks = 256 * 32, iters = 4096;
    const char* names[3] = {"buffer_load_dwordx4 (16 B/lane)", "buffer_load_dword (4 B/lane)", "buffer_load_format_xyzw u8x4 (4 B fetched, 16 B returned)"};
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) k<0><<<blocks, 64>>>(d, o, iters); else if (mode == 1) k<1><<<blocks, 64>>>(d, o, iters); else k<2><<<blocks, 64>>>(d, o, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double insts = (double)blocks * iters * 8.0;
        printf("{\"load\": \"%s\", \"ms\": %.3f, \"wave_loads_per_ns_chip\": %.2f, \"cycles_per_wave_load_per_cu_at_2p4\": %.2f}\n", names[mode], ms,
               insts / (ms * 1e6), 256.0 * 2.4 / (insts / (ms * 1e6)));
    }
    return 0;
}
