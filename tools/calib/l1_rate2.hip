// l1_rate2.hip -- second L1-return calibration on gfx950: what a wave-level load costs the CU's L1 return path (TA/TD) by
// width, by number of active lanes and by address pattern.  Every wave re-reads a 4 KB window (L1 resident), 8 independent
// loads per loop iteration, 8 waves per SIMD.  Patterns: "scatter" = every lane its own 64-byte record (node fetches);
// "quad" = 4 consecutive lanes read the 4 quarters of one record; "linear" = lane i reads bytes [W*i, W*i+W).
// build: hipcc --offload-arch=gfx950 -O2 tools/calib/l1_rate2.hip -o tools/calib/l1_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define L8(OP, T, STRIDE)                                                                                              \
    { T v0, v1, v2, v3, v4, v5, v6, v7;                                                                                \
      asm volatile(OP " %0, %8, off\n" OP " %1, %8, off offset:" #STRIDE "*1\n" OP " %2, %8, off offset:" #STRIDE "*2\n"     \
                   OP " %3, %8, off offset:" #STRIDE "*3\n" OP " %4, %8, off offset:" #STRIDE "*4\n" OP " %5, %8, off offset:" #STRIDE "*5\n" \
                   OP " %6, %8, off offset:" #STRIDE "*6\n" OP " %7, %8, off offset:" #STRIDE "*7\n s_waitcnt vmcnt(0)"      \
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(ptr) : "memory"); \
      acc += first(v0) + first(v1) + first(v2) + first(v3) + first(v4) + first(v5) + first(v6) + first(v7); }
__device__ inline float first(f4 v) { return v.x + v.w; }
__device__ inline float first(f3 v) { return v.x + v.z; }
__device__ inline float first(f2 v) { return v.x + v.y; }
__device__ inline float first(float v) { return v; }

// WIDTH in dwords; PATTERN 0 scatter, 1 quad, 2 linear; ACTIVE = number of active lanes (64, 32, 16: every 1st/2nd/4th lane)
template <int WIDTH, int PATTERN, int ACTIVE>
__global__ __launch_bounds__(64) void k(const char* p, float* out, int iters)
{
    const unsigned lane = threadIdx.x;
    unsigned off;
    if (PATTERN == 0) off = ((lane * 37u) & 63u) * 64u;                       // own record, shuffled
    else if (PATTERN == 1) off = (((lane >> 2) * 37u) & 15u) * 64u + (lane & 3u) * 16u;   // 4 lanes per record
    else off = lane * (WIDTH * 4u);
    const char* ptr = p + (blockIdx.x & 7u) * 8192u + off;
    float acc = 0.f;
    if ((lane & (64 / ACTIVE - 1)) == 0) {
        for (int i = 0; i < iters; ++i) {
            if (PATTERN == 1) {                                               // the 8 loads walk 8 record groups
                L8("global_load_dwordx4", f4, 512)
            } else if (WIDTH == 4) {
                if (PATTERN == 0) L8("global_load_dwordx4", f4, 16) else L8("global_load_dwordx4", f4, 512)
            } else if (WIDTH == 3) {
                L8("global_load_dwordx3", f3, 12)
            } else if (WIDTH == 2) {
                if (PATTERN == 0) L8("global_load_dwordx2", f2, 8) else L8("global_load_dwordx2", f2, 512)
            } else {
                if (PATTERN == 0) L8("global_load_dword", float, 4) else L8("global_load_dword", float, 256)
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int WIDTH, int PATTERN, int ACTIVE>
static void run(const char* name, const char* d, float* o)
{
    const int blocks = 256 * 32, iters = 2048;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        k<WIDTH, PATTERN, ACTIVE><<<blocks, 64>>>(d, o, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double insts = (double)blocks * iters * 8.0;
    printf("{\"load\": \"%s\", \"dwords\": %d, \"active_lanes\": %d, \"ms\": %.3f, \"cycles_per_wave_load_per_cu_at_2p4\": %.2f, \"bytes_per_cycle_per_cu\": %.1f}\n",
           name, WIDTH, ACTIVE, ms, 256.0 * 2.4 / (insts / (ms * 1e6)), (double)ACTIVE * WIDTH * 4.0 / (256.0 * 2.4 / (insts / (ms * 1e6))));
}

int main()
{
    char* d; float* o;
    CK(hipMalloc(&d, 1 << 20)); CK(hipMalloc(&o, 16));
    CK(hipMemset(d, 0x11, 1 << 20));
    run<4, 0, 64>("dwordx4 scatter", d, o);
    run<4, 0, 32>("dwordx4 scatter", d, o);
    run<4, 0, 16>("dwordx4 scatter", d, o);
    run<3, 0, 64>("dwordx3 scatter", d, o);
    run<2, 0, 64>("dwordx2 scatter", d, o);
    run<1, 0, 64>("dword scatter", d, o);
    run<4, 1, 64>("dwordx4 quad (4 lanes per 64-B record)", d, o);
    run<4, 2, 64>("dwordx4 linear", d, o);
    run<2, 2, 64>("dwordx2 linear", d, o);
    run<1, 2, 64>("dword linear", d, o);
    return 0;
}
