// typed_load.hip -- does the texture-address path of gfx950 convert packed bytes to floats for free?
// buffer_load_format_xyzw through a V# with DATA_FORMAT 8_8_8_8 / NUM_FORMAT USCALED returns the four bytes of a dword as
// four floats (0..255).  Checks the values and times it against global_load_dword + 4 x v_cvt_f32_ubyteN.
// build: hipcc --offload-arch=gfx950 -O2 tools/calib/typed_load.hip -o tools/calib/typed_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i4 make_rsrc(const void* p, unsigned bytes, unsigned word3)
{
    unsigned long long a = (unsigned long long)p;
    i4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));   // stride 0
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = __builtin_amdgcn_readfirstlane((int)word3);
    return r;
}
constexpr unsigned W3_U8X4_USCALED = 4u | (5u << 3) | (6u << 6) | (7u << 9) | (2u << 12) | (10u << 15);

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* p, float* out, unsigned n, int iters)
{
    const i4 rs = make_rsrc(p, n * 4u, W3_U8X4_USCALED);
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 977u % n;
    f4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        f4 v;
        if (MODE == 0) {
            unsigned off = idx * 4u;
            asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(off), "s"(rs) : "memory");
        } else {
            unsigned w = p[idx];
            v = f4{(float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24)};
        }
        acc += v;
        idx = (idx + 7919u) % n;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc.x + 2.0f * acc.y + 4.0f * acc.z + 8.0f * acc.w;
}

int main()
{
    const unsigned n = 1u << 20;
    std::vector<unsigned> h(n);
    for (unsigned i = 0; i < n; ++i) h[i] = i * 2654435761u;
    unsigned* d; float *o0, *o1;
    const int blocks = 2048, iters = 256;
    CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&o0, blocks * 256 * 4)); CK(hipMalloc(&o1, blocks * 256 * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    float ms[2];
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) k<0><<<blocks, 256>>>(d, o0, n, iters); else k<1><<<blocks, 256>>>(d, o1, n, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[mode], e0, e1));
        }
    }
    std::vector<float> a(blocks * 256), b(blocks * 256);
    CK(hipMemcpy(a.data(), o0, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o1, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += (a[i] != b[i]);
    printf("{\"typed_load_mismatches\": %zu, \"of\": %zu, \"ms_typed\": %.3f, \"ms_dword_cvt\": %.3f, \"sample\": [%.1f, %.1f]}\n", bad, a.size(), ms[0], ms[1], a[5], b[5]);
    return 0;
}
