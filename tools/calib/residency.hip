// residency.hip -- census of co-resident workgroups per CU on gfx950 for small workgroups.
// Every workgroup stamps s_memrealtime at start and end of a ~100 us spin and records HW_REG_HW_ID / XCC_ID; the host
// counts, per CU, the maximum number of workgroups whose [start, end] intervals overlap.
// build: hipcc --offload-arch=gfx950 -O2 tools/calib/residency.hip -o tools/calib/residency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Rec { unsigned long long t0, t1; unsigned hw_id, xcc_id; };

template <int LDS_BYTES>
__global__ void k_census(Rec* out, unsigned long long spin_ticks)
{
    __shared__ char pad[LDS_BYTES > 0 ? LDS_BYTES : 4];
    if (threadIdx.x == 1000) pad[0] = 1;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t1 = t0;
    while (t1 - t0 < spin_ticks) { __builtin_amdgcn_s_sleep(8); t1 = __builtin_amdgcn_s_memrealtime(); }
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = Rec{t0, t1, hw, xcc};
    }
}

template <int LDS_BYTES>
static void census(int threads, int blocks_per_cu, Rec* d)
{
    const int blocks = 256 * blocks_per_cu;
    k_census<LDS_BYTES><<<blocks, threads>>>(d, 10000);   // 100 us at 100 MHz
    CK(hipDeviceSynchronize());
    std::vector<Rec> r(blocks);
    CK(hipMemcpy(r.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost));
    // CU key: xcc_id (low 4 bits) + HW_ID se_id [15:13], sh_id [12], cu_id [11:8]
    std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
    for (auto& x : r) {
        unsigned key = ((x.xcc_id & 0xf) << 16) | (x.hw_id & 0xff00);
        ev[key].push_back({x.t0, +1});
        ev[key].push_back({x.t1, -1});
    }
    int mx = 0, mn = 1 << 30; double avg = 0;
    for (auto& kv : ev) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end());
        int cur = 0, best = 0;
        for (auto& e : v) { cur += e.second; best = std::max(best, cur); }
        mx = std::max(mx, best); mn = std::min(mn, best); avg += best;
    }
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto& x : r) { tmin = std::min(tmin, x.t0); tmax = std::max(tmax, x.t1); }
    printf("{\"threads_per_wg\": %d, \"lds_bytes\": %d, \"wgs_per_cu_launched\": %d, \"cus_seen\": %zu, \"max_resident_wgs_per_cu\": %d, "
           "\"min\": %d, \"mean\": %.1f, \"span_us\": %.0f}\n", threads, LDS_BYTES, blocks_per_cu, ev.size(), mx, mn, avg / ev.size(),
           (double)(tmax - tmin) / 100.0);
}

int main()
{
    Rec* d;
    CK(hipMalloc(&d, 256 * 64 * sizeof(Rec)));
    census<0>(64, 16, d); census<0>(64, 24, d); census<0>(64, 32, d); census<0>(64, 40, d);
    census<0>(128, 16, d); census<0>(128, 20, d);
    census<0>(256, 8, d); census<0>(256, 10, d);
    census<6400>(64, 24, d); census<6400>(64, 32, d);
    census<6400>(128, 12, d);
    return 0;
}
