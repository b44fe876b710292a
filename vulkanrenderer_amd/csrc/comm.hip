// comm.hip -- the collective of the multi-GPU row-band schedule behind the C boundary (no reference
// counterpart: the reference is single-GPU, Renderer.cpp:359).  A C++ host attaches its communicator once
// (vkr_comm_attach: an RCCL ncclComm_t, or vkr_comm_attach_custom: its own all-gather -- MPI, gloo, a test harness), then
// per frame
//     vkr_rtao_trace_stripes -> vkr_rtao_gather_history -> vkr_rtao_denoise_band -> vkr_gather_bands
// Each gather is ONE fused pack kernel, ONE all-gather of raw bytes on the caller's stream, ONE fused unpack kernel.
// RCCL is resolved with dlopen at attach time, so the library has no link-time dependency on it.
#include "internal.hpp"
#include <dlfcn.h>
#include <mutex>

namespace vkr {

// ---- rows of the interleaved-stripe exchange: rank r owns the 8-row stripes r, r+n, r+2n, ...; its gather slot holds
// k = ceil(ceil(H/8)/n) stripes (ragged ends padded; padded rows travel as copies of the last image row and are dropped)
__host__ __device__ inline uint32_t stripes_per_rank(uint32_t H, uint32_t n) { return (((H + 7u) / 8u) + n - 1u) / n; }

// send[(slot row j) * W + x] = plane[row(rank, j) * W + x]
__global__ __launch_bounds__(256) void k_pack_stripes(const uint16_t* plane, uint16_t* send, uint32_t W, uint32_t H, uint32_t rank,
                                                      uint32_t n, uint32_t slot_rows)
{
    const uint32_t x2 = blockIdx.x * 256u + threadIdx.x, j = blockIdx.y;       // two pixels (4 bytes) per thread
    if (j >= slot_rows || 2u * x2 >= W) return;
    uint32_t row = ((j >> 3) * n + rank) * 8u + (j & 7u);
    if (row > H - 1u) row = H - 1u;
    if (2u * x2 + 1u < W && (W & 1u) == 0u) ((uint32_t*)send)[((size_t)j * W) / 2u + x2] = ((const uint32_t*)plane)[((size_t)row * W) / 2u + x2];
    else { send[(size_t)j * W + 2u * x2] = plane[(size_t)row * W + 2u * x2]; if (2u * x2 + 1u < W) send[(size_t)j * W + 2u * x2 + 1u] = plane[(size_t)row * W + 2u * x2 + 1u]; }
}
// plane[row * W + x] = recv[(r * slot_rows + j) * W + x] for every image row exactly once
__global__ __launch_bounds__(256) void k_unpack_stripes(const uint16_t* recv, uint16_t* plane, uint32_t W, uint32_t H, uint32_t n,
                                                        uint32_t slot_rows)
{
    const uint32_t x2 = blockIdx.x * 256u + threadIdx.x, row = blockIdx.y;
    if (row >= H || 2u * x2 >= W) return;
    const uint32_t s = row >> 3, r = s % n, j = (s / n) * 8u + (row & 7u);
    const size_t src = ((size_t)r * slot_rows + j) * W, dst = (size_t)row * W;
    if (2u * x2 + 1u < W && (W & 1u) == 0u) ((uint32_t*)plane)[dst / 2u + x2] = ((const uint32_t*)recv)[src / 2u + x2];
    else { plane[dst + 2u * x2] = recv[src + 2u * x2]; if (2u * x2 + 1u < W) plane[dst + 2u * x2 + 1u] = recv[src + 2u * x2 + 1u]; }
}

// ---- contiguous row bands: rank r owns rows [r*Hb, min(H, (r+1)*Hb)), Hb = ceil(H/n); slot = [plane][Hb rows][W]
struct PlanePtrs { uint16_t* p[VKR_GATHER_MAX_PLANES]; };
__global__ __launch_bounds__(256) void k_pack_bands(PlanePtrs planes, uint16_t* send, uint32_t W, uint32_t H, uint32_t rank, uint32_t hb)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    const uint32_t row = rank * hb + j;
    if (x >= W || j >= hb || row >= H) return;                 // pad rows of the last band stay as they are
    send[((size_t)k * hb + j) * W + x] = planes.p[k][(size_t)row * W + x];
}
__global__ __launch_bounds__(256) void k_unpack_bands(const uint16_t* recv, PlanePtrs planes, uint32_t W, uint32_t H, uint32_t hb,
                                                      uint32_t n_planes)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, row = blockIdx.y, k = blockIdx.z;
    if (x >= W || row >= H) return;
    const uint32_t r = row / hb, j = row - r * hb;
    planes.p[k][(size_t)row * W + x] = recv[(((size_t)r * n_planes + k) * hb + j) * W + x];
}

// ---- RCCL through dlopen
typedef int (*fn_allgather_t)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);
typedef const char* (*fn_errstr_t)(int);
typedef int (*fn_commint_t)(void*, int*);
struct RcclApi { void* handle = nullptr; fn_allgather_t all_gather = nullptr; fn_errstr_t err = nullptr; fn_commint_t count = nullptr, user_rank = nullptr; };
static RcclApi g_rccl;
static std::mutex g_rccl_mutex;       // contexts on different host threads may attach at the same time

static const char* rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return nullptr;
    // the copy of RCCL the host process already holds made the communicator: take that one (matched by soname) before
    // loading another; VKR_RCCL_LIB names a path when neither is right
    void* h = nullptr;
    if (const char* path = getenv("VKR_RCCL_LIB")) h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return "librccl.so not found (dlopen)";
    RcclApi a;
    a.handle = h;
    a.all_gather = (fn_allgather_t)dlsym(h, "ncclAllGather");
    a.err = (fn_errstr_t)dlsym(h, "ncclGetErrorString");
    a.count = (fn_commint_t)dlsym(h, "ncclCommCount");
    a.user_rank = (fn_commint_t)dlsym(h, "ncclCommUserRank");
    if (!a.all_gather || !a.err || !a.count || !a.user_rank) return "librccl.so lacks ncclAllGather / ncclCommCount / ncclCommUserRank";
    g_rccl = a;
    return nullptr;
}

// all-gather of raw bytes on `st`: RCCL, the host's own transport, or (one rank, nothing attached) a device copy
static int gather_bytes(vkr_ctx* c, const void* send, void* recv, size_t bytes, hipStream_t st, std::string& why)
{
    if (c->comm_fn) {
        int rc = c->comm_fn(c->comm_user, send, recv, bytes, (void*)st);
        if (rc != 0) { why = "the attached all-gather callback returned " + std::to_string(rc); return VKR_ERR_RCCL; }
        return VKR_OK;
    }
    if (c->comm_nccl) {
        int rc = g_rccl.all_gather(send, recv, bytes, 0 /* ncclChar */, c->comm_nccl, st);
        if (rc != 0) { why = std::string("ncclAllGather: ") + g_rccl.err(rc); return VKR_ERR_RCCL; }
        return VKR_OK;
    }
    why = "no communicator attached (vkr_comm_attach)";
    return VKR_ERR_STATE;
}

}  // namespace vkr

using namespace vkr;

static int comm_fail(vkr_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}

static int comm_alloc(vkr_ctx* c, uint32_t rank, uint32_t n)
{
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail(c, VKR_ERR_HIP, "hipSetDevice failed");
    const size_t slot_rows = (size_t)stripes_per_rank(c->H, n) * 8u;
    const size_t hb = ((size_t)c->H + n - 1u) / n;
    size_t send = slot_rows * c->W;
    if (send < (size_t)VKR_GATHER_MAX_PLANES * hb * c->W) send = (size_t)VKR_GATHER_MAX_PLANES * hb * c->W;
    if (c->d_comm_send.alloc(send) != hipSuccess || c->d_comm_recv.alloc(send * n) != hipSuccess)
        return comm_fail(c, VKR_ERR_OOM, "vkr_comm_attach: gather buffers");
    c->comm_rank = rank;
    c->comm_n = n;
    return VKR_OK;
}

extern "C" {

int vkr_comm_attach(vkr_ctx* c, void* nccl_comm, uint32_t rank, uint32_t n_ranks)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    c->comm_nccl = nullptr; c->comm_fn = nullptr; c->comm_user = nullptr; c->comm_n = 0; c->comm_rank = 0;
    if (!nccl_comm) { c->d_comm_send.release(); c->d_comm_recv.release(); return VKR_OK; }      // detach
    if (n_ranks == 0 || rank >= n_ranks || n_ranks > 4096u) return comm_fail(c, VKR_ERR_INVALID_ARG, "vkr_comm_attach: bad rank / rank count");
    if (const char* e = rccl_load()) return comm_fail(c, VKR_ERR_RCCL, std::string("vkr_comm_attach: ") + e);
    int cnt = -1, ur = -1, rc;
    if ((rc = g_rccl.count(nccl_comm, &cnt)) != 0 || (rc = g_rccl.user_rank(nccl_comm, &ur)) != 0)
        return comm_fail(c, VKR_ERR_RCCL, std::string("vkr_comm_attach: ") + g_rccl.err(rc));
    if (cnt != (int)n_ranks || ur != (int)rank)
        return comm_fail(c, VKR_ERR_INVALID_ARG, "vkr_comm_attach: the communicator is rank " + std::to_string(ur) + " of " + std::to_string(cnt) +
                                                 ", not rank " + std::to_string(rank) + " of " + std::to_string(n_ranks));
    if (int r = comm_alloc(c, rank, n_ranks)) return r;
    c->comm_nccl = nccl_comm;
    return VKR_OK;
}

int vkr_comm_attach_custom(vkr_ctx* c, vkr_allgather_fn fn, void* user, uint32_t rank, uint32_t n_ranks)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    c->comm_nccl = nullptr; c->comm_fn = nullptr; c->comm_user = nullptr; c->comm_n = 0; c->comm_rank = 0;
    if (!fn) { c->d_comm_send.release(); c->d_comm_recv.release(); return VKR_OK; }
    if (n_ranks == 0 || rank >= n_ranks || n_ranks > 4096u) return comm_fail(c, VKR_ERR_INVALID_ARG, "vkr_comm_attach_custom: bad rank / rank count");
    if (int r = comm_alloc(c, rank, n_ranks)) return r;
    c->comm_fn = fn;
    c->comm_user = user;
    return VKR_OK;
}

int vkr_rtao_gather_history(vkr_ctx* c, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (c->comm_n == 0) return comm_fail(c, VKR_ERR_STATE, "vkr_rtao_gather_history: no communicator attached (vkr_comm_attach)");
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail(c, VKR_ERR_HIP, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    uint16_t* hist = c->d_hist_img[c->frame_parity].p;              // history[curr] of the frame in flight (vkr_rtao_history_curr)
    const uint32_t W = c->W, H = c->H, n = c->comm_n, slot_rows = stripes_per_rank(H, n) * 8u;
    dim3 gp(((W + 1u) / 2u + 255u) / 256u, slot_rows), gu(((W + 1u) / 2u + 255u) / 256u, H);
    k_pack_stripes<<<gp, 256, 0, st>>>(hist, c->d_comm_send.p, W, H, c->comm_rank, n, slot_rows);
    std::string why;
    if (int rc = gather_bytes(c, c->d_comm_send.p, c->d_comm_recv.p, (size_t)slot_rows * W * 2u, st, why)) return comm_fail(c, rc, "vkr_rtao_gather_history: " + why);
    k_unpack_stripes<<<gu, 256, 0, st>>>(c->d_comm_recv.p, hist, W, H, n, slot_rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return comm_fail(c, VKR_ERR_HIP, std::string("vkr_rtao_gather_history: ") + hipGetErrorString(e));
    return VKR_OK;
}

int vkr_gather_bands(vkr_ctx* c, uint16_t* const* planes_r16f, uint32_t n_planes, void* stream)
{
    if (!c) return VKR_ERR_INVALID_ARG;
    if (!planes_r16f || n_planes == 0 || n_planes > VKR_GATHER_MAX_PLANES) return comm_fail(c, VKR_ERR_INVALID_ARG, "vkr_gather_bands: 1..4 planes");
    for (uint32_t k = 0; k < n_planes; ++k) if (!planes_r16f[k]) return comm_fail(c, VKR_ERR_INVALID_ARG, "vkr_gather_bands: NULL plane");
    if (c->comm_n == 0) return comm_fail(c, VKR_ERR_STATE, "vkr_gather_bands: no communicator attached (vkr_comm_attach)");
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail(c, VKR_ERR_HIP, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t W = c->W, H = c->H, n = c->comm_n, hb = (H + n - 1u) / n;
    PlanePtrs pp;
    for (uint32_t k = 0; k < VKR_GATHER_MAX_PLANES; ++k) pp.p[k] = k < n_planes ? planes_r16f[k] : nullptr;
    dim3 gp((W + 255u) / 256u, hb, n_planes), gu((W + 255u) / 256u, H, n_planes);
    k_pack_bands<<<gp, 256, 0, st>>>(pp, c->d_comm_send.p, W, H, c->comm_rank, hb);
    std::string why;
    if (int rc = gather_bytes(c, c->d_comm_send.p, c->d_comm_recv.p, (size_t)n_planes * hb * W * 2u, st, why)) return comm_fail(c, rc, "vkr_gather_bands: " + why);
    k_unpack_bands<<<gu, 256, 0, st>>>(c->d_comm_recv.p, pp, W, H, hb, n_planes);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return comm_fail(c, VKR_ERR_HIP, std::string("vkr_gather_bands: ") + hipGetErrorString(e));
    return VKR_OK;
}

}  // extern "C"
