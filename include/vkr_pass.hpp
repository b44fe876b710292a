// vkr_pass.hpp -- C++ adapter from the reference's pass interface to the C ABI of vkr_rt.h (header only).
//
// The reference registers passes as `class X : public RenderPass` with three protected overrides
// (app/include/Rendering/core/RenderPass.h:46-67) and runs them from Rendergraph::Execute
// (app/src/Rendering/core/Rendergraph.cpp:295-322).  HipRtaoPass keeps exactly that shape, so swapping it for
// RtaoComputePass (app/include/Rendering/pass/RtaoComputePass.h:10-41) is one line in Renderer::init.
// The base class and its execute-context type are template parameters: inside the reference they are
// `RenderPass` / `PassExecuteContext`; this repository's demo and tests instantiate it over a minimal base of
// the same shape (vulkanrenderer_amd/csrc/pass_adapter_demo.cpp), so nothing here depends on Vulkan headers.
//
// Error convention: the C ABI returns a status and never throws; the adapter rethrows as std::runtime_error, which
// is what the reference's callers already handle (RayTracingContext.cpp:49-53, Rendergraph.cpp:95,120).
#pragma once

#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vkr_rt.h"

namespace vkr {

inline void check(vkr_ctx* ctx, int status, const char* what)
{
    if (status == VKR_OK) return;
    const char* msg = vkr_last_error(ctx);
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string(status) + "): " + (msg ? msg : ""));
}

// Owns one vkr_ctx: the RTAO slice of FrameManager (history x2, ping-pong x2), RtaoComputePipeline and
// RayTracingContext on one GPU.  Release order mirrors Renderer::cleanup (Renderer.cpp:224-253): one destroy.
class HipContext {
public:
    HipContext(int device, uint32_t width, uint32_t height)
    {
        vkr_create_desc d{device, width, height};
        check(nullptr, vkr_create(&d, &ctx_), "vkr_create");
    }
    ~HipContext() { vkr_destroy(ctx_); }
    HipContext(const HipContext&) = delete;
    HipContext& operator=(const HipContext&) = delete;
    vkr_ctx* get() const { return ctx_; }

private:
    vkr_ctx* ctx_ = nullptr;
};

// RayTracingContext (app/include/Rendering/RHI/Vulkan/RayTracingContext.h:15-66): init() builds, updateTopLevelAS()
// refits when transforms changed (gated by the caller like Renderer.cpp:470-479).
class HipRayTracingContext {
public:
    explicit HipRayTracingContext(HipContext& c) : ctx_(c.get()) {}
    void init(const vkr_scene_desc& scene, void* stream = nullptr)
    {
        check(ctx_, vkr_scene_upload(ctx_, &scene), "vkr_scene_upload");
        check(ctx_, vkr_bvh_build(ctx_, stream), "vkr_bvh_build");
    }
    void updateTopLevelAS(const std::vector<vkr_instance>& instances, void* stream = nullptr)
    {
        check(ctx_, vkr_bvh_refit(ctx_, instances.data(), (uint32_t)instances.size(), stream), "vkr_bvh_refit");
    }
    vkr_bvh_info info() const
    {
        vkr_bvh_info i{};
        check(ctx_, vkr_bvh_info_get(ctx_, &i), "vkr_bvh_info_get");
        return i;
    }

private:
    vkr_ctx* ctx_;
};

// FrameManager's shared-opaque indirect path (FrameManager.cpp:81-175): buildSharedDrawSlots once per model,
// prepareSharedOpaqueIndirect per frame; drawData / indirect commands are DEVICE buffers of maxDraws entries.
class HipSharedOpaqueIndirect {
public:
    explicit HipSharedOpaqueIndirect(HipContext& c) : ctx_(c.get()) {}
    void buildSharedDrawSlots(const vkr_indirect_desc& desc)
    {
        check(ctx_, vkr_indirect_upload(ctx_, &desc), "vkr_indirect_upload");
        maxDraws_ = desc.max_draws < 1u ? 1u : desc.max_draws;
        spans_.assign(desc.slot_count ? desc.slot_count : 1u, vkr_bucket_span{});
    }
    // viewProj == nullptr: the reference as it is (no culling)
    void prepareSharedOpaqueIndirect(const float modelMatrix[16], const float* viewProj, float* drawData,
                                     vkr_draw_indexed_indirect_command* commands, void* stream = nullptr)
    {
        check(ctx_, vkr_indirect_prepare(ctx_, modelMatrix, viewProj, drawData, commands, maxDraws_, stream), "vkr_indirect_prepare");
        uint32_t n = 0;
        check(ctx_, vkr_indirect_result(ctx_, &drawCount_, spans_.data(), (uint32_t)spans_.size(), &n, stream), "vkr_indirect_result");
        spanCount_ = n < spans_.size() ? n : (uint32_t)spans_.size();
    }
    uint32_t getSharedOpaqueDrawCount() const { return drawCount_; }
    const vkr_bucket_span* getSharedOpaqueBucketSpans() const { return spans_.data(); }
    uint32_t getSharedOpaqueBucketSpanCount() const { return spanCount_; }

private:
    vkr_ctx* ctx_;
    uint32_t maxDraws_ = 1, drawCount_ = 0, spanCount_ = 0;
    std::vector<vkr_bucket_span> spans_;
};

// EquirectToCubemap::convert + IblPrecompute::compute (EquirectToCubemap.h, IblPrecompute.h) over caller-owned device arrays.
struct HipIblPrecompute {
    static void convert(HipContext& c, const float* equirect, uint32_t width, uint32_t height, float* envCube,
                        uint32_t cubeSize = 512, void* stream = nullptr)
    {
        check(c.get(), vkr_ibl_equirect_to_cube(c.get(), equirect, width, height, envCube, cubeSize, stream), "vkr_ibl_equirect_to_cube");
    }
    static void compute(HipContext& c, const float* envCube, uint32_t envSize, uint16_t* irradiance, uint16_t* prefilter,
                        uint16_t* brdfLut, uint32_t irradianceSize = 32, uint32_t prefilterSize = 128,
                        uint32_t brdfLutSize = 512, void* stream = nullptr)
    {
        check(c.get(), vkr_ibl_compute(c.get(), envCube, envSize, irradiance, irradianceSize, prefilter, prefilterSize,
                                       brdfLut, brdfLutSize, stream), "vkr_ibl_compute");
    }
};

// What RtaoComputePass pulls from FrameManager getters each frame (RtaoComputePass.cpp:289-368), as one struct the
// host fills: the UBO it already wrote (FrameManager.cpp:448-511) and the device pointers of the interop images.
struct RtaoFrameInputs {
    const float* ubo = nullptr;              // PBRUniformBufferObject, 120 floats (host)
    const uint16_t* linearDepth = nullptr;   // linearDepthResolve R16F (device)
    const uint16_t* normal = nullptr;        // normalResolve RGBA16F (device)
    uint16_t* rtaoFull = nullptr;            // rtaoFull R16F (device)
    void* stream = nullptr;                  // hipStream_t standing in for ctx.commandBuffer
    uint32_t atrousIterations = 5;           // AppConfig::RTAO_ATROUS_ITERATIONS
    bool enableRtao = true;                  // AppConfig::ENABLE_RTAO (RtaoComputePass.cpp:104-106)
    bool spatialDenoise = true;              // AppConfig::ENABLE_RTAO_SPATIAL_DENOISE (RtaoComputePass.cpp:529)
};

// The synchronisation half of the pass.  In the reference beginPass records three image barriers that make the prepass
// attachments readable by the compute stage -- depth (late fragment tests -> compute), normal and linearDepth (colour
// attachment output -> compute), RtaoComputePass.cpp:27-100 -- and endPass hands rtao_full to the fragment stage and returns
// the three attachments to their attachment layouts (:158-255).  On the HIP side there are no layouts; what is left of those
// barriers is ordering between the host's Vulkan queue and the HIP stream, which only the host can express (it owns the
// exported timeline semaphore / the events).  The adapter therefore takes two callables and runs them exactly where the
// reference records its barriers:
//   beforePass(ectx, stream): make `stream` wait until the prepass has written linearDepth / normal
//                             (hipWaitExternalSemaphoresAsync on the semaphore the prepass submit signals, or hipStreamWaitEvent)
//   afterPass(ectx, stream):  publish rtao_full to the consumer (hipSignalExternalSemaphoresAsync, which the ScenePass
//                             submit waits on at the fragment stage, or hipEventRecord) -- and, by the same signal, tell
//                             the host that linearDepth / normal may be attachments again
// Both are optional; an empty hook is the single-stream case, where stream order already is the barrier.
template <class ExecuteContext>
struct RtaoPassSync {
    std::function<void(const ExecuteContext&, void* stream)> beforePass;   // RtaoComputePass::beginPass, RtaoComputePass.cpp:27-100
    std::function<void(const ExecuteContext&, void* stream)> afterPass;    // RtaoComputePass::endPass,   RtaoComputePass.cpp:158-255
};

template <class RenderPassBase, class ExecuteContext>
class HipRtaoPass : public RenderPassBase {
public:
    using InputProvider = std::function<RtaoFrameInputs(const ExecuteContext&)>;
    using Sync = RtaoPassSync<ExecuteContext>;

    HipRtaoPass(HipContext& context, InputProvider provider, Sync sync = Sync{})
        : RenderPassBase("RtaoComputePass", {"depth"}, {"rtao_full"}),   // RtaoComputePass.cpp:16-25
          ctx_(context.get()), provider_(std::move(provider)), sync_(std::move(sync))
    {
    }

protected:
    void beginPass(const ExecuteContext& ectx) override                   // RtaoComputePass.cpp:27-100: the three acquire barriers
    {
        in_ = provider_(ectx);                                            // the frame's inputs, once per execute()
        if (sync_.beforePass) sync_.beforePass(ectx, in_.stream);
    }
    void render(const ExecuteContext&) override                           // RtaoComputePass.cpp:102-156
    {
        const RtaoFrameInputs& in = in_;
        if (!in.enableRtao) return;                                       // :104-106
        if (!in.ubo) throw std::runtime_error("HipRtaoPass: ubo is null");
        vkr_rtao_frame_io io{};
        std::memcpy(io.ubo, in.ubo, sizeof(io.ubo));
        io.atrous_iterations = in.spatialDenoise ? (in.atrousIterations < 1u ? 1u : in.atrousIterations) : 1u;  // :529
        io.linear_depth_r16f = in.linearDepth;
        io.normal_rgba16f = in.normal;
        io.rtao_full_r16f = in.rtaoFull;
        io.ray_hit_mask = nullptr;
        check(ctx_, vkr_rtao_execute(ctx_, &io, in.stream), "vkr_rtao_execute");
    }
    void endPass(const ExecuteContext& ectx) override                     // RtaoComputePass.cpp:158-255: release rtao_full + the attachments
    {
        if (sync_.afterPass) sync_.afterPass(ectx, in_.stream);
    }

private:
    vkr_ctx* ctx_;
    InputProvider provider_;
    Sync sync_;
    RtaoFrameInputs in_{};
};

}  // namespace vkr
