// pass_adapter_demo.cpp -- drives include/vkr_pass.hpp the way Renderer/Rendergraph drive RtaoComputePass, using only the
// C ABI (links libvkr_rt.so + the HIP runtime for device memory).  Used by tests/test_gpu_adapter.py:
//   pass_adapter_demo <scene.bin> <out.bin>
// scene.bin (little endian): u32 W,H,iters,frames,nverts,nidx,ninst | f32 verts[nverts*3] | u32 idx[nidx] |
//   f32 xf[ninst*12] | f32 ubo[frames*120].   out.bin: u16 rtao_full[W*H] of the last frame.
// The base class below has the shape of the reference's RenderPass (RenderPass.h:46-67, RenderPass.cpp:11-16) and the
// mini graph the order rule of Rendergraph::Compile/Execute (Rendergraph.cpp:150-204,295-322).
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "../../include/vkr_pass.hpp"

namespace demo {

struct PassExecuteContext {   // RenderPass.h:38-44 with the command buffer replaced by a stream
    void* stream = nullptr;
    uint32_t imageIndex = 0;
    const float* ubo = nullptr;
};

class RenderPass {
public:
    RenderPass(std::string passName, std::vector<std::string> in, std::vector<std::string> out)
        : name(std::move(passName)), inputs(std::move(in)), outputs(std::move(out)) {}
    virtual ~RenderPass() = default;
    const std::string& getName() const { return name; }
    const std::vector<std::string>& getInputs() const { return inputs; }
    const std::vector<std::string>& getOutputs() const { return outputs; }
    void execute(const PassExecuteContext& ctx) { beginPass(ctx); render(ctx); endPass(ctx); }

protected:
    virtual void beginPass(const PassExecuteContext&) = 0;
    virtual void render(const PassExecuteContext&) = 0;
    virtual void endPass(const PassExecuteContext&) = 0;
    std::string name;
    std::vector<std::string> inputs, outputs;
};

// stands in for DepthPrepass: outputs {"depth","normal"}.  It runs on its own stream (the host's graphics queue); the two
// events are the demo's stand-in for the exported timeline semaphore of INTEGRATION.md: `written` is what the RTAO pass waits
// on before it reads the attachments, `released` is what it signals once it no longer needs them.
class GBufferPass : public RenderPass {
public:
    GBufferPass(vkr_ctx* c, uint16_t* depth, uint16_t* normal, hipStream_t gfx, hipEvent_t written, hipEvent_t released)
        : RenderPass("DepthPrepass", {}, {"depth", "normal"}), ctx(c), d(depth), n(normal), st(gfx), evWritten(written), evReleased(released) {}

protected:
    void beginPass(const PassExecuteContext&) override
    {
        if (hipStreamWaitEvent(st, evReleased, 0) != hipSuccess) throw std::runtime_error("hipStreamWaitEvent");   // last frame's reader is done
    }
    void render(const PassExecuteContext& e) override
    {
        vkr::check(ctx, vkr_gbuffer_primary(ctx, e.ubo, d, n, nullptr, nullptr, nullptr, st), "vkr_gbuffer_primary");
    }
    void endPass(const PassExecuteContext&) override
    {
        if (hipEventRecord(evWritten, st) != hipSuccess) throw std::runtime_error("hipEventRecord");
    }
    vkr_ctx* ctx;
    uint16_t *d, *n;
    hipStream_t st;
    hipEvent_t evWritten, evReleased;
};

class Rendergraph {
public:
    void AddPass(std::unique_ptr<RenderPass> p) { passes.push_back(std::move(p)); }
    void Compile()
    {
        order.clear();
        std::vector<int> state(passes.size(), 0);
        for (size_t i = 0; i < passes.size(); ++i) visit(i, state);
    }
    void Execute(const PassExecuteContext& ctx) { for (size_t i : order) passes[i]->execute(ctx); }
    std::vector<std::string> orderNames() const
    {
        std::vector<std::string> r;
        for (size_t i : order) r.push_back(passes[i]->getName());
        return r;
    }

private:
    void visit(size_t i, std::vector<int>& state)
    {
        if (state[i] == 2) return;
        if (state[i] == 1) throw std::runtime_error("Rendergraph: cycle detected");
        state[i] = 1;
        for (const auto& in : passes[i]->getInputs())
            for (size_t j = 0; j < passes.size(); ++j)
                if (j != i)
                    for (const auto& out : passes[j]->getOutputs())
                        if (out == in) visit(j, state);
        state[i] = 2;
        order.push_back(i);
    }
    std::vector<std::unique_ptr<RenderPass>> passes;
    std::vector<size_t> order;
};

}  // namespace demo

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

// `indirect in.bin out.bin`: FrameManager::buildSharedDrawSlots + prepareSharedOpaqueIndirect through the C++ wrapper.
// in:  u32 {nodes, slots, meshes, maxDraws, cull}, i32 parent[nodes], f32 local[nodes*16], slots, infos, f32 bounds[meshes*6],
//      f32 model[16], f32 viewProj[16]      out: u32 drawCount, u32 spanCount, spans, commands[drawCount], drawData[drawCount*16]
static int run_indirect(const char* in_path, const char* out_path)
{
    FILE* f = fopen(in_path, "rb");
    if (!f) { perror(in_path); return 66; }
    uint32_t h[5];
    if (fread(h, 4, 5, f) != 5) return 65;
    const uint32_t nn = h[0], ns = h[1], nm = h[2], maxDraws = h[3], cull = h[4];
    std::vector<int32_t> parent(nn);
    std::vector<float> local((size_t)nn * 16), bounds((size_t)nm * 6);
    std::vector<vkr_draw_slot> slots(ns);
    std::vector<vkr_mesh_draw_info> infos(nm);
    float model[16], vp[16];
    if (fread(parent.data(), 4, nn, f) != nn || fread(local.data(), 4, local.size(), f) != local.size() ||
        fread(slots.data(), sizeof(vkr_draw_slot), ns, f) != ns || fread(infos.data(), sizeof(vkr_mesh_draw_info), nm, f) != nm ||
        fread(bounds.data(), 4, bounds.size(), f) != bounds.size() || fread(model, 4, 16, f) != 16 || fread(vp, 4, 16, f) != 16) return 65;
    fclose(f);
    try {
        vkr::HipContext context(0, 16, 16);
        vkr::HipSharedOpaqueIndirect ind(context);
        vkr_indirect_desc d{parent.data(), local.data(), nn, slots.data(), ns, infos.data(), nm, bounds.data(), maxDraws};
        ind.buildSharedDrawSlots(d);
        float* drawData = nullptr;
        vkr_draw_indexed_indirect_command* cmds = nullptr;
        const uint32_t cap = maxDraws < 1u ? 1u : maxDraws;
        HIP_CHECK(hipMalloc((void**)&drawData, (size_t)cap * 64));
        HIP_CHECK(hipMalloc((void**)&cmds, (size_t)cap * sizeof(vkr_draw_indexed_indirect_command)));
        ind.prepareSharedOpaqueIndirect(model, cull ? vp : nullptr, drawData, cmds);
        const uint32_t n = ind.getSharedOpaqueDrawCount(), sc = ind.getSharedOpaqueBucketSpanCount();
        std::vector<vkr_draw_indexed_indirect_command> hc(n);
        std::vector<float> hd((size_t)n * 16);
        if (n) {
            HIP_CHECK(hipMemcpy(hc.data(), cmds, (size_t)n * sizeof(hc[0]), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(hd.data(), drawData, (size_t)n * 64, hipMemcpyDeviceToHost));
        }
        FILE* o = fopen(out_path, "wb");
        if (!o) { perror(out_path); return 73; }
        fwrite(&n, 4, 1, o); fwrite(&sc, 4, 1, o);
        fwrite(ind.getSharedOpaqueBucketSpans(), sizeof(vkr_bucket_span), sc, o);
        fwrite(hc.data(), sizeof(hc[0]), n, o); fwrite(hd.data(), 4, hd.size(), o);
        fclose(o);
        (void)hipFree(drawData); (void)hipFree(cmds);
        printf("ok indirect draws=%u spans=%u\n", n, sc);
        return 0;
    } catch (const std::exception& ex) {
        fprintf(stderr, "error: %s\n", ex.what());
        return 1;
    }
}

// `ibl in.bin out.bin`: EquirectToCubemap::convert + IblPrecompute::compute through the C++ wrapper.
// in: u32 {W, H, env, irr, pre, lut}, f32 equirect[H*W*4]     out: irradiance | prefilter | lut (raw halves)
static int run_ibl(const char* in_path, const char* out_path)
{
    FILE* f = fopen(in_path, "rb");
    if (!f) { perror(in_path); return 66; }
    uint32_t h[6];
    if (fread(h, 4, 6, f) != 6) return 65;
    std::vector<float> eq((size_t)h[0] * h[1] * 4);
    if (fread(eq.data(), 4, eq.size(), f) != eq.size()) return 65;
    fclose(f);
    try {
        vkr::HipContext context(0, 16, 16);
        const size_t nIrr = (size_t)6 * h[3] * h[3] * 4, nPre = vkr_ibl_prefilter_texels(h[4]) * 4, nLut = (size_t)h[5] * h[5] * 2;
        float *dEq = nullptr, *dCube = nullptr;
        uint16_t* dOut = nullptr;
        HIP_CHECK(hipMalloc((void**)&dEq, eq.size() * 4));
        HIP_CHECK(hipMalloc((void**)&dCube, (size_t)6 * h[2] * h[2] * 16));
        HIP_CHECK(hipMalloc((void**)&dOut, (nIrr + nPre + nLut) * 2));
        HIP_CHECK(hipMemcpy(dEq, eq.data(), eq.size() * 4, hipMemcpyHostToDevice));
        vkr::HipIblPrecompute::convert(context, dEq, h[0], h[1], dCube, h[2]);
        vkr::HipIblPrecompute::compute(context, dCube, h[2], dOut, dOut + nIrr, dOut + nIrr + nPre, h[3], h[4], h[5]);
        vkr::check(context.get(), vkr_sync(context.get(), nullptr), "vkr_sync");
        std::vector<uint16_t> out(nIrr + nPre + nLut);
        HIP_CHECK(hipMemcpy(out.data(), dOut, out.size() * 2, hipMemcpyDeviceToHost));
        FILE* o = fopen(out_path, "wb");
        if (!o) { perror(out_path); return 73; }
        fwrite(out.data(), 2, out.size(), o);
        fclose(o);
        (void)hipFree(dEq); (void)hipFree(dCube); (void)hipFree(dOut);
        bool threw = false;   // sizes that are not powers of two surface as the adapter's exception
        try { vkr::HipIblPrecompute::convert(context, dEq, h[0], h[1], dCube, 24); } catch (const std::runtime_error&) { threw = true; }
        if (!threw) { fprintf(stderr, "expected a bad cube size to throw\n"); return 3; }
        printf("ok ibl halves=%zu\n", out.size());
        return 0;
    } catch (const std::exception& ex) {
        fprintf(stderr, "error: %s\n", ex.what());
        return 1;
    }
}

int main(int argc, char** argv)
{
    if (argc >= 4 && std::strcmp(argv[1], "indirect") == 0) return run_indirect(argv[2], argv[3]);
    if (argc >= 4 && std::strcmp(argv[1], "ibl") == 0) return run_ibl(argv[2], argv[3]);
    if (argc < 3) { fprintf(stderr, "usage: %s scene.bin out.bin | indirect in.bin out.bin | ibl in.bin out.bin\n", argv[0]); return 64; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 66; }
    uint32_t hdr[7];
    if (fread(hdr, 4, 7, f) != 7) return 65;
    const uint32_t W = hdr[0], H = hdr[1], iters = hdr[2], frames = hdr[3], nverts = hdr[4], nidx = hdr[5], ninst = hdr[6];
    std::vector<float> verts((size_t)nverts * 3), xf((size_t)ninst * 12), ubos((size_t)frames * 120);
    std::vector<uint32_t> idx(nidx);
    if (fread(verts.data(), 4, verts.size(), f) != verts.size() || fread(idx.data(), 4, idx.size(), f) != idx.size() ||
        fread(xf.data(), 4, xf.size(), f) != xf.size() || fread(ubos.data(), 4, ubos.size(), f) != ubos.size()) return 65;
    fclose(f);
    try {
        vkr::HipContext context(0, W, H);
        vkr::HipRayTracingContext rt(context);
        vkr_mesh_desc mesh{};
        mesh.positions = verts.data(); mesh.position_stride = 12; mesh.vertex_count = nverts;
        mesh.indices = idx.data(); mesh.index_count = nidx; mesh.material_index = -1; mesh.opaque = 1;
        std::vector<vkr_instance> inst(ninst);
        for (uint32_t i = 0; i < ninst; ++i) { inst[i].mesh_index = 0; std::memcpy(inst[i].transform, &xf[(size_t)i * 12], 48); }
        vkr_scene_desc scene{};
        scene.meshes = &mesh; scene.mesh_count = 1; scene.instances = inst.data(); scene.instance_count = ninst;
        rt.init(scene);
        // a wrong instance count must surface as the reference's exception (RayTracingContext.cpp:51-53)
        bool threw = false;
        try { rt.updateTopLevelAS(std::vector<vkr_instance>(ninst + 1)); } catch (const std::runtime_error&) { threw = true; }
        if (!threw) { fprintf(stderr, "expected instance count mismatch to throw\n"); return 3; }

        uint16_t *depth = nullptr, *normal = nullptr, *full = nullptr;
        HIP_CHECK(hipMalloc((void**)&depth, (size_t)W * H * 2));
        HIP_CHECK(hipMalloc((void**)&normal, (size_t)W * H * 8));
        HIP_CHECK(hipMalloc((void**)&full, (size_t)W * H * 2));
        using Pass = vkr::HipRtaoPass<demo::RenderPass, demo::PassExecuteContext>;
        demo::Rendergraph graph;
        // Two streams, like a host whose prepass runs on the graphics queue and the RTAO pass on a compute stream: the pass's
        // beforePass / afterPass hooks (RtaoComputePass::beginPass / endPass barriers, RtaoComputePass.cpp:27-100,158-255) carry
        // the ordering between them.  Without the hooks the two passes race; with them the output is the single-stream one.
        hipStream_t gfx = nullptr, comp = nullptr;
        hipEvent_t evWritten = nullptr, evReleased = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&gfx, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&comp, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&evWritten, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&evReleased, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(evReleased, comp));                       // nothing to wait for before the first frame
        uint32_t nBefore = 0, nAfter = 0;
        Pass::Sync sync;
        sync.beforePass = [&](const demo::PassExecuteContext&, void* stream) {
            ++nBefore;
            if (hipStreamWaitEvent((hipStream_t)stream, evWritten, 0) != hipSuccess) throw std::runtime_error("hipStreamWaitEvent");
        };
        sync.afterPass = [&](const demo::PassExecuteContext&, void* stream) {
            ++nAfter;
            if (hipEventRecord(evReleased, (hipStream_t)stream) != hipSuccess) throw std::runtime_error("hipEventRecord");
        };
        // registered out of order on purpose: Compile() must put DepthPrepass first
        graph.AddPass(std::make_unique<Pass>(context, [&](const demo::PassExecuteContext& e) {
            vkr::RtaoFrameInputs in;
            in.ubo = e.ubo; in.linearDepth = depth; in.normal = normal; in.rtaoFull = full; in.stream = comp;
            in.atrousIterations = iters;
            return in;
        }, sync));
        graph.AddPass(std::make_unique<demo::GBufferPass>(context.get(), depth, normal, gfx, evWritten, evReleased));
        graph.Compile();
        auto names = graph.orderNames();
        if (names.size() != 2 || names[0] != "DepthPrepass" || names[1] != "RtaoComputePass") { fprintf(stderr, "bad pass order\n"); return 4; }
        for (uint32_t fr = 0; fr < frames; ++fr) {
            demo::PassExecuteContext e;
            e.ubo = &ubos[(size_t)fr * 120];
            e.imageIndex = fr & 1u;
            graph.Execute(e);
        }
        vkr::check(context.get(), vkr_sync(context.get(), comp), "vkr_sync");
        vkr::check(context.get(), vkr_sync(context.get(), gfx), "vkr_sync");
        if (nBefore != frames || nAfter != frames) { fprintf(stderr, "pass hooks ran %u/%u times for %u frames\n", nBefore, nAfter, frames); return 5; }
        std::vector<uint16_t> out((size_t)W * H);
        HIP_CHECK(hipMemcpy(out.data(), full, out.size() * 2, hipMemcpyDeviceToHost));
        FILE* o = fopen(argv[2], "wb");
        if (!o) { perror(argv[2]); return 73; }
        fwrite(out.data(), 2, out.size(), o);
        fclose(o);
        vkr_bvh_info info = rt.info();
        printf("ok %ux%u frames=%u triangles=%u nodes=%u hooks=%u+%u\n", W, H, frames, info.triangle_count, info.node_count, nBefore, nAfter);
        (void)hipFree(depth); (void)hipFree(normal); (void)hipFree(full);
        (void)hipEventDestroy(evWritten); (void)hipEventDestroy(evReleased); (void)hipStreamDestroy(gfx); (void)hipStreamDestroy(comp);
    } catch (const std::exception& ex) {
        fprintf(stderr, "error: %s\n", ex.what());
        return 1;
    }
    return 0;
}
