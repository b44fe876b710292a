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

// stands in for DepthPrepass: outputs {"depth","normal"}
class GBufferPass : public RenderPass {
public:
    GBufferPass(vkr_ctx* c, uint16_t* depth, uint16_t* normal)
        : RenderPass("DepthPrepass", {}, {"depth", "normal"}), ctx(c), d(depth), n(normal) {}

protected:
    void beginPass(const PassExecuteContext&) override {}
    void render(const PassExecuteContext& e) override
    {
        vkr::check(ctx, vkr_gbuffer_primary(ctx, e.ubo, d, n, nullptr, nullptr, nullptr, e.stream), "vkr_gbuffer_primary");
    }
    void endPass(const PassExecuteContext&) override {}
    vkr_ctx* ctx;
    uint16_t *d, *n;
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

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]); return 64; }
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
        // registered out of order on purpose: Compile() must put DepthPrepass first
        graph.AddPass(std::make_unique<Pass>(context, [&](const demo::PassExecuteContext& e) {
            vkr::RtaoFrameInputs in;
            in.ubo = e.ubo; in.linearDepth = depth; in.normal = normal; in.rtaoFull = full; in.stream = e.stream;
            in.atrousIterations = iters;
            return in;
        }));
        graph.AddPass(std::make_unique<demo::GBufferPass>(context.get(), depth, normal));
        graph.Compile();
        auto names = graph.orderNames();
        if (names.size() != 2 || names[0] != "DepthPrepass" || names[1] != "RtaoComputePass") { fprintf(stderr, "bad pass order\n"); return 4; }
        for (uint32_t fr = 0; fr < frames; ++fr) {
            demo::PassExecuteContext e;
            e.ubo = &ubos[(size_t)fr * 120];
            e.imageIndex = fr & 1u;
            graph.Execute(e);
        }
        vkr::check(context.get(), vkr_sync(context.get(), nullptr), "vkr_sync");
        std::vector<uint16_t> out((size_t)W * H);
        HIP_CHECK(hipMemcpy(out.data(), full, out.size() * 2, hipMemcpyDeviceToHost));
        FILE* o = fopen(argv[2], "wb");
        if (!o) { perror(argv[2]); return 73; }
        fwrite(out.data(), 2, out.size(), o);
        fclose(o);
        vkr_bvh_info info = rt.info();
        printf("ok %ux%u frames=%u triangles=%u nodes=%u\n", W, H, frames, info.triangle_count, info.node_count);
        (void)hipFree(depth); (void)hipFree(normal); (void)hipFree(full);
    } catch (const std::exception& ex) {
        fprintf(stderr, "error: %s\n", ex.what());
        return 1;
    }
    return 0;
}
