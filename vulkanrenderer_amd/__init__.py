"""MI355X-native RTAO / shadow-ray path of chenjunjia-java/VulkanRenderer (HIP LBVH + traversal + denoise)
behind the reference's RenderPass interface.  See DESIGN.md."""
from . import bands
from .renderer import (Context, GBuffer, PassExecuteContext, RayTracingContext, Rendergraph, RenderPass, RenderStats,
                       RtaoComputePass, ShadowRays, SharedOpaqueIndirect, EquirectToCubemap, IblPrecompute, IblResult, generate_gbuffer, trace_stats, trace_stats_enable)
from .scenes import (SceneData, Mesh, Material, FrameState, make_bistro_class, make_sponza_class, make_test_scene,
                     make_material_matrix_scene, make_street_scene, make_scene_graph, make_draw_model, make_sky_hdr, camera_path)

__all__ = ["SceneData", "Mesh", "Material", "FrameState", "make_bistro_class", "make_sponza_class",
           "make_test_scene", "make_material_matrix_scene", "make_street_scene", "make_scene_graph", "make_draw_model", "make_sky_hdr", "camera_path", "bands",
           "Context", "GBuffer", "PassExecuteContext", "RayTracingContext", "Rendergraph", "RenderPass", "RenderStats",
           "RtaoComputePass", "ShadowRays", "SharedOpaqueIndirect", "EquirectToCubemap", "IblPrecompute", "IblResult", "generate_gbuffer", "trace_stats", "trace_stats_enable"]
