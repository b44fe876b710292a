"""Development aid: build (and refit) the C3 scene's BVH a few times (for `rocprofv3 --kernel-trace --stats` of the build kernels)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (device runtime)

from vulkanrenderer_amd import Context, RayTracingContext, make_bistro_class

scene = make_bistro_class()
ctx = Context(1920, 1080, 0)
rt = RayTracingContext(ctx)
rt.init(scene)
ctx.sync()
for _ in range(3):
    rt.rebuild()
ctx.sync()
for _ in range(3):
    rt.updateTopLevelAS(scene.instance_mesh, scene.instance_transform)
ctx.sync()
info = rt.info()
print("wide nodes", info.node_count, "levels", info.height, "binary height", info.binary_height)
