"""GPU box: build BVHs with the treelet pass under VKR_VALIDATE_TREE=1 (host-side structural check before the fit and
any traversal runs), small to large scenes.  No rays are traced here."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["VKR_VALIDATE_TREE"] = "1"
import numpy as np, torch
from vulkanrenderer_amd import make_test_scene, make_street_scene, make_sponza_class, make_material_matrix_scene
from vulkanrenderer_amd.renderer import Context, RayTracingContext
for name, sc in [("matrix", make_material_matrix_scene()), ("test", make_test_scene()), ("street0.02", make_street_scene(21, 0.02, "b")),
                 ("sponza", make_sponza_class())]:
    ctx = Context(64, 64)
    rt = RayTracingContext(ctx)
    t = time.time()
    rt.init(sc)
    ctx.sync()
    i = rt.info()
    print(name, "ok: tris", i.triangle_count, "wide nodes", i.node_count, "levels", i.height, "binary height", i.binary_height, f"{time.time()-t:.2f}s", flush=True)
    ctx.close()
