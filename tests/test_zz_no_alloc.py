"""vkr_shadow_sun_noalloc (the entry point that allocates nothing: its scratch is a fixed device array inside the library)
against vkr_shadow_sun, through the C ABI: same bits on two calls in a row, the second with a smaller input; sizes beyond
the stated maximum are rejected with a clear error and nothing is launched."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_sun_shadows_match_the_allocating_entry_point():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from vulkanrenderer_amd import FrameState, _capi, make_street_scene
    from vulkanrenderer_amd.renderer import Context, GBuffer, RayTracingContext, generate_gbuffer
    from vulkanrenderer_amd.scenes import UBO_DIR_LIGHT_PARAMS
    scene = make_street_scene(5, 0.02, "noalloc")
    W, H = 320, 180
    ctx = Context(W, H)
    lib, h = ctx.lib, ctx.handle
    rt = RayTracingContext(ctx)
    rt.init(scene)
    gb = GBuffer(ctx, with_shading_inputs=True)
    fs = FrameState(W, H, ray_count=4, sun_samples=16)
    ubo = fs.update(*scene.cameras[0])
    generate_gbuffer(ctx, ubo, gb)
    ctx.sync()
    assert int(gb.valid.sum().item()) > 1000
    L = (C.c_float * 3)(*[float(x) for x in FrameState.sun_dir_to_light(ubo)])
    radius, samples = float(ubo[UBO_DIR_LIGHT_PARAMS]), float(ubo[UBO_DIR_LIGHT_PARAMS + 1])

    def run(fn, out, first_row, rows, n_samples):
        off = first_row * W
        rc = fn(h, gb.world_pos.data_ptr() + off * 12, gb.geom_normal.data_ptr() + off * 12, gb.valid.data_ptr() + off,
                rows * W, L, radius, float(n_samples), out.data_ptr() + off * 4, ctx.stream())
        assert rc == _capi.VKR_OK, lib.vkr_last_error(h)
        ctx.sync()

    # two calls in a row on the same GPU array: the whole frame with 16 samples, then a smaller input (60 rows, 8 samples)
    for first_row, rows, n_samples in ((0, H, samples), (40, 60, 8.0), (0, H, 1.0)):
        want = torch.full((H, W), -1.0, dtype=torch.float32, device=ctx.device)
        got = torch.full((H, W), -1.0, dtype=torch.float32, device=ctx.device)
        run(lib.vkr_shadow_sun_noalloc, got, first_row, rows, n_samples)
        run(lib.vkr_shadow_sun, want, first_row, rows, n_samples)
        g, w = got.cpu().numpy(), want.cpu().numpy()
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), (first_row, rows, n_samples)     # bit-exact, untouched rows included
        assert (w[first_row:first_row + rows] >= 0.0).all() and (w[first_row:first_row + rows] < 1.0).any()
    # beyond the stated maximum: a clear error, nothing launched
    rc = lib.vkr_shadow_sun_noalloc(h, gb.world_pos.data_ptr(), gb.geom_normal.data_ptr(), gb.valid.data_ptr(),
                                    (1 << 28) + 1, L, radius, samples, got.data_ptr(), ctx.stream())
    assert rc == _capi.VKR_ERR_LIMIT and b"maximum" in lib.vkr_last_error(h)
    ctx.close()
