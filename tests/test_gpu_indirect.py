"""GPU parity of the indirect build (BASELINE configs[2]) through the C ABI: drawData,
DrawIndexedIndirectCommand[], bucket spans and draw count bit-exact against the oracle's restatement of
FrameManager::prepareSharedOpaqueIndirect (FrameManager.cpp:118-175)."""
import numpy as np
import pytest

from oracle import oracle
from vulkanrenderer_amd.scenes import make_draw_model, look_at_rh, perspective_rh_zo_yflip, col_major

pytestmark = pytest.mark.gpu

I16 = np.eye(4, dtype=np.float32).reshape(16)
MODEL = col_major(np.array([[0.8, 0, 0.6, 3], [0, 1.25, 0, -2], [-0.6, 0, 0.8, 7], [0, 0, 0, 1]], dtype=np.float32))


def view_proj(eye=(0, 8, 70), target=(0, 0, 0), fov=55.0):
    return col_major((perspective_rh_zo_yflip(fov, 16 / 9, 0.1, 400.0) @ look_at_rh(eye, target)).astype(np.float32))


@pytest.fixture(scope="module")
def ctx():
    from vulkanrenderer_amd import Context
    c = Context(64, 36)
    yield c
    c.close()


def run_both(ctx, m, model, max_draws, vp=None, bounds=True):
    from vulkanrenderer_amd import SharedOpaqueIndirect
    ib = SharedOpaqueIndirect(ctx)
    ib.buildSharedDrawSlots(m["node_parent"], m["node_local"], m["slots"], m["mesh_infos"], max_draws,
                            mesh_bounds=m["mesh_bounds"] if bounds else None)
    want_slots = oracle.sort_draw_slots(m["slots"])
    got_slots = ib.sortedSlots()
    assert got_slots.tobytes() == want_slots.tobytes()
    ib.prepareSharedOpaqueIndirect(model, vp)
    n, dd, cm, sp = ib.download()
    wn, wdd, wcm, wsp = oracle.prepare_indirect(m["node_parent"], m["node_local"], want_slots, m["mesh_infos"], model,
                                                max_draws, view_proj=vp, mesh_bounds=m["mesh_bounds"] if vp is not None else None)
    assert n == wn
    assert sp.tobytes() == wsp.tobytes()
    assert cm.tobytes() == wcm.tobytes()
    assert dd.tobytes() == wdd.tobytes()
    return ib, n


@pytest.mark.parametrize("nodes,meshes,mats,seed", [(1, 1, 1, 1), (7, 3, 2, 2), (300, 20, 8, 3), (5000, 200, 64, 4), (70000, 3000, 256, 5)])
def test_prepare_matches_reference_loop(ctx, nodes, meshes, mats, seed):
    m = make_draw_model(nodes, meshes, mats, seed=seed)
    run_both(ctx, m, MODEL, 1 << 20)


def test_deep_and_interleaved_hierarchies(ctx):
    """depth > 16 takes the one-launch chain-walk kernel; a DFS-ordered hierarchy (levels interleaved in linear order)
    exercises the level permutation of the level-per-launch kernel."""
    deep = make_draw_model(6000, 50, 8, seed=21, max_depth=40)
    run_both(ctx, deep, MODEL, 1 << 16)
    m = make_draw_model(5000, 50, 8, seed=22, max_depth=6)
    # re-number the nodes in DFS pre-order: parents still precede children, levels interleave
    n = len(m["node_parent"])
    kids = [[] for _ in range(n)]
    roots = []
    for i, p in enumerate(m["node_parent"]):
        (roots if p < 0 else kids[p]).append(i)
    dfs, stack = [], roots[::-1]
    while stack:
        i = stack.pop()
        dfs.append(i)
        stack.extend(kids[i][::-1])
    new_of = np.empty(n, dtype=np.int64)
    new_of[dfs] = np.arange(n)
    par = np.array([-1 if m["node_parent"][i] < 0 else new_of[m["node_parent"][i]] for i in dfs], dtype=np.int32)
    assert (par < np.arange(n)).all() and (np.diff(par) < 0).any()
    m2 = dict(m, node_parent=par, node_local=m["node_local"][dfs])
    m2["slots"] = m["slots"].copy()
    m2["slots"]["node_linear_index"] = new_of[m["slots"]["node_linear_index"]].astype(np.uint32)
    run_both(ctx, m2, MODEL, 1 << 16, vp=view_proj())


def test_many_buckets(ctx):
    """Thousands of (doubleSided, material) buckets: the span pass walks them in chunks of 1024 with a running offset;
    a bucket per slot is the extreme."""
    m = make_draw_model(40000, 9000, 8000, seed=23)
    keys = len(np.unique(m["slots"]["mat_index"].astype(np.uint64) * 2 + m["slots"]["double_sided"]))
    assert keys > 4096
    run_both(ctx, m, MODEL, 1 << 17)
    run_both(ctx, m, MODEL, 1 << 17, vp=view_proj())
    run_both(ctx, m, MODEL, 3001, vp=view_proj())
    one_each = make_draw_model(3000, 5000, 5000, seed=24)
    one_each["slots"]["mat_index"] = np.arange(len(one_each["slots"]), dtype=np.uint32)[::-1]
    run_both(ctx, one_each, MODEL, 1 << 17, vp=view_proj())


def test_invalid_slots_are_skipped(ctx):
    m = make_draw_model(9000, 100, 16, seed=11, invalid_fraction=0.07)
    ib, n = run_both(ctx, m, I16, 1 << 20)
    assert n < len(m["slots"])


@pytest.mark.parametrize("max_draws", [0, 1, 63, 64, 2048, 2049, 7777])
def test_max_draws_truncation(ctx, max_draws):
    m = make_draw_model(9000, 100, 16, seed=12, invalid_fraction=0.03)
    ib, n = run_both(ctx, m, MODEL, max_draws)
    assert n == max(1, max_draws)


def test_untouched_tail_and_repeat_calls(ctx):
    m = make_draw_model(4000, 64, 16, seed=13)
    ib, n = run_both(ctx, m, MODEL, 1 << 14)
    ib.draw_data.fill_(-7.0)
    ib.commands.fill_(0xAB)
    vp = view_proj()
    ib.prepareSharedOpaqueIndirect(I16, vp)
    n2, dd, cm, sp = ib.download()
    assert 0 < n2 < n
    assert bool((ib.draw_data[n2:] == -7.0).all()) and bool((ib.commands[n2:] == 0xAB).all())
    wn, wdd, wcm, wsp = oracle.prepare_indirect(m["node_parent"], m["node_local"], oracle.sort_draw_slots(m["slots"]),
                                                m["mesh_infos"], I16, 1 << 14, view_proj=vp, mesh_bounds=m["mesh_bounds"])
    assert (n2, dd.tobytes(), cm.tobytes(), sp.tobytes()) == (wn, wdd.tobytes(), wcm.tobytes(), wsp.tobytes())


@pytest.mark.parametrize("eye,target,fov", [((0, 8, 70), (0, 0, 0), 55.0), ((0, 2, 0), (30, 0, 5), 30.0), ((500, 0, 0), (600, 0, 0), 40.0),
                                            ((0, 300, 0.01), (0, 0, 0), 90.0)])
def test_frustum_cull(ctx, eye, target, fov):
    m = make_draw_model(30000, 500, 64, seed=14, invalid_fraction=0.01)
    run_both(ctx, m, MODEL, 1 << 20, vp=view_proj(eye, target, fov))


def test_empty_model_and_errors(ctx):
    import torch
    from vulkanrenderer_amd import Context, SharedOpaqueIndirect
    fresh = Context(16, 16)
    ib0 = SharedOpaqueIndirect(fresh)
    ib0.draw_data = torch.zeros((4, 16), device=fresh.device)
    ib0.commands = torch.zeros((4, 20), dtype=torch.uint8, device=fresh.device)
    with pytest.raises(RuntimeError, match="before vkr_indirect_upload"):
        ib0.prepareSharedOpaqueIndirect(I16)
    fresh.close()
    ib = SharedOpaqueIndirect(ctx)
    empty = make_draw_model(3, 2, 1, seed=1)
    empty["slots"] = empty["slots"][:0]
    ib2, n = run_both(ctx, empty, I16, 16)
    assert n == 0
    bad = make_draw_model(10, 2, 1, seed=1)
    bad["node_parent"][3] = 5
    with pytest.raises(RuntimeError, match="must precede"):
        ib.buildSharedDrawSlots(bad["node_parent"], bad["node_local"], bad["slots"], bad["mesh_infos"], 8)
    ok = make_draw_model(10, 2, 1, seed=1)
    ib.buildSharedDrawSlots(ok["node_parent"], ok["node_local"], ok["slots"], ok["mesh_infos"], 8)
    with pytest.raises(RuntimeError, match="no mesh_bounds"):
        ib.prepareSharedOpaqueIndirect(I16, view_proj())
    # output arrays smaller than min(max_draws, slot_count): refused on the host, nothing launched
    ib.draw_data = ib.draw_data[:4]
    with pytest.raises(RuntimeError, match="capacity"):
        ib.prepareSharedOpaqueIndirect(I16)


def test_one_million_instances(ctx):
    """BASELINE configs[2] at full size: bit-exact against the oracle (0.2 s of CPU) with and without the frustum."""
    m = make_draw_model(1_000_000, 20_000, 256, seed=15, invalid_fraction=0.005)
    assert len(m["slots"]) > 1_000_000
    run_both(ctx, m, MODEL, 1 << 21)
    run_both(ctx, m, MODEL, 1 << 21, vp=view_proj((0, 40, 160), (0, 0, 0), 60.0))
    run_both(ctx, m, MODEL, 500_000, vp=view_proj((0, 40, 160), (0, 0, 0), 60.0))
