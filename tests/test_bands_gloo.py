"""Row-band sharding: host logic on CPU, and the N>1 path with world_size-2/3 gloo processes."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from vulkanrenderer_amd import bands

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("H,n", [(1080, 1), (1080, 2), (1080, 4), (1080, 8), (2160, 8), (37, 2), (37, 3), (5, 8), (1, 2)])
def test_band_rows_partition(H, n):
    rows = bands.split_rows(H, n)
    assert rows[0][0] == 0 and rows[-1][1] == H
    covered = np.zeros(H, dtype=np.int32)
    for (a, b) in rows:
        assert 0 <= a <= b <= H and b - a <= bands.band_height(H, n)
        covered[a:b] += 1
    assert np.all(covered == 1)
    for r in range(n - 1):
        assert rows[r][1] == rows[r + 1][0]


def test_atrous_halo():
    # taps reach 2*step rows per iteration with step 1,2,4,... (rtao_atrous.comp 5x5 kernel)
    assert [bands.atrous_halo(i) for i in (0, 1, 2, 3, 4, 5)] == [2, 2, 6, 14, 30, 62]


@pytest.mark.parametrize("H,n", [(37, 2), (37, 3), (40, 4), (5, 8)])
def test_pack_unpack_roundtrip_single_process(H, n):
    W = 11
    g = torch.Generator().manual_seed(H * 100 + n)
    planes = [torch.randint(-32768, 32767, (H, W), dtype=torch.int16, generator=g) for _ in range(bands.PLANES)]
    hb = bands.band_height(H, n)
    gathered = torch.zeros((n, bands.PLANES, hb, W), dtype=torch.int16)
    for r in range(n):
        bands.pack_band(planes, H, n, r, gathered[r])
    out = [torch.zeros((H, W), dtype=torch.int16) for _ in range(bands.PLANES)]
    bands.unpack_frame(gathered, out, H, n)
    for a, b in zip(planes, out):
        assert torch.equal(a, b)


def test_all_gather_single_process_is_copy():
    packed = torch.arange(3 * 4 * 5, dtype=torch.int16).view(3, 4, 5)
    gathered = torch.zeros((1, 3, 4, 5), dtype=torch.int16)
    bands.all_gather_frame(packed, gathered)
    assert torch.equal(gathered[0], packed)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("H,n", [(37, 2), (37, 3), (2160, 8), (1080, 4), (5, 8), (64, 8), (8, 1)])
def test_stripe_plan_roundtrip(H, n):
    """Interleaved 8-row stripes: every image row travels exactly once, ragged ends included."""
    W = 5
    plan = bands.StripePlan(H, n)
    assert plan.slot_rows == bands.stripes_per_rank(H, n) * bands.STRIPE
    full = torch.arange(H * W, dtype=torch.int32).view(H, W).to(torch.int16)
    g = torch.zeros((n, plan.slot_rows, W), dtype=torch.int16)
    owner = (torch.arange(H) // bands.STRIPE) % n
    for r in range(n):
        mine = torch.where((owner == r)[:, None], full, torch.full_like(full, -7))     # rank r only has its own rows
        plan.pack(mine, r, g[r])
    out = torch.zeros_like(full)
    plan.unpack(g, out)
    assert torch.equal(out, full)


@pytest.mark.parametrize("world,schedule", [(2, "bands"), (3, "bands"), (2, "stripes"), (3, "stripes"), (2, "capi"), (3, "capi")])
def test_band_schedule_over_gloo_matches_full_frame(world, schedule, tmp_path, oracle_mod):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="2")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_band_worker.py"), str(tmp_path), schedule], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-2000:]}"
        assert (tmp_path / f"rank{r}.txt").read_text() == "ok"
