"""Row-band sharding of the frame over N ranks (no reference counterpart -- the reference is
single-GPU, Renderer.cpp:359).

Rank r of N owns rows [r*Hb, (r+1)*Hb) with Hb = ceil(H/N) (the last band may be short; buffers are padded to Hb
so one fixed-size all-gather assembles the frame).  Each rank traces/filters its band plus the A-Trous halo
(vkr_rtao_execute_band), then ONE all-gather per frame moves a packed R16F buffer
    [history band | rtao_full band | sun-shadow band]            (6 B/px)
so every rank holds the full previous history (temporal reprojection is non-local) and the full outputs.
Device-agnostic torch code: runs over RCCL (backend "nccl") on GPUs and over gloo on CPU in the tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist

PLANES = 3  # history, rtao_full, sun shadow (all R16F bit patterns in int16)


def band_height(H: int, n: int) -> int:
    return (H + n - 1) // n


def band_rows(H: int, n: int, r: int) -> Tuple[int, int]:
    hb = band_height(H, n)
    y0 = min(H, r * hb)
    return y0, min(H, y0 + hb)


def atrous_halo(iterations: int) -> int:
    """Rows of source needed beyond the band for N A-Trous iterations: sum 2*2^i = 2*(2^N - 1)."""
    n = max(1, int(iterations))
    return 2 * ((1 << n) - 1)


def pack_band(planes: Sequence[torch.Tensor], H: int, n: int, r: int, out: torch.Tensor) -> torch.Tensor:
    """Copy rows [y0,y1) of each full-frame int16 plane (H, W) into out (PLANES, Hb, W); pad rows stay as they are."""
    y0, y1 = band_rows(H, n, r)
    for k, p in enumerate(planes):
        out[k, : y1 - y0].copy_(p[y0:y1])
    return out


def all_gather_frame(packed: torch.Tensor, gathered: torch.Tensor, group=None, async_op: bool = False):
    """packed (PLANES, Hb, W) from every rank -> gathered (n, PLANES, Hb, W).  One collective.
    The payload travels as raw bytes: neither RCCL nor gloo has a 16-bit integer type, and no reduction is done.
    Returns a handle whose wait() orders the current stream after the collective (see all_gather_bytes)."""
    return all_gather_bytes(packed, gathered, group=group, async_op=async_op)


def unpack_frame(gathered: torch.Tensor, planes: Sequence[torch.Tensor], H: int, n: int) -> None:
    """Scatter gathered (n, PLANES, Hb, W) back into the full-frame planes (each (H, W) int16)."""
    hb = gathered.shape[2]
    full_bands = H // hb
    for k, p in enumerate(planes):
        if full_bands:
            p[: full_bands * hb].view(full_bands, hb, -1).copy_(gathered[:full_bands, k])
        if full_bands < n and full_bands * hb < H:
            rem = H - full_bands * hb
            p[full_bands * hb:].copy_(gathered[full_bands, k, :rem])


# ---------------------------------------------------------------------------------------------------------------
# Two-phase schedule: K1 on interleaved 8-row stripes (stripe s belongs to rank s % n), one all-gather of the history
# stripes, A-Trous / shadows on contiguous bands, one all-gather of [rtao_full | sun].
# ---------------------------------------------------------------------------------------------------------------
STRIPE = 8


def stripes_per_rank(H: int, n: int) -> int:
    """Stripes each rank's gather slot holds (padded: ranks past the ragged end own one stripe less)."""
    total = (H + STRIPE - 1) // STRIPE
    return (total + n - 1) // n


class StripePlan:
    """Row bookkeeping of the interleaved-stripe exchange for an (H, W) plane over n ranks.  Rank r owns the rows of
    stripes r, r+n, ...; every rank's gather slot holds k = stripes_per_rank stripes (ragged ends padded: padded rows
    are sent as copies of the last image row and dropped on arrival)."""

    def __init__(self, H: int, n: int, device=None):
        k = stripes_per_rank(H, n)
        rows = (torch.arange(k)[None, :, None] * n + torch.arange(n)[:, None, None]) * STRIPE + torch.arange(STRIPE)[None, None, :]
        rows = rows.reshape(n, k * STRIPE)                       # [rank, slot row] -> image row
        valid = rows < H
        self.H, self.n, self.k = H, n, k
        self.slot_rows = k * STRIPE
        self.rows_clamped = rows.clamp(max=H - 1).to(device)
        self.src = torch.nonzero(valid.reshape(-1)).squeeze(1).to(device)      # positions in the gathered (n*k*8) rows
        self.dst = rows.reshape(-1)[valid.reshape(-1)].to(device)               # ... and the image rows they fill
        assert self.dst.numel() == H and int(self.dst.sort().values.ne(torch.arange(H, device=self.dst.device)).sum()) == 0

    def pack(self, plane: torch.Tensor, r: int, out: torch.Tensor) -> torch.Tensor:
        """Rank r's rows of plane (H, W) -> out (k*8, W)."""
        torch.index_select(plane, 0, self.rows_clamped[r], out=out)
        return out

    def unpack(self, gathered: torch.Tensor, plane: torch.Tensor) -> None:
        """gathered (n, k*8, W) -> plane (H, W): every row exactly once."""
        plane.index_copy_(0, self.dst, gathered.view(-1, gathered.shape[-1]).index_select(0, self.src))


class _Done:
    def wait(self):
        return True


def all_gather_bytes(send: torch.Tensor, recv: torch.Tensor, group=None, async_op: bool = False):
    """recv (n, *send.shape) <- every rank's send; payload as raw bytes (see all_gather_frame).  Returns a handle whose
    wait() orders the current stream after the collective; with async_op=True (RCCL only) the collective runs on the
    communicator's stream while the caller keeps enqueuing independent kernels."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if send.is_cuda and dist.get_backend(group) == "gloo":
            h_out = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(h_out.view(-1).view(torch.uint8), send.contiguous().cpu().view(-1).view(torch.uint8), group=group)
            recv.copy_(h_out)
            return _Done()
        work = dist.all_gather_into_tensor(recv.view(-1).view(torch.uint8), send.contiguous().view(-1).view(torch.uint8),
                                           group=group, async_op=async_op)
        return work if async_op and work is not None else _Done()
    recv.view(-1)[: send.numel()].copy_(send.reshape(-1))
    return _Done()


def split_rows(H: int, n: int) -> List[Tuple[int, int]]:
    return [band_rows(H, n, r) for r in range(n)]
