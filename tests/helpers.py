"""Shared helpers of the parity tests: run the same seeded inputs through the CPU oracle and the HIP path."""
import numpy as np


def h2f(a_u16: np.ndarray) -> np.ndarray:
    return a_u16.view(np.float16).astype(np.float32)


def compare_r16f(gpu_u16: np.ndarray, ref_u16: np.ndarray):
    """-> dict(bit_mismatch, frac_gt_1_255, max_abs)"""
    g, r = h2f(gpu_u16), h2f(ref_u16)
    diff = np.abs(g - r)
    return {"bit_mismatch": int((gpu_u16 != ref_u16).sum()), "frac_gt_1_255": float((diff > 1.0 / 255.0).mean()),
            "max_abs": float(diff.max()) if diff.size else 0.0, "n": int(diff.size)}


def to_np_u16(t):
    return t.detach().cpu().numpy().view(np.uint16)
