"""Fingerprint of the kernel sources a PMC model in profiles/ was measured on.  bench.py's roofline objects take their
instruction counts, traffic and busy fractions from that model (rocprofv3 --pmc cannot run inside the bench), so the model
carries this hash and bench.py / the CPU tests compare it with the tree: a kernel change that was not followed by
tools/refresh_r3.sh shows up as `model_fresh: false` (and a failing test) instead of a silently replayed fraction."""
import hashlib
import os
import re

_FILES = ("rtao_kernels.hip", "traverse.hpp", "device_math.hpp", "bvh_build.hip", "internal.hpp")


def _code_only(text: str) -> str:
    """The source without comments and white space: rewording a comment does not make a measured model stale.  (A `//`
    inside a string literal would be cut too; the kernels have none, and the only cost would be a coarser hash.)"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return "".join(text.split())


def kernel_source_fingerprint() -> str:
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for name in _FILES:
        with open(os.path.join(root, name), "r", encoding="utf-8") as f:
            h.update(name.encode() + b"\0" + _code_only(f.read()).encode())
    return h.hexdigest()[:16]
