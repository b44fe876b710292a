"""Fingerprint of the kernel sources a PMC model in profiles/ was measured on.  bench.py's roofline objects take their
instruction counts, traffic and busy fractions from that model (rocprofv3 --pmc cannot run inside the bench), so the model
carries this hash and bench.py / the CPU tests compare it with the tree: a kernel change that was not followed by
tools/refresh_r3.sh shows up as `model_fresh: false` (and a failing test) instead of a silently replayed fraction."""
import hashlib
import os

_FILES = ("rtao_kernels.hip", "traverse.hpp", "device_math.hpp", "bvh_build.hip", "internal.hpp")


def kernel_source_fingerprint() -> str:
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for name in _FILES:
        with open(os.path.join(root, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]
