"""Test-only builds of the library (make -C vulkanrenderer_amd/csrc variants), both with device-side bound checks on every
queue / ray index of k_trace_q: (1) 224-item LDS queues -- node steps are throttled down to strict depth-first order and
spill to the global overflow stack most of the time; (2) K1 compiled for 7 waves per SIMD, where the register allocator
has to spill (round 1 recorded an undiagnosed GPU fault for its 7-wave build).  Results must not change by a bit and no
index may leave its array."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["smallq", "w7dbg"])
def test_variant_build_matches_oracle_and_stays_in_bounds(variant):
    lib = os.path.join(ROOT, "vulkanrenderer_amd", "lib", f"libvkr_rt_{variant}.so")
    assert os.path.exists(lib), f"{lib} is not built (make -C vulkanrenderer_amd/csrc variants)"
    env = dict(os.environ, VKR_LIB_PATH=lib, VKR_PHASE_DUMP="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_smallq_worker.py")], env=env, capture_output=True,
                       text=True, timeout=200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "smallq ok" in r.stdout
    bits = re.findall(r'"bound_violation_bits": (\d+)', r.stderr)
    assert bits and all(int(b) == 0 for b in bits), r.stderr[-2000:]
