"""The streaming kernels with 224-item LDS queues (a test-only build): node steps are throttled down to strict
depth-first order most of the time, and the results must not change by a bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "vulkanrenderer_amd", "lib", "libvkr_rt_smallq.so")


def test_small_queue_build_matches_oracle():
    assert os.path.exists(LIB), "libvkr_rt_smallq.so is not built (make -C vulkanrenderer_amd/csrc smallq)"
    env = dict(os.environ, VKR_LIB_PATH=LIB)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_smallq_worker.py")], env=env, capture_output=True,
                       text=True, timeout=200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "smallq ok" in r.stdout
