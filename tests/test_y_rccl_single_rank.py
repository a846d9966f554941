"""RCCL on the GPU box with a world of one rank (the box has one GPU, and RCCL refuses two ranks on one device):
library initialisation plus the all-reduce / broadcast calls nerf/parallel.py issues, on the real bucket, in a
fresh child process.  The multi-rank arithmetic is covered by tests/test_a_dp_gpu.py (gloo)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_single_rank_bucket_round_trip():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], env=env, capture_output=True, text=True,
                         timeout=240)
    assert res.returncode == 0 and "rccl single rank ok" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
