"""Two data-parallel ranks driving the product kernels (not an oracle stand-in) on the GPU box's one device.

Named to be collected first: the ranks are started as a fresh child process tree (torch.distributed.run)
before this process has touched the GPU, and all GPU work happens in the children (tests/dp_worker.py).
The exchange runs over gloo -- a one-GPU box has no second device for RCCL -- through the same
nerf/parallel.py calls the RCCL path uses.  The reference has no distributed code (train.py:98 picks one
device); what is pinned is the equivalence to its single-process step (train.py:179-183):
  * the 2-rank bucket after all_reduce_mean equals the 1-rank gradient of the concatenated batch to 1e-6;
  * after 3 Adam steps both ranks hold bit-identical parameters, equal to the single-process run."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_match_single_process(tmp_path):
    port = 29600 + os.getpid() % 300
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert torch.equal(r0["start"], r1["start"])                     # broadcast repaired rank 1's weights
    assert torch.equal(r0["draw"], r1["draw"])                       # shared stratified draw without a collective
    assert r0["grad_norm"] > 0
    assert r0["grad_err"] < 1e-6 and r1["grad_err"] < 1e-6, (r0["grad_err"], r1["grad_err"])
    assert torch.equal(r0["dp_params"], r1["dp_params"])             # ranks stay in lock step
    assert not torch.equal(r0["dp_params"], r0["start"])
    moved = (r0["single_params"] - r0["start"]).double().norm()
    diff = (r0["dp_params"] - r0["single_params"]).double().norm()
    # Adam's first steps move every weight by ~lr whatever its gradient's size, so a last-bit difference of a
    # near-zero gradient entry shows up at the size of the step: compare against the distance travelled
    assert float(diff / moved) < 2e-2, float(diff / moved)
