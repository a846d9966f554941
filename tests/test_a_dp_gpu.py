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


@pytest.mark.parametrize("optimizer", ["fused", "torch"])
def test_train_driver_two_ranks_match_one(tmp_path, optimizer):
    """nerf-pytorch_amd/train.py under torch.distributed.run with two ranks (gloo, both on the box's one GPU) against the
    same script as a single process: the ranks draw the same batches and each renders half of them, so after four
    iterations rank 0's checkpoint must equal the single-process one up to summation order.  Covers the driver's own
    data-parallel code: parameter broadcast, shared draws, full batches, the bucket all-reduce (fused optimizer) and the
    flattened per-tensor all-reduce (torch.optim.Adam)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_callers import make_dataset, write_config
    root = make_dataset(str(tmp_path / "ds"))
    train_py = os.path.join(ROOT, "nerf-pytorch_amd", "train.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    # common starting point: the checkpoint of a zero-iteration run (process-group start-up may consume host random
    # numbers, so the two runs would not initialise their networks alike on their own)
    cfg0 = str(tmp_path / "cfg0.txt")
    write_config(cfg0, root, str(tmp_path / "models0"), "siren")   # (a default NeRF renders empty space here: zero gradients)
    res = subprocess.run([sys.executable, train_py, "--config", cfg0, "--optimizer", optimizer, "--max-iters", "0"], env=env,
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    init = str(tmp_path / "models0" / "siren_model_000000.pth")
    def one_chunk(path):
        # one ray chunk per batch on every rank: the renderer draws one stratified vector PER CHUNK (rendering.py:191-195),
        # so a batch that is chunked differently on one rank than in the single process is a different sample
        text = open(path).read().replace("chunk_size = 64", "chunk_size = 128")
        open(path, "w").write(text)

    flatten = lambda path: torch.cat([v.reshape(-1) for v in torch.load(path, map_location="cpu", weights_only=True)[
        "model_state_dict"].values()]).double()
    finals = {}
    for world in (1, 2):
        save = str(tmp_path / f"models{world}")
        cfgp = str(tmp_path / f"cfg{world}.txt")
        write_config(cfgp, root, save, "siren")
        one_chunk(cfgp)
        common = [train_py, "--config", cfgp, "--optimizer", optimizer, "--max-iters", "4", "--resume", init]
        if world == 1:
            cmd = [sys.executable] + common
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                   "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200)] + common + ["--dist-backend", "gloo",
                                                                                            "--one-device"]
        res = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=420)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
        assert "Training complete" in res.stdout
        finals[world] = flatten(os.path.join(save, "siren_model_000004.pth"))
    moved = (finals[1] - flatten(init)).norm()             # distance travelled in four iterations
    assert float(moved) > 0
    # Adam's first steps move every weight by about lr whatever its gradient's size, so last-bit differences of
    # near-zero gradient entries show at the size of a step: compare with the distance travelled
    assert float((finals[2] - finals[1]).norm() / moved) < 5e-2, float((finals[2] - finals[1]).norm() / moved)
