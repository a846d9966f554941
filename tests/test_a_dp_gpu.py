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


def test_two_ranks_headline_shape_match_single_process(tmp_path):
    """The shape the headline rate is quoted on, with two ranks: coarse + fine pair, ONE bucket over the ModuleDict,
    two-stream backward, one all-reduce, one fused Adam launch (tests/dp_worker_hier.py).
      * each network's slice of the bucket after all_reduce_mean equals the 1-rank gradient of the concatenated
        batch to 1e-6 (the step the reference takes in one process, train.py:179-183);
      * the 1/world factor that rides in the Adam kernel is applied: exp_avg after the first step is
        (1 - beta1) * mean gradient to 1e-6 (Adam's update itself is nearly scale-invariant and would hide it);
      * both ranks hold bit-identical parameters after 3 steps, equal to the single-process run up to
        summation order."""
    port = 29300 + os.getpid() % 300
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker_hier.py"), str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert torch.equal(r0["start"], r1["start"]) and torch.equal(r0["draw"], r1["draw"])
    assert min(r0["grad_norm"]) > 0
    for r in (r0, r1):
        assert max(r["grad_err"]) < 1e-6, r["grad_err"]              # coarse slice, fine slice
        assert r["exp_avg_err"] < 1e-6, r["exp_avg_err"]
    assert torch.equal(r0["dp_params"], r1["dp_params"])
    assert not torch.equal(r0["dp_params"], r0["start"])
    moved = (r0["single_params"] - r0["start"]).double().norm()
    diff = (r0["dp_params"] - r0["single_params"]).double().norm()
    assert float(diff / moved) < 2e-2, float(diff / moved)           # see test_two_ranks_match_single_process


def test_bench_two_ranks_on_one_device(tmp_path):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per process), rehearsed on the
    box's one GPU over gloo: the line must be the whole job's (n_gpus 2, 8192 rays per step, weak scaling) with the
    roofline object in place."""
    import json
    port = 29000 + os.getpid() % 300
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
           "--one-device", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                       # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["rays_per_step"] == 8192 and line["config"]["rays_per_gpu"] == 4096
    assert line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1 and line["dtype"] == "f32"
    assert line["value"] > 0 and abs(line["value"] - 8192 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1
    assert "dp2" in line["config"]["parallelism"]


def test_eval_two_ranks_write_the_single_process_frames(tmp_path):
    """eval.py's pose sharding (no collective: every rank renders its own poses and writes its own frames) with two
    ranks on the box's one device: the union of the frames equals the single-process frames bit for bit."""
    import numpy as np
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_callers import make_dataset, write_config
    root = make_dataset(str(tmp_path / "ds"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    train_py = os.path.join(ROOT, "nerf-pytorch_amd", "train.py")
    eval_py = os.path.join(ROOT, "nerf-pytorch_amd", "eval.py")
    cfg = str(tmp_path / "cfg.txt")
    write_config(cfg, root, str(tmp_path / "models"), "siren", extra="num_fine_samples = 16")
    open(cfg, "a").write("num_render_poses = 5\n")
    res = subprocess.run([sys.executable, train_py, "--config", cfg, "--max-iters", "2"], env=env, cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    ckpt = str(tmp_path / "models" / "siren_model_000002.pth")
    assert os.path.exists(ckpt), os.listdir(str(tmp_path / "models"))
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / f"frames{world}")
        common = [eval_py, "--config", cfg, "--checkpoint", ckpt, "--output", out]
        if world == 1:
            cmd = [sys.executable] + common
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                   "127.0.0.1", "--master-port", str(29900 + os.getpid() % 90)] + common + ["--one-device"]
        res = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
        outs[world] = {f: np.asarray(Image.open(os.path.join(out, f))) for f in sorted(os.listdir(out))}
    assert sorted(outs[1]) == sorted(outs[2]) == [f"frame_{i:04d}.png" for i in range(5)]
    for f in outs[1]:
        assert np.array_equal(outs[1][f], outs[2][f]), f
    assert len({outs[1][f].tobytes() for f in outs[1]}) > 1           # the poses are not all one image
