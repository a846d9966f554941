"""End-to-end rate of the training driver (nerf-pytorch_amd/train.py) at the benchmark shape: 100 views of
800x800 synthetic pixels resident in HBM (64 M rays), 4096 rays per step, 64 coarse + 128 fine samples.
The PNG loader is replaced by a synthetic array; everything else is the driver's own loop.  GPU box."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import numpy as np
import torch
import train as driver
import eval as ev

VIEWS, RES, ITERS = 100, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 160


def synthetic(dataset_path, mode='train', single_image=False):
    n = 1 if single_image else (VIEWS if mode == 'train' else 4)
    rng = np.random.default_rng(0 if mode == 'train' else 1)
    images = rng.random((n, RES, RES, 3), dtype=np.float32)
    poses = np.stack([ev.pose_spherical(360.0 * i / n, -30.0, 4.0) for i in range(n)]).astype(np.float32)
    focal = 0.5 * RES / np.tan(0.5 * 0.6911112)
    return images, poses, focal


driver.load_dataset = synthetic
with tempfile.TemporaryDirectory() as tmp:
    cfg = os.path.join(tmp, "cfg.txt")
    with open(cfg, "w") as f:
        f.write(f"dataset_path = {tmp}\nsave_path = {tmp}/models\nmodel_type = nerf\nnum_random_rays = 4096\n"
                f"num_samples = 64\nnum_fine_samples = 128\nchunk_size = 8192\nnum_iters = {ITERS}\nlog_interval = 40\n"
                f"val_interval = 100000\nsave_interval = 100000\nnear = 2.0\nfar = 6.0\n")
    # time the steady part: stamp when iteration 40 and the last logged iteration print
    stamps = {}
    orig = driver.log_training_metrics

    def stamped(step, *a, **k):
        torch.cuda.synchronize()
        stamps[step] = time.perf_counter()
        return orig(step, *a, **k)

    driver.log_training_metrics = stamped
    driver.main(["--config", cfg])
steps = sorted(stamps)
a, b = steps[1], steps[-1]
dt = (stamps[b] - stamps[a]) / (b - a)
print(f"train.py steady state: {dt * 1e3:.2f} ms per iteration over iterations {a}..{b} = {4096 / dt:,.0f} rays/s")
