"""Render a spherical camera path from a checkpoint (counterpart of the reference's eval.py).

Poses as reference eval.py:14-41,91-97 (theta sweep, phi = -30 deg, radius 4), rays generated
on the device, frames quantised as (clip(rgb,0,1)*255) truncated to uint8 (eval.py:149-150) and
written as PNG.  With torchrun the poses are sharded over ranks (no collective).  A checkpoint of a
coarse + fine pair (train.py with `num_fine_samples`) is rendered hierarchically; `--precision bf16`
or `bf16x3` selects the opt-in inference kernels on the bf16 matrix pipe (NeRF only; bf16x3 keeps fp32 accuracy)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from nerf.data import compute_rays_device, load_dataset  # noqa: E402
from nerf.models import NeRF, Siren  # noqa: E402
from nerf.rendering import render_nerf, render_nerf_hierarchical  # noqa: E402
from nerf.utils import parse_config  # noqa: E402


def pose_spherical(theta: float, phi: float, radius: float) -> np.ndarray:
    """translate z by radius, rotate x by phi, rotate y by theta (degrees), swap axes."""
    t = np.eye(4, dtype=np.float32)
    t[2, 3] = radius
    p = phi / 180.0 * np.pi
    rx = np.eye(4, dtype=np.float32)
    rx[1, 1] = rx[2, 2] = np.cos(p)
    rx[1, 2], rx[2, 1] = -np.sin(p), np.sin(p)
    th = theta / 180.0 * np.pi
    ry = np.eye(4, dtype=np.float32)
    ry[0, 0] = ry[2, 2] = np.cos(th)
    ry[0, 2], ry[2, 0] = -np.sin(th), np.sin(th)
    swap = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    return (swap @ (ry @ (rx @ t))).astype(np.float32)


def write_png(path: str, frame: np.ndarray) -> None:
    try:
        import imageio
        imageio.imwrite(path, frame)
    except ImportError:
        from PIL import Image
        Image.fromarray(frame).save(path)


def poses_for_rank(n_poses: int, rank: int, world: int) -> range:
    """Pose indices rendered by `rank` of `world` processes (one per GPU): round-robin, every pose exactly once,
    no exchange between ranks -- each writes its own frames."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    return range(rank, n_poses, world)


def quantise(rgb: torch.Tensor) -> torch.Tensor:
    """(clip(rgb, 0, 1) * 255) truncated to uint8, on the device (eval.py:149-150)."""
    return (rgb.clamp(0.0, 1.0) * 255).to(torch.uint8)


def main(argv=None):
    np.random.seed(42)
    torch.manual_seed(42)
    ap = argparse.ArgumentParser(description="Render a camera path with a trained model.")
    ap.add_argument('--config', type=str, required=True)
    ap.add_argument('--checkpoint', type=str, required=True)
    ap.add_argument('--output', type=str, default='rendered_frames')
    ap.add_argument('--resolution', type=int, default=None, help='override H=W (default: the test image size)')
    ap.add_argument('--precision', choices=['fp32', 'bf16', 'bf16x3'], default='fp32')
    ap.add_argument('--one-device', action='store_true', help='rehearsal only: every rank renders on cuda:0')
    args = ap.parse_args(argv)
    cfg = parse_config(args.config)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0")))
    if not torch.cuda.is_available():
        raise RuntimeError("this build needs a HIP device (no CPU path)")
    device = torch.device("cuda", torch.cuda.current_device())

    ckpt = torch.load(args.checkpoint, map_location='cpu', weights_only=True)
    model_type = ckpt.get('model_type', cfg.get('model_type', 'NeRF')).lower()
    near, far = float(cfg.get('near', 2.0)), float(cfg.get('far', 6.0))
    num_samples = int(cfg.get('num_samples', 256))
    chunk_size = int(cfg.get('chunk_size', 8192))
    n_poses = int(cfg.get('num_render_poses', 40))
    os.makedirs(args.output, exist_ok=True)
    if model_type not in ('nerf', 'siren'):
        raise ValueError(f"Invalid model type: {model_type}")
    make = NeRF if model_type == 'nerf' else Siren
    state = ckpt['model_state_dict']
    pair = any(k.startswith('coarse.') for k in state)
    if pair:
        model = torch.nn.ModuleDict({"coarse": make(), "fine": make()}).to(device)
        nets = [model["coarse"], model["fine"]]
        num_fine = int(cfg.get('num_fine_samples', 128))
    else:
        model = make().to(device)
        nets = [model]
    model.load_state_dict(state)
    model.eval()
    for n in nets:
        n.set_inference_precision(args.precision)

    def render(o, d):
        if pair:
            return render_nerf_hierarchical(nets[0], nets[1], o, d, near, far, num_samples, num_fine, device=device,
                                            white_background=True, chunk_size=chunk_size, coarse_rgb=False)[1]
        return render_nerf(nets[0], o, d, near, far, num_samples=num_samples, device=device, white_background=True,
                           chunk_size=chunk_size)

    images, _, focal = load_dataset(cfg.get('dataset_path', './datasets/lego'), mode='test', single_image=True)
    H, W = images.shape[1:3]
    if args.resolution:
        focal = focal * args.resolution / W
        H = W = args.resolution
    angles = np.linspace(-180, 180, n_poses + 1)[:-1]
    for i in poses_for_rank(n_poses, rank, world):
        # The stratified jitter stays on at render time (rendering.py:26 has no switch).  Its generator is re-seeded
        # per pose, so that a frame does not depend on which rank rendered it or on the frames before it: the frames
        # of a pose-sharded run are bit for bit those of a single process.
        torch.manual_seed(42 + i)
        c2w = torch.from_numpy(pose_spherical(float(angles[i]), -30.0, 4.0))[None].to(device)
        ro, rd = compute_rays_device(H, W, c2w, focal)
        with torch.no_grad():
            rgb = render(ro[0], rd[0])
        frame = quantise(rgb.reshape(H, W, 3)).cpu().numpy()
        write_png(os.path.join(args.output, f"frame_{i:04d}.png"), frame)
    if rank == 0:
        print(f"rendered {n_poses} frames of {H}x{W} to {args.output}")


if __name__ == '__main__':
    main()
