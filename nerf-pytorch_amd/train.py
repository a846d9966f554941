"""Training driver on the HIP path (counterpart of the reference's train.py).

Same config keys, seeding, Adam + LambdaLR schedule, logging / checkpoint / validation cadence
(reference train.py:20-259).  Differences: rays live on the device (nerf.data.DeviceRays), the
optimizer is one fused Adam launch over a flat parameter buffer (`--optimizer torch` selects
torch.optim.Adam + LambdaLR exactly as the reference), TensorBoard is optional, and
`torchrun --nproc-per-node N train.py ...` shards every batch's rays over N GPUs.
Extension (no counterpart in the reference): the config key `num_fine_samples = N` (N > 0) trains a
coarse and a fine network with inverse-CDF re-sampling (`num_samples` coarse + N fine samples per
ray, both losses); checkpoints then hold both networks under `coarse.` / `fine.` key prefixes.
Extension: the config key `train_precision = bf16x3` (default fp32) runs the training kernels on the bf16
matrix pipe at fp32 accuracy (three bf16 terms per operand; nerf/models.py set_training_precision).
"""
import argparse
import datetime
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from nerf import parallel  # noqa: E402
from nerf.data import DeviceRays, compute_rays_device, load_dataset  # noqa: E402
from nerf.loss import mse_to_psnr  # noqa: E402
from nerf.models import NeRF, Siren  # noqa: E402
from nerf.rendering import render_nerf, render_nerf_hierarchical  # noqa: E402
from nerf.utils import format_elapsed_time, log_training_metrics, parse_config, save_checkpoint  # noqa: E402


def build_model(model_type: str):
    if model_type == 'nerf':
        return NeRF()
    if model_type == 'siren':
        return Siren()
    raise ValueError(f"Invalid model type: {model_type}")


def main(argv=None):
    seed = 42
    np.random.seed(seed)
    torch.manual_seed(seed)
    ap = argparse.ArgumentParser(description="Train NeRF on a given dataset using volumetric rendering.")
    ap.add_argument('--config', type=str, required=True)
    ap.add_argument('--resume', type=str, default=None)
    ap.add_argument('--optimizer', choices=['fused', 'torch'], default='fused')
    ap.add_argument('--max-iters', type=int, default=None, help='stop early (smoke runs)')
    ap.add_argument('--dist-backend', choices=['nccl', 'gloo'], default='nccl',
                    help='nccl = RCCL, one GPU per rank; gloo + --one-device rehearses the multi-rank path on one GPU')
    ap.add_argument('--one-device', action='store_true', help='rehearsal only: every rank uses cuda:0')
    args = ap.parse_args(argv)
    cfg = parse_config(args.config)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        torch.cuda.set_device(0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(args.dist_backend)
    if not torch.cuda.is_available():
        raise RuntimeError("this build needs a HIP device (no CPU path)")
    device = torch.device("cuda", torch.cuda.current_device())
    torch.cuda.manual_seed_all(seed)

    dataset_path = cfg.get('dataset_path', './datasets/lego')
    num_random_rays = int(cfg.get('num_random_rays', 1024))
    chunk_size = int(cfg.get('chunk_size', 8192))
    num_samples = int(cfg.get('num_samples', 256))
    num_iters = int(cfg.get('num_iters', 150000))
    learning_rate = float(cfg.get('learning_rate', 5e-4))
    near, far = float(cfg.get('near', 2.0)), float(cfg.get('far', 6.0))
    save_path = cfg.get('save_path', './models')
    save_interval = int(cfg.get('save_interval', 5000))
    lr_decay = float(cfg.get('lr_decay', 150))
    lr_decay_factor = float(cfg.get('lr_decay_factor', 0.1))
    lr_min = float(cfg.get('lr_min', 1e-5))
    first_step_render = cfg.get('first_step_render', 'False').lower() == 'true'
    log_interval = int(cfg.get('log_interval', 10))
    val_interval = int(cfg.get('val_interval', 1000))
    num_fine = int(cfg.get('num_fine_samples', 0))
    if num_fine > 0 and args.optimizer != 'fused':
        raise ValueError("num_fine_samples > 0 (coarse + fine networks) needs --optimizer fused")
    if args.max_iters is not None:
        num_iters = min(num_iters, args.max_iters)
    os.makedirs(save_path, exist_ok=True)

    ckpt = None
    if args.resume is not None:
        ckpt = torch.load(args.resume, map_location='cpu', weights_only=True)
        model_type = ckpt.get('model_type', cfg.get('model_type', 'NeRF')).lower()
    else:
        model_type = cfg.get('model_type', 'NeRF').lower()
    if rank == 0:
        print(f"===== Training: {model_type}, {num_random_rays} rays x {num_samples} samples, {num_iters} iters, "
              f"lr {learning_rate}, world {world}, device {torch.cuda.get_device_name(device)} =====")

    if num_fine > 0:
        model = torch.nn.ModuleDict({"coarse": build_model(model_type), "fine": build_model(model_type)}).to(device)
        nets = [model["coarse"], model["fine"]]
    else:
        model = build_model(model_type).to(device)
        nets = [model]

    train_precision = cfg.get('train_precision', 'fp32').lower()
    if train_precision != 'fp32':
        for net in nets:
            net.set_training_precision(train_precision)   # raises for anything but 'fp32' / 'bf16x3'

    def render(o, d, us=None, ufs=None):
        """rgb predictions of every network that is trained ([single] or [coarse, fine])."""
        if num_fine > 0:
            return list(render_nerf_hierarchical(nets[0], nets[1], o, d, near, far, num_samples, num_fine, device=device,
                                                 white_background=True, chunk_size=chunk_size, u=us, u_fine=ufs))
        return [render_nerf(nets[0], o, d, near, far, num_samples=num_samples, device=device, white_background=True,
                            chunk_size=chunk_size, u=us)]

    images, c2w, focal = load_dataset(dataset_path, mode='train')
    images_val, c2w_val, focal_val = load_dataset(dataset_path, mode='val')
    N_val = images.shape[0]  # the reference indexes validation images with the TRAIN count (train.py:115)
    # data-parallel ranks draw the SAME batches (identically seeded generator) and each renders an equal slice:
    # the short last batch of an epoch is dropped there, so no rank ever gets an empty or unequal slice
    pool = DeviceRays(images, c2w, focal, device, generator=torch.Generator(device=device).manual_seed(seed),
                      drop_last=world > 1)
    if world > 1 and num_random_rays > len(pool):
        raise ValueError(f"num_random_rays = {num_random_rays} exceeds the {len(pool)} rays of the dataset")

    if ckpt is not None:
        model.load_state_dict(ckpt['model_state_dict'])
    sched = None
    if args.optimizer == 'torch':
        opt = torch.optim.Adam(model.parameters(), lr=learning_rate)
        gamma = lr_decay_factor ** (1 / (lr_decay * 1000))
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: max(gamma ** s, lr_min / learning_rate))
        flat = None
        parallel.broadcast_module_parameters(model, 0)
    else:
        flat = parallel.FlatParameters(model)   # one bucket for the network, or for the coarse + fine pair
        flat.broadcast_parameters(0)
        opt = parallel.FusedAdam(flat, lr=learning_rate)
    # the stratified draw is shared by all rays of a batch (rendering.py:26), hence by all ranks: every rank
    # draws it from its own generator with the same seed; one broadcast here verifies that they agree
    ugen = torch.Generator(device=device).manual_seed(seed + 1)
    parallel.check_shared_draws(device, ugen)
    start_iter = 0
    if ckpt is not None:
        try:
            opt.load_state_dict(ckpt['optimizer_state_dict'])
            if sched is not None and ckpt['scheduler_state_dict']:
                sched.load_state_dict(ckpt['scheduler_state_dict'])
        except (KeyError, ValueError, TypeError) as e:
            print(f"optimizer state not restored ({e}); continuing with fresh moments")
        start_iter = ckpt['step']

    writer = None
    if rank == 0:
        try:
            from torch.utils.tensorboard import SummaryWriter
            stamp = datetime.datetime.now().strftime('%Y-%m-%d_%H-%M-%S')
            writer = SummaryWriter(log_dir=f"./logs/{model_type}_{os.path.basename(dataset_path)}_{stamp}")
            writer.add_text('config', str(cfg))
        except Exception:
            writer = None

    start_time = datetime.datetime.now()
    mse = torch.nn.MSELoss()
    per_rank = parallel.shard_slice(num_random_rays, rank, world) if world > 1 else slice(0, num_random_rays)
    step = start_iter
    try:
        for step in range(start_iter, num_iters):
            o, d, tgt = pool.next_batch(num_random_rays)
            o, d, tgt = o[per_rank], d[per_rank], tgt[per_rank]
            n_chunks = (o.shape[0] + chunk_size - 1) // chunk_size
            us = [parallel.shared_uniform(num_samples, device, ugen) for _ in range(n_chunks)]
            preds = render(o, d, us)
            opt.zero_grad()
            loss = sum(mse(p, tgt) for p in preds)
            loss.backward()
            loss = mse(preds[-1].detach(), tgt)   # what is logged: the loss of the image that is kept
            if world > 1 and args.dist_backend == 'gloo':
                torch.cuda.synchronize()    # rehearsal backend only: start the host-staged collective on an idle device
            if flat is not None:
                flat.all_reduce_mean(scale_in_optimizer=True)   # 1/world rides in the Adam kernel's gradient scale
                opt.lr = learning_rate * parallel.lr_lambda(step, learning_rate, lr_decay, lr_decay_factor, lr_min)
            else:
                parallel.all_reduce_module_grads(model)         # torch.optim path: one flattened all-reduce
            opt.step()
            if sched is not None:
                sched.step()
            cur_lr = sched.get_last_lr()[0] if sched is not None else \
                learning_rate * parallel.lr_lambda(step + 1, learning_rate, lr_decay, lr_decay_factor, lr_min)

            if rank == 0 and step % log_interval == 0:
                log_training_metrics(step, sched if sched is not None else cur_lr, loss, start_time, writer)
            if rank == 0 and step % save_interval == 0 and 0 < step < num_iters - 1:
                name = save_checkpoint(step, model, opt, sched, save_path, model_type)
                print(f"[{format_elapsed_time(start_time)}] Model saved to {name} at iteration {step}")
            if rank == 0 and step % val_interval == 0 and (step > 0 or first_step_render):
                i = np.random.randint(min(N_val, images_val.shape[0]))
                H, W = images_val.shape[1:3]
                ro, rd = compute_rays_device(H, W, torch.from_numpy(c2w_val[i:i + 1]).to(device), focal_val)
                model.eval()
                with torch.no_grad():
                    img = render(ro[0], rd[0])[-1]
                model.train()
                img = img.reshape(H, W, 3).cpu().numpy()
                val_mse = float(np.mean((img - images_val[i]) ** 2))
                print(f"[Validation Step] Iter {step}  PSNR: {mse_to_psnr(val_mse):.2f}", flush=True)
                if writer is not None:
                    writer.add_scalar("val/psnr", mse_to_psnr(val_mse), step)
                    writer.add_image("val/render", torch.from_numpy(np.clip(img, 0, 1)).permute(2, 0, 1), step)
        if rank == 0:
            name = save_checkpoint(num_iters, model, opt, sched, save_path, model_type)
            print(f"[{format_elapsed_time(start_time)}] Training complete! Final model saved to {name}")
    except KeyboardInterrupt:
        if rank == 0:
            name = save_checkpoint(step, model, opt, sched, save_path, model_type)
            print(f"Keyboard interrupt: checkpoint saved to {name}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
