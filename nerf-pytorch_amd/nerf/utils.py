"""Config parsing, checkpoints and logging helpers (counterpart of the reference's nerf/utils.py)."""
from __future__ import annotations

import datetime
import os

import torch

from .loss import mse_to_psnr


def parse_config(config_path: str) -> dict:
    """Flat `key = value  # comment` text -> dict of strings (reference utils.py:9-34)."""
    out = {}
    with open(config_path, 'r') as f:
        for raw in f:
            line = raw.split('#', 1)[0].strip()
            if not line:
                continue
            if '=' not in line:
                print(f"Warning: Invalid line in config file: {line}")
                continue
            k, v = line.split('=', 1)
            out[k.strip()] = v.strip()
    return out


def format_elapsed_time(start_time: datetime.datetime) -> str:
    """HH:MM:SS since start_time (reference utils.py:37-47)."""
    s = int((datetime.datetime.now() - start_time).total_seconds())
    return f"{s // 3600:02d}:{s % 3600 // 60:02d}:{s % 60:02d}"


def save_checkpoint(step, model, optimizer, scheduler, save_path, model_type) -> str:
    """{step, model_type, model/optimizer/scheduler state_dict} -> {model_type}_model_{step:06d}.pth
    (reference utils.py:50-63).  `scheduler` may be None for optimizers that carry their own lr."""
    ckpt = {'step': step, 'model_type': model_type,
            'model_state_dict': {k: v.detach().clone() for k, v in model.state_dict().items()},
            'optimizer_state_dict': optimizer.state_dict(),
            'scheduler_state_dict': scheduler.state_dict() if scheduler is not None else {}}
    name = os.path.join(save_path, f"{model_type}_model_{step:06d}.pth")
    torch.save(ckpt, name)
    return name


def log_training_metrics(step, current_lr, loss, start_time, writer=None) -> str:
    """One log line (and TensorBoard scalars when a writer exists) as reference utils.py:66-77."""
    l = float(loss)
    msg = (f"[{format_elapsed_time(start_time)}] [Iter {step:07d}] LR: {current_lr:.6f} "
           f"MSE: {l:.4f} PSNR: {mse_to_psnr(l):.2f}")
    if writer is not None:
        writer.add_scalar('loss', l, step)
        writer.add_scalar('psnr', mse_to_psnr(l), step)
        writer.add_scalar('learning_rate', current_lr, step)
    return msg
