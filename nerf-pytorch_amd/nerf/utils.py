"""Run-time plumbing around the render path: `key = value` config files, timestamps,
checkpoint files and the per-iteration log line.  Same function names, file format and
checkpoint keys as the reference's nerf/utils.py (:9-34 parse_config, :37-47 format_elapsed_time,
:50-63 save_checkpoint, :66-77 log_training_metrics), so its configs and .pth files interchange."""
from __future__ import annotations

import datetime as _dt
import os
import re
from typing import Dict, Optional

import torch

from .loss import mse_to_psnr

_ENTRY = re.compile(r"^\s*([^=#\s][^=#]*?)\s*=\s*(.*?)\s*$")


def parse_config(config_path: str) -> Dict[str, str]:
    """Read a flat text config into {key: value-string}; '#' starts a comment, blank lines are skipped,
    a non-blank line without '=' is reported and ignored."""
    entries: Dict[str, str] = {}
    with open(config_path, "r") as handle:
        text = handle.read()
    for raw in text.splitlines():
        body = raw.partition("#")[0]
        if not body.strip():
            continue
        m = _ENTRY.match(body)
        if m is None:
            print(f"Warning: Invalid line in config file: {body.strip()}")
        else:
            entries[m.group(1)] = m.group(2)
    return entries


def format_elapsed_time(start_time: _dt.datetime) -> str:
    """Wall time since `start_time` as zero-padded HH:MM:SS (hours are not wrapped at 24)."""
    minutes, seconds = divmod(int((_dt.datetime.now() - start_time).total_seconds()), 60)
    hours, minutes = divmod(minutes, 60)
    return "%02d:%02d:%02d" % (hours, minutes, seconds)


def checkpoint_name(save_path: str, model_type: str, step: int) -> str:
    return os.path.join(save_path, "%s_model_%06d.pth" % (model_type, step))


def save_checkpoint(step, model, optimizer, scheduler, save_path, model_type) -> str:
    """Write step, model_type and the three state_dicts under the reference's keys and file name.
    Tensors are cloned first (parameters may be views of a flat buffer that the optimizer keeps
    updating); `scheduler=None` is allowed for optimizers that schedule their own rate."""
    weights = {name: t.detach().clone() for name, t in model.state_dict().items()}
    payload = dict(step=step, model_type=model_type, model_state_dict=weights,
                   optimizer_state_dict=optimizer.state_dict(),
                   scheduler_state_dict={} if scheduler is None else scheduler.state_dict())
    target = checkpoint_name(save_path, model_type, step)
    torch.save(payload, target)
    return target


def log_training_metrics(step, current_lr, loss, start_time, writer: Optional[object] = None) -> str:
    """The per-iteration report line; also mirrored to a TensorBoard-like writer if one is given."""
    mse = float(loss)
    psnr = mse_to_psnr(mse)
    if writer is not None:
        for tag, value in (("loss", mse), ("psnr", psnr), ("learning_rate", current_lr)):
            writer.add_scalar(tag, value, step)
    return "[%s] [Iter %07d] LR: %.6f MSE: %.4f PSNR: %.2f" % (format_elapsed_time(start_time), step, current_lr, mse, psnr)
