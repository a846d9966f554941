"""Run-time plumbing around the render path: `key = value` config files, timestamps,
checkpoint files and the per-iteration log line.  Same function names, file format and
checkpoint keys as the reference's nerf/utils.py (:9-34 parse_config, :37-47 format_elapsed_time,
:50-63 save_checkpoint, :66-77 log_training_metrics) and the same call signatures, so the reference's
train.py / eval.py call them unchanged.  Config files interchange; of a checkpoint only
`model_state_dict` (and `step`, `model_type`) interchanges in general: the optimizer / scheduler entries
are those of whatever optimizer wrote the file (torch.optim.Adam + LambdaLR, or this package's FusedAdam,
whose state is not a torch.optim state_dict)."""
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


def _emit(line: str) -> None:
    """tqdm.write when tqdm is importable (keeps a progress bar intact, as the reference does), else print."""
    try:
        from tqdm import tqdm
    except ImportError:
        print(line, flush=True)
    else:
        tqdm.write(line)


def log_training_metrics(step, scheduler, loss, start_time, writer: Optional[object] = None) -> str:
    """Write the per-iteration report line and mirror loss / psnr / learning_rate to a TensorBoard-like
    writer (reference utils.py:66-77, called as train.py:187 does: `(step, scheduler, loss, start_time,
    writer)`).  `scheduler` is anything with get_last_lr() (torch LR schedulers) or the current rate as
    a float; `loss` a one-element tensor (synchronises through .item(), like the reference) or a float;
    `writer` may be None.  The line is also returned."""
    current_lr = float(scheduler.get_last_lr()[0]) if hasattr(scheduler, "get_last_lr") else float(scheduler)
    mse = float(loss.item()) if hasattr(loss, "item") else float(loss)
    psnr = mse_to_psnr(mse)
    line = "[%s] [Iter %07d] LR: %.6f MSE: %.4f PSNR: %.2f" % (format_elapsed_time(start_time), step, current_lr, mse, psnr)
    _emit(line)
    if writer is not None:
        for tag, value in (("loss", mse), ("psnr", psnr), ("learning_rate", current_lr)):
            writer.add_scalar(tag, value, step)
    return line
