"""The reference's own scripts must be able to call this package unchanged: every call that
train.py / eval.py of josedelrey/nerf-pytorch make into `nerf.*` (extracted from their AST by
tests/golden/make_callsites.py into tests/golden/callsites.json) is bound against the signature
of the function of the same name here.  CPU only: nothing is launched."""
import datetime
import importlib
import inspect
import json
import os

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "callsites.json")) as _f:
    FIXTURE = json.load(_f)


def _id(call):
    return f"{call['file']}:{call['line']}:{call['name']}"


def test_fixture_is_complete():
    names = {c["name"] for c in FIXTURE["calls"]}
    assert len(FIXTURE["calls"]) >= 24
    assert {"render_nerf", "NeRF", "Siren", "log_training_metrics", "save_checkpoint", "compute_rays", "RayDataset",
            "load_dataset", "parse_config", "format_elapsed_time", "mse_to_psnr"} <= names


def test_every_imported_name_exists():
    for script, dotted in FIXTURE["imports"].items():
        for full in dotted:
            module, name = full.rsplit(".", 1)
            assert hasattr(importlib.import_module(module), name), f"{script} imports {full}"


@pytest.mark.parametrize("call", FIXTURE["calls"], ids=_id)
def test_call_site_binds(call):
    fn = getattr(importlib.import_module(call["module"]), call["name"])
    target = fn.__init__ if inspect.isclass(fn) else fn
    args = [object()] * (call["positional"] + (1 if inspect.isclass(fn) else 0))
    inspect.signature(target).bind(*args, **{k: object() for k in call["keywords"]})


def test_render_nerf_defaults_are_the_references():
    from nerf.rendering import render_nerf
    p = inspect.signature(render_nerf).parameters
    assert list(p)[:9] == ["model", "rays_o", "rays_d", "near", "far", "num_samples", "device", "white_background",
                           "chunk_size"]
    assert (p["num_samples"].default, p["device"].default, p["white_background"].default, p["chunk_size"].default) == \
        (256, "cpu", True, 8192)
    extras = [n for n in list(p)[9:]]
    assert all(p[n].kind is inspect.Parameter.KEYWORD_ONLY for n in extras)


class _Writer:
    def __init__(self):
        self.rows = []

    def add_scalar(self, tag, value, step):
        self.rows.append((tag, float(value), step))


def test_log_training_metrics_as_train_py_calls_it(capsys):
    """train.py:187: log_training_metrics(step, scheduler, loss, start_time, writer) with a torch LR
    scheduler and the loss tensor; the line format is utils.py:72-73."""
    from nerf.utils import log_training_metrics
    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([w], lr=5e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: 0.5)
    writer = _Writer()
    loss = torch.tensor(0.01)
    line = log_training_metrics(30, sched, loss, datetime.datetime.now(), writer)
    out = capsys.readouterr()
    assert "[Iter 0000030] LR: 0.000250 MSE: 0.0100 PSNR: 20.00" in line
    assert line in out.out + out.err
    assert [r[0] for r in writer.rows] == ["loss", "psnr", "learning_rate"]
    assert writer.rows[2][1] == pytest.approx(2.5e-4) and writer.rows[1][1] == pytest.approx(20.0)
    # this package's own driver passes the rate as a float and may have no writer
    assert "LR: 0.001000" in log_training_metrics(1, 1e-3, 0.5, datetime.datetime.now(), None)


def test_save_checkpoint_as_train_py_calls_it(tmp_path):
    """train.py:191 / eval.py:108-109: positional (step, model, optimizer, scheduler, save_path, model_type),
    file `{model_type}_model_{step:06d}.pth`, loadable with weights_only=True."""
    from nerf.utils import save_checkpoint
    model = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: 1.0)
    name = save_checkpoint(7, model, opt, sched, str(tmp_path), "nerf")
    assert os.path.basename(name) == "nerf_model_000007.pth"
    ck = torch.load(name, map_location="cpu", weights_only=True)
    assert set(ck) == {"step", "model_type", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict"}
    assert ck["step"] == 7 and ck["model_type"] == "nerf"
    assert torch.equal(ck["model_state_dict"]["weight"], model.weight.detach())
