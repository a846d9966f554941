"""bench.py's roofline.traffic lookup against the committed rocprofv3 counter summaries (CPU only).

The lookup goes by function name + leading template arguments (profiles/kernel_names.py), so a kernel that gains
a trailing template parameter keeps matching; a counter file without a row for the wanted kernel is reported on
stderr instead of silently emptying the figure."""
import csv
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))

import bench  # noqa: E402
from kernel_names import kernel_key, matches  # noqa: E402


def test_kernel_key_forms():
    full = "void (anonymous namespace)::nerf_fwd_kernel<true, false>((anonymous namespace)::FwdArgs)"
    assert kernel_key(full) == "nerf_fwd_kernel<true,false>"
    assert kernel_key("(anonymous namespace)::nerf_dgrad_kernel((anonymous namespace)::BwdArgs)") == "nerf_dgrad_kernel"
    assert kernel_key("stratified_t_kernel(float, float, float, int, float const*, float*, float*)") == "stratified_t_kernel"
    assert matches(full, "nerf_fwd_kernel", ("true",))
    assert matches(full, "nerf_fwd_kernel", ("true", "false"))
    assert not matches(full, "nerf_fwd_kernel", ("false",))
    assert not matches(full, "siren_fwd_kernel", ("true",))
    # a third template parameter added later still matches the two-argument request
    assert matches(full.replace("<true, false>", "<true, false, 3>"), "nerf_fwd_kernel", ("true",))
    assert matches("nerf_fwd_kernel<true,false>", "nerf_fwd_kernel", ("true",))      # key form as input


@pytest.mark.parametrize("entry", sorted(bench.DESIGN_BYTES))
def test_committed_passes_give_traffic_close_to_design(entry):
    """Every training kernel bench.py can name as `roofline.kernel` finds its rows in the newest committed passes of
    its model (if that model has been profiled), and the HBM bytes per sample are the by-design ones: within 10 % for
    the forward kernels (workspace writes dominate), 25 % for the data-gradient kernels (weight re-fetches by
    eight L2s are on top of the by-design reads)."""
    model, kernel = entry
    per_sample, files = bench.pmc_traffic_per_sample(model, kernel)
    infix = "_x3" if kernel.endswith("_x3") else ("" if model == "nerf" else f"_{model}")
    have = [f for f in os.listdir(bench.PROFILES) if f.endswith(f"_pmc{infix}_fetch.csv")]
    if not have:
        pytest.skip(f"no committed counter pass for {model}{infix}")
    assert per_sample is not None, f"{entry}: counter files {have} exist but no row matched"
    assert all(os.path.exists(os.path.join(ROOT, f)) for f in files)
    # (the bf16x3 kernels stream a weight image three times the fp32 one's size per 64-sample tile: its re-fetches past the
    # eight L2s, 3.1 GB per 1,048,576-sample forward, come on top of the by-design workspace bytes)
    tol = 0.10 if kernel == "mlp_fwd_train" else 0.25
    want = bench.DESIGN_BYTES[entry]
    assert abs(per_sample - want) <= tol * want, (entry, per_sample, want)


def _write_pass(tmp, tag, kernel_name, fetch_kib, write_kib, with_key):
    for which, counter, val in (("fetch", "FETCH_SIZE", fetch_kib), ("write", "WRITE_SIZE", write_kib)):
        with open(os.path.join(tmp, f"{tag}_pmc_{which}.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel"] + (["kernel_key"] if with_key else []) + ["counter", "dispatches", "mean_per_dispatch"])
            w.writerow([kernel_name] + ([kernel_key(kernel_name)] if with_key else []) + [counter, 4, val])
    with open(os.path.join(tmp, f"{tag}_pmc_meta.json"), "w") as f:
        json.dump({"samples_per_dispatch": 1024}, f)


@pytest.mark.parametrize("with_key", [False, True])
def test_lookup_survives_a_new_template_parameter(tmp_path, with_key):
    name = "void (anonymous namespace)::nerf_fwd_kernel<true, false, 7>((anonymous namespace)::FwdArgs)"
    _write_pass(str(tmp_path), "r9", name, 100.0, 1000.0, with_key)
    per_sample, files = bench.pmc_traffic_per_sample("nerf", "mlp_fwd_train", profiles=str(tmp_path))
    assert per_sample == pytest.approx((2 * 100.0 + 1000.0) * 1024 / 1024)
    assert any("r9_pmc_fetch.csv" in f for f in files)


def test_stale_name_is_reported_not_swallowed(tmp_path, capsys):
    _write_pass(str(tmp_path), "r9", "void (anonymous namespace)::renamed_fwd_kernel<true>(Args)", 1.0, 1.0, True)
    assert bench.pmc_traffic_per_sample("nerf", "mlp_fwd_train", profiles=str(tmp_path)) == (None, None)
    assert "no FETCH_SIZE/WRITE_SIZE row" in capsys.readouterr().err


def test_newest_round_with_files_wins(tmp_path):
    name = "void (anonymous namespace)::nerf_fwd_kernel<true, false>((anonymous namespace)::FwdArgs)"
    _write_pass(str(tmp_path), "r2", name, 10.0, 10.0, False)
    _write_pass(str(tmp_path), "r10", name, 20.0, 20.0, True)
    per_sample, files = bench.pmc_traffic_per_sample("nerf", "mlp_fwd_train", profiles=str(tmp_path))
    assert per_sample == pytest.approx(60.0) and any("r10_" in f for f in files)
    # a model without passes of its own in the newest round falls back to an older round that has them
    os.rename(os.path.join(str(tmp_path), "r2_pmc_fetch.csv"), os.path.join(str(tmp_path), "r2_pmc_siren_fetch.csv"))
    os.rename(os.path.join(str(tmp_path), "r2_pmc_write.csv"), os.path.join(str(tmp_path), "r2_pmc_siren_write.csv"))
    assert bench.pmc_traffic_per_sample("siren", "mlp_fwd_train", profiles=str(tmp_path)) == (None, None)  # wrong kernel in it
