"""bench.py's JSON line on the GPU box, at small sizes: the headline is the reference's configuration (numeric Jacobians) with a
same-mode cpu_baseline; for N > 1 (rehearsed with two gloo ranks sharing the one GPU) the line carries the §8(e) reporting collectives."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    return json.loads(run.stdout.strip().splitlines()[-1])


def test_headline_is_the_reference_configuration(gpu):
    j = _run(["--batch", "4096", "--epochs", "16", "--steps", "3", "--warmup", "1", "--legs", "cfg2_analytic", "--cpu-tags", "256", "--cpu-epochs", "32"])
    assert j["config"]["jacobian"] == "numeric" and j["dtype"] == "f64" and j["n_gpus"] == 1
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["kernel"].endswith("numeric>") and 0 < j["roofline"]["frac"] < 1
    cb = j["cpu_baseline"]
    # numeric vs numeric: 1e-5 m per update, up to the rare update whose LM accept / reject decision the 1e-7 noise of the difference quotient flips
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["median_abs_diff_vs_gpu_m"] < 1e-7 and cb["max_abs_diff_vs_gpu_m"] < 1e-4
    assert cb["frac_updates_diff_gt_1e-5_m"] <= 1e-3
    leg = j["legs"]["cfg2_analytic"]
    assert leg["jacobian"] == "analytic" and j["value_fast_mode"] == leg["value"] and leg["max_abs_diff_vs_cpu_baseline_m"] < 1e-3
    assert j["collectives"]["ranks"] == 1


def test_two_rank_gloo_rehearsal_carries_the_reporting_collectives(gpu):
    j = _run(["--gpus", "2", "--dist-backend", "gloo", "--batch", "4096", "--epochs", "8", "--steps", "2", "--warmup", "1", "--legs", "none", "--no-cpu-baseline"])
    assert j["n_gpus"] == 2 and j["ranks_joined"] == 2
    c = j["collectives"]
    assert c["ranks"] == 2 and c["backend"].startswith("gloo") and c["all_gather"]["checksum_ok"] is True
    assert c["all_gather"]["bytes"] == 2 * 4 * 4096 * 8 and c["all_reduce"]["bytes"] == 32 and c["bytes"] == c["all_gather"]["bytes"] + 32
    v = c["all_reduce"]["values"]
    assert v[0] > 0 and v[1] >= 2 * 4096 * 8 * 2 * 1 and v[2] == 0     # sum chi2, sum LM trials over both ranks' timed epochs, no non-finite estimate
