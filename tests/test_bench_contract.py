"""The bench line the driver parses: the committed line of the final tree (profiles/r03_bench_cfg2_default.json, printed by ``python bench.py``
on one MI355X) carries every key of the contract, with consistent values.  CPU only: it reads the committed evidence, not the GPU."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_default_bench_line_has_the_contract_keys():
    d = _line("r03_bench_cfg2_default.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["metric"].split(" at ")[0] in base["metric"] and d["unit"] == "steps/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"      # no published number to compare with
    assert "workload" in d["config"] and d["config"]["workload"].startswith("cfg2") and "model" not in d["config"]
    assert d["value"] == pytest.approx(d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3), rel=1e-9)
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0.5 < r["frac"] < 1.0
    # achieved = algorithmic flops of a launch / its average duration: 2 * pairs * 128 * 128
    assert r["achieved"] == pytest.approx(2 * r["avg_pairs_per_launch"] * 128 * 128 / (r["avg_launch_us"] * 1e-6) / 1e12, rel=1e-6)
    assert r["traffic"] > 2 * 4 * 128 * 80000                        # counter bytes per launch: at least the rows read and written once
    # the kernel's average inside the replayed graph (committed kernel trace) agrees with the eager timing
    assert r["in_graph_trace"]["frac"] == pytest.approx(r["frac"], abs=0.03)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == "steps/s" and c["cores"] >= 1 and 0 < c["value"] < d["value"]
    # the roofline cross-checks of the review: two launches of the kernel fit in a step, the whole-step rate is below the kernel's
    assert 2 * r["avg_launch_us"] < d["ms_per_step"] * 1e3
    assert d["step_work"]["tflops_whole_step"] < r["achieved"]


@pytest.mark.parametrize("name,lo,hi", [("r03_bench_cfg3.json", 1.5, 2.5), ("r03_bench_cfg4.json", 60, 130), ("r03_bench_cfg5.json", 300, 700),
                                        ("r03_bench_shard_0of8.json", 0.3, 0.6), ("r03_bench_shard_2of8.json", 0.25, 0.5)])
def test_other_committed_lines_name_their_workload(name, lo, hi):
    d = _line(name)
    assert lo < d["ms_per_step"] < hi and d["unit"] == "steps/s" and "workload" in d["config"]
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-6)
