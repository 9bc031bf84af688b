"""bench.py's final line stays parseable by the driver (a 21-KB line came back as `parsed: null`).

compact_line is run on canned full records -- the complete bench records of earlier rounds committed under profiles/ -- and on a
synthetic worst case; the result must be ONE line of JSON under 4 KB that still carries the contract keys, `roofline` and
`cpu_baseline`."""
import glob
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")
CANNED = sorted(glob.glob(os.path.join(REPO, "profiles", "r05z_bench_*.json")))


def _records():
    for p in CANNED:
        txt = open(p).read().strip().splitlines()[-1]
        yield os.path.basename(p), json.loads(txt)


def test_canned_records_exist():
    assert len(CANNED) >= 5


@pytest.mark.parametrize("name,res", list(_records()))
def test_line_fits_and_keeps_the_contract(name, res):
    line = bench.compact_line(res, bench.REPORT_NAME)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4096, (name, len(line))
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, (name, k)
    assert d["value"] == res["value"] and d["ms_per_step"] == res["ms_per_step"]
    assert set(d["config"]) <= {"workload", "mode", "graphs_per_gpu", "global_batch", "avg_nodes_per_batch", "avg_edges_per_batch",
                                "parallelism", "step"}
    assert "model" not in d["config"] and " " not in d["config"].get("parallelism", "")
    if "roofline" in res:
        for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in d["roofline"], (name, k)
        assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 2e-3
    if "cpu_baseline" in res:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in d["cpu_baseline"], (name, k)
        assert len(d["cpu_baseline"]["sample"]) <= 160
    if "modes" in res and "fp32" in res["modes"]:
        assert d["value_fp32_contract"] == res["modes"]["fp32"]["value"]
        assert d["ms_per_step_fp32_contract"] == res["modes"]["fp32"]["ms_per_step"]
    if "modes" in res and "bf16" in res["modes"]:
        assert d["value_bf16"] == res["modes"]["bf16"]["value"]
    if isinstance(res.get("precision_vs_oracle"), dict) and "precision_vs_oracle" in d:
        for m, v in d["precision_vs_oracle"].items():
            assert set(v) <= {"loss_rel_err", "grad_rel_l2_worst", "grad_rel_l2_median"}
    assert d["report"] == bench.REPORT_NAME


def test_headline_fp32_mode_reports_itself_as_the_contract_value():
    _, res = next(_records())
    res = dict(res, config=dict(res["config"], mode="fp32"))
    res.pop("modes", None)
    d = json.loads(bench.compact_line(res))
    assert d["value_fp32_contract"] == res["value"]


def test_oversized_optional_blocks_are_dropped_not_the_contract():
    _, res = next(r for r in _records() if "roofline" in r[1] and "cpu_baseline" in r[1])
    res = json.loads(json.dumps(res))
    res["precision_vs_oracle"] = {"m%d" % i: {"loss_rel_err": 1e-5, "grad_rel_l2_worst": 0.5, "grad_rel_l2_median": 0.25} for i in range(80)}
    res["roofline"]["traffic_source"] = "x" * 5000
    res["cpu_baseline"]["sample"] = "y" * 5000
    res["cpu_baseline"].pop("sample_short", None)
    line = bench.compact_line(res, bench.REPORT_NAME)
    assert len(line) < bench.LINE_LIMIT
    d = json.loads(line)
    for k in CONTRACT + ("roofline", "cpu_baseline"):
        assert k in d
    assert "precision_vs_oracle" not in d


def test_pooled_kernel_names_match_rocprof_rows():
    assert bench.pooled_name("k_lin3r[fwd]") == bench.pooled_name("k_lin3r[dx]") == "k_lin3r"
    assert bench.pooled_name("k_lin1[dx]") == "k_lin1"
    assert bench.pooled_name("k_lin1[fwd+ln]") == "k_lin1[fwd+ln]"
    assert bench.pooled_name("k_lin3[dx]") == "k_lin3[dx]"          # forward and dX are different instantiations of k_lin3
    recs = [("k_lin3r[fwd]", 0.040, (31600, 300, 300, 0, 0, 0)), ("k_lin3r[dx]", 0.050, (31600, 300, 300, 0, 0, 0)),
            ("k_dw16", 0.039, (31600, 128, 512, 1, 1, 1))]
    import torch
    fam = bench.kernel_report(recs, 1.0, torch.bfloat16, torch.float32, pool=True)
    assert set(fam) == {"k_lin3r", "k_dw16[bf16]"}
    assert fam["k_lin3r"]["calls"] == 2 and abs(fam["k_lin3r"]["avg_us"] - 45.0) < 1e-6
    per = bench.kernel_report(recs, 1.0, torch.bfloat16, torch.float32)
    assert {"k_lin3r[fwd]", "k_lin3r[dx]"} <= set(per)
