"""The ONE stdout line of bench.py must stay small enough for the driver to parse: the compact
headline is built from the full record by bench.headline(); the full record goes to bench_detail.json / stderr."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _detail():
    with open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")) as f:      # a real full record (27.8 KB: the one the driver could not parse)
        return json.load(f)


def test_headline_fits_and_keeps_the_contract():
    import bench
    d = _detail()
    assert len(json.dumps(d)) > 20000
    line = bench.headline(d)
    text = json.dumps(line)
    assert len(text) < bench.HEADLINE_LIMIT <= 4096
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in line, key
    assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"]
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    assert abs(line["roofline"]["frac"] - d["roofline"]["frac"]) < 1e-4
    for cfg in ("cfg3", "cfg4", "cfg5_on_1_gpu"):
        assert line[cfg]["iter_per_s"] > 0 and 0 < line[cfg]["frac_of_hbm_peak"] < 1
    assert line["exact_f32_iter_per_s"] > 0
    assert "cfg5_n8" in line["scaling_model_predicted_speedup"]
    assert "\n" not in text


def test_headline_sheds_optional_slots_rather_than_grow():
    import bench
    d = _detail()
    d["time_to_tol"] = d["time_to_tol"] * 40                      # something upstream grew
    d["config"]["workload"] = d["config"]["workload"] + " x" * 200
    line = bench.headline(d)
    assert len(json.dumps(line)) < bench.HEADLINE_LIMIT
    assert "roofline" in line and "cpu_baseline" in line and "value" in line


def test_headline_without_optional_legs():
    import bench
    d = _detail()
    for k in ("other_configs", "scaling_model", "time_to_tol", "parity", "cpu_baseline"):
        d[k] = None
    line = bench.headline(d)
    assert line["cpu_baseline"] is None and line["parity"] is None and "cfg3" not in line


def test_this_rounds_record_and_line_agree():
    """profiles/r05_bench_n1.json is the line bench.py printed, profiles/r05_bench_n1_detail.json the record it was built from: the same
    function gives the same line, and the fused KL-ADMM legs carry their stream counts (3 per AO-ADMM round, 4 for ADMM's stored S)."""
    import bench
    with open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")) as f:
        printed = json.loads(f.read().strip().splitlines()[-1])
    with open(os.path.join(ROOT, "profiles", "r05_bench_n1_detail.json")) as f:
        detail = json.load(f)
    assert len(json.dumps(printed)) < bench.HEADLINE_LIMIT
    rebuilt = bench.headline(detail)
    for key in ("value", "ms_per_step", "roofline", "cpu_baseline", "parity", "cfg3", "cfg4", "cfg5_on_1_gpu"):
        assert rebuilt[key] == printed[key], key
    legs = {leg["config"]: leg for leg in detail["other_configs"] if "config" in leg}
    ao = legs["aoadmm_kl_on_cfg3_shape"]["dominant_kernel"]
    assert ao["name"] == "kl_vaux_fused" and ao["algorithmic_bytes_per_launch"] == 3 * 16384 * 8192 * 4.0
    assert 0.3 < ao["frac"] < 1.0
