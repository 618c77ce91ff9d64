"""The bench line's contract (the driver parses ONE JSON line; the judge recomputes the roofline from it) checked on the line committed
with the round's profiles — no GPU, no bench run: the keys the driver and the review rely on are there, the derived figures follow from
the raw ones, and the traffic the line quotes is the committed PMC file's."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_line():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3_10M_bench.json")))
    assert paths, "no committed bench line under profiles/"
    return paths[-1], json.loads(open(paths[-1]).read().strip().splitlines()[-1])


def test_committed_bench_line_keeps_the_contract():
    path, d = latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (path, k)
    assert d["unit"] == "requests/s" and d["higher_is_better"] is True and d["n_gpus"] == 1 and d["dtype"] == "u8" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "configs[2]" in d["config"]["workload"] and "model" not in d["config"]
    n = d["config"]["requests_per_gpu"]
    assert abs(d["value"] - n / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6  # value = requests of the step / time of the step
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / the launch's average duration (HIP events inside the timed region)
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] / 1e3) / 1e9) / r["achieved"] < 1e-6
    assert r["avg_launch_ms"] * r["launches_per_step"] < d["ms_per_step"]  # the dominant kernel fits inside the step
    assert r["alg_bytes_per_step"] / (d["ms_per_step"] / 1e3) / 1e9 < r["peak"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "requests/s" and c["cores"] >= 1 and c["value"] > 0 and c["verdicts_match_gpu"] is True and "sample" in c


def test_quoted_traffic_is_the_committed_pmc_file():
    _, d = latest_line()
    r = d["roofline"]
    if r["traffic"] is None:
        return
    name = r["traffic_source"].split(" ")[0]
    tj = json.load(open(os.path.join(ROOT, name)))
    sk = [v for k, v in tj["kernels"].items() if "::filter_kernel<" in k]
    now = sum(sum(v["fetch_bytes"]) + sum(v["write_bytes"]) for v in sk) // max(1, sum(v["launches"] for v in sk))
    # (the line was written BEFORE the PMC passes of its own closing run replaced the file: the figure it quotes is the previous passes',
    # of the same kernel — the two agree to a fraction of a percent)
    assert sk and abs(r["traffic"] - now) / now < 0.01, (r["traffic"], now)
    assert 0.95 < r["traffic"] / r["alg_bytes_per_launch"] < 1.10  # no wasted re-reads in the streaming kernel (and it cannot read less than it must)
