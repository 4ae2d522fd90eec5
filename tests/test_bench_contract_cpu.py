"""bench.py's host-side pieces that need no GPU: the metric string, the slot-count prime, the committed PMC traffic
lookup, the argument defaults the driver relies on, and the committed bench lines' keys."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_metric_is_baseline_json_metric():
    want = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert bench.metric_name(False, 150) == want
    assert "PE" in bench.metric_name(True, 150) and "250 bp SE" in bench.metric_name(False, 250)


def test_next_prime():
    assert [bench.next_prime(n) for n in (2, 3, 4, 90, 7919, 7920)] == [2, 3, 5, 97, 7919, 7927]


def test_defaults_are_one_gpu_and_the_headline_workload(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    for k in ("URMAP_BENCH_GENOME_MBP", "URMAP_BENCH_READS"):
        monkeypatch.delenv(k, raising=False)
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup, a.read_len, a.mode) == (1, 10, 2, 150, "se")
    assert a.genome_mbp == 3100 and a.reads_per_step == 1_000_000


def test_pmc_traffic_lookup_uses_committed_profiles():
    t = bench.pmc_traffic("search_se_kernel", 1_000_000, 3_100_000_727)
    assert t is not None and 1e10 < t < 1e11            # 25.5 GB per 1 M reads in profiles/r1
    assert bench.pmc_traffic("search_se_kernel", 500_000, 3_100_000_727) == round(t / 2)
    assert bench.pmc_traffic("search_se_kernel", 1_000_000, 800_000_000) is None   # another workload: no claim
    assert bench.pmc_traffic("no_such_kernel", 1_000_000, 3_100_000_727) is None


def test_committed_bench_lines_carry_the_contract_keys():
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}
    roof = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r1", "bench_*hg38scale_v1[01]*.json")))
    assert files
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        assert need <= set(d), (f, need - set(d))
        assert roof <= set(d["roofline"]), f
        assert "workload" in d["config"] and "model" not in d["config"]
        assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak"
        if "cpu_baseline" in d and d["cpu_baseline"]:
            assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]), f
