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


def test_pmc_traffic_lookup_matches_workload_and_code_version(tmp_path, monkeypatch):
    """A committed FETCH_SIZE profile counts only for the same mode (se / pe), read length, genome size and code
    version: round 1's lookup took the paired-end file for the single-end probe kernel and kept quoting stale code."""
    prof = tmp_path / "profiles" / "r9"
    prof.mkdir(parents=True)

    def put(name, mode, code, probe_bytes):
        (prof / name).write_text(json.dumps({"mode": mode, "read_len": 150, "code_version": code, "genome_bp": 3.1e9,
                                             "reads_per_launch": 1_000_000,
                                             "kernels": {"seed_probe_kernel": {"hbm_read_bytes_per_launch": probe_bytes}}}))
    put("pmc_fetch_a_se.json", "se", bench.CODE_VERSION, 16.5e9)
    put("pmc_fetch_b_pe.json", "pe", bench.CODE_VERSION, 16.2e9)
    put("pmc_fetch_c_old.json", "se", "r0", 99e9)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("seed_probe_kernel", 1_000_000, 3_100_000_727, 150, "se") == round(16.5e9)
    assert bench.pmc_traffic("seed_probe_kernel", 500_000, 3_100_000_727, 150, "pe") == round(8.1e9)
    assert bench.pmc_traffic("seed_probe_kernel", 1_000_000, 800_000_000, 150, "se") is None   # another genome: no claim
    assert bench.pmc_traffic("seed_probe_kernel", 1_000_000, 3_100_000_727, 250, "se") is None
    assert bench.pmc_traffic("no_such_kernel", 1_000_000, 3_100_000_727) is None
    assert bench.pmc_traffic("seed_probe_kernel", 1_000_000, 3_100_000_727, 150, "se", code_version="r7") is None


def test_round1_profiles_are_stale_for_this_code():
    assert bench.pmc_traffic("search_se_kernel", 1_000_000, 3_100_000_727, 150, "se", code_version="r1-not-recorded") is None


def test_get_prime_ladder():
    """GetPrime (prime.cpp:11-21): the rungs the reference's table holds around the hg38 size."""
    assert bench.get_prime(100) == 101 and bench.get_prime(102) == 107
    assert bench.get_prime(5_250_000_000) == 5392814809
    assert bench.get_prime(5_392_814_810) == 5676647183
    slots, size = bench.default_slot_count([1000, 61], ["a", "bc"])
    assert size == (3 + 1000 + 17) + (4 + 61 + 2) and slots == bench.get_prime(int(size / 0.6))


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


def test_fastq_writer_of_the_file_to_file_run(tmp_path):
    import numpy as np
    reads = np.frombuffer(b"ACGTACGTAC" + b"TTTTTGGGGG", dtype=np.uint8)
    p = tmp_path / "r.fq"
    n = bench.write_fastq_fixed(str(p), reads, 2, 10)
    txt = p.read_bytes()
    assert n == len(txt)
    assert txt == b"@r00000000\nACGTACGTAC\n+\nIIIIIIIIII\n@r00000001\nTTTTTGGGGG\n+\nIIIIIIIIII\n"


def test_kernel_table_lists_the_launches_of_a_single_end_step():
    """Single-end: seed + probe run inside the search kernel (its algorithmic bytes are both stages'), then the DP launches,
    the finalize launches, the second pass and the general kernel; paired-end is one search kernel with the probe inside."""
    class FakeApi:
        class RESULT_DTYPE:
            itemsize = 28
    c = {"n_getblob": 190.0, "n_rowhop": 220.0, "n_extbases": 5000.0, "n_dptarget": 90.0}
    k = bench.kernel_table(FakeApi, False, 150, 1_000_000, [0.0, 30.0], c, 3.1e9, 4.6e10, stage_ms=[20.0, 8.0, 0.5, 0.5, 0.3, 0.1, 0.05])
    names = [x["kernel"] for x in k]
    assert names[:3] == ["search_se_kernel", "dp_kernel", "finalize_se_kernel"] and names[3].startswith("second pass") and names[4].startswith("general kernel")
    alg = 5 * 190 + 150 + 5 * 220 + 5000 + 28
    assert abs(k[0]["alg_bytes_per_read"] - alg) < 1e-6
    assert abs(k[0]["achieved_GBs"] - alg * 1e6 / 20e-3 / 1e9) < 0.01
    kp = bench.kernel_table(FakeApi, True, 150, 1_000_000, [6.4, 47.0], c, 3.1e9, 4.6e10)
    assert [x["kernel"] for x in kp] == ["search_pe_kernel"]   # seed + probe run inside it since round 3
    assert abs(kp[0]["alg_bytes_per_read"] - (alg + 90)) < 1e-6
