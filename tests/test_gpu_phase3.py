"""Phase 3 of Search_Lo (search1m6.cpp:162-171: AlignHSP over the HSPs of phases 1-2 when the best of them is long) parked like
phase 6 (round 5, URMAPX_PARK_PHASE3=1; off by default: it measured slower): the first launch of the search kernel ends such a read
there, dp_kernel runs the flank DPs, a second launch resumes the read from its parked state.  Every field against the oracle, on reads
chosen so that many of them park; the same with phase 3 kept inside the kernel (the default), and with a parking lot / job array too
small for the batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case(small_case):
    from urmap_amd import api, synth
    from conftest import reads_to_arrays
    import os
    os.environ["URMAPX_KEEP_ROWINFO"] = "1"  # the parked variant looks rows up in the info entries, which an upload otherwise drops once slot16 is built
    try:
        idx = api.Index.open(small_case["ufi"]).upload(0)
    finally:
        del os.environ["URMAPX_KEEP_ROWINFO"]
    assert idx.chain_row_bytes() > 0  # phase 3 is parked on indexes that carry the row layout
    # indels put an HSP (not a full-length hit) on most reads: phase 3 aligns it
    reads = synth.make_reads(31, small_case["genome"], 2500, read_len=150, sub=0.01, ins=0.004, dele=0.004, random_frac=0.03)
    reads += synth.make_reads(32, small_case["genome"], 800, read_len=250, sub=0.03, ins=0.006, dele=0.006)
    reads += synth.make_reads(33, small_case["genome"], 500, read_len=100, sub=0.02, ins=0.005, dele=0.005)
    reads += synth.make_reads(34, small_case["genome"], 300, read_len=300, sub=0.02, ins=0.004, dele=0.004)
    out = {"index": idx, "mapper": api.Mapper(idx, device=0), "oi": small_case["oracle_index"], "sets": []}
    lo = 0
    for n in (2500, 800, 500, 300):  # one batch per read-length class: 192, 256, 128, 320
        part = reads[lo:lo + n]
        lo += n
        bases, offs = reads_to_arrays(part)
        ores, opaths, _ = out["oi"].map_se(bases, offs, threads=4)
        out["sets"].append((bases, offs, ores, opaths))
    return out


def _check(m, bases, offs, ores, opaths):
    from test_gpu_parity import compare_results
    gres, gops = m.map_se(bases, offs)
    compare_results(gres, gops, ores, opaths)
    return gres


def test_reads_park_at_phase_3_and_come_back_with_the_oracles_results(case, monkeypatch):
    monkeypatch.setenv("URMAPX_PARK_PHASE3", "1")
    m = case["mapper"]
    for bases, offs, ores, opaths in case["sets"]:
        n = len(offs) - 1
        _check(m, bases, offs, ores, opaths)
        ms, st = m.phase3()
        assert st[1] > n // 20 and st[0] >= st[1], (n, st)  # reads parked, at least one DpJob each
        assert ms[0] > 0 and ms[1] > 0 and ms[2] > 0, ms
        assert abs(sum(ms) - m.stage_ms()[0]) < 0.05 * max(1e-3, sum(ms)) + 0.02  # the three launches are the search stage
    assert any((o[2]["exit_phase"] == 6).sum() > 100 for o in case["sets"])


def test_phase_3_inside_the_kernel_gives_the_same(case, monkeypatch):
    monkeypatch.delenv("URMAPX_PARK_PHASE3", raising=False)
    m = case["mapper"]
    for bases, offs, ores, opaths in case["sets"][:2]:
        _check(m, bases, offs, ores, opaths)
        ms, st = m.phase3()
        assert st == [0, 0], st


@pytest.mark.parametrize("env", [{"URMAPX_TEST_P3_FIN_CAP": "7"}, {"URMAPX_TEST_P3_JOBS_CAP": "9"}, {"URMAPX_TEST_P3_FIN_CAP": "0"}])
def test_no_room_to_park_sends_the_read_to_the_second_pass(case, monkeypatch, env):
    """a parking lot of 7 reads / a job array of 9 jobs: the reads that find no room are flagged and mapped again by the second
    pass (which aligns inside the kernel) -- same results"""
    monkeypatch.setenv("URMAPX_PARK_PHASE3", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m = case["mapper"]
    bases, offs, ores, opaths = case["sets"][0]
    _check(m, bases, offs, ores, opaths)
    ms, st = m.phase3()
    assert st[0] > 100  # the job counter keeps counting what the reads asked for


def test_two_contexts_side_by_side(case, monkeypatch):
    """two contexts on one device keep their parking lots apart"""
    import threading
    monkeypatch.setenv("URMAPX_PARK_PHASE3", "1")
    from urmap_amd import api
    bases, offs, ores, opaths = case["sets"][0]
    ms = [api.Mapper(case["index"], device=0) for _ in range(2)]
    outs = [None, None]

    def run(k):
        outs[k] = ms[k].map_se(bases, offs)
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    from test_gpu_parity import compare_results
    for g, ops in outs:
        compare_results(g, ops, ores, opaths)
