"""GPU tests of the multi-device paths: read batches sharded over N mapping lanes / N ranks, one index
replica per device, no exchange between devices.  The reference fans reads over the threads of one process the same way
(map.cpp:58-61, seqsource.cpp:30-66).  On a box with one GPU every lane / rank is put on device 0 (URMAPX_FORCE_DEVICE
for the command line, URMAP_RANK_DEVICES for bench.py's ranks): the code path is the N-device one, only the device
numbers collapse."""
import gzip
import json
import os
import subprocess
import time
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "urmap_amd", "urmap")


def _golden_ufi(tmp_path, name="g.ufi.gz"):
    ufi = os.path.join(tmp_path, "g.ufi")
    with gzip.open(os.path.join(GOLD, name), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    return ufi


def _records(path):
    return [l for l in open(path, "rb").read().split(b"\n") if l and not l.startswith(b"@PG")]


def _force_env():
    import torch
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env["URMAPX_FORCE_DEVICE"] = "0"
    return env


@pytest.mark.parametrize("gpus,streams,batch", [(2, 1, 64), (2, 2, 50), (3, 2, 17)])
def test_cli_map_on_two_devices_reproduces_reference_sam(tmp_path, gpus, streams, batch):
    """`urmap -map ... -gpus N -streams K`: batch b -> lane b mod (N K) on device b mod N; the SAM is the reference's
    golden SAM, in input order."""
    ufi = _golden_ufi(tmp_path)
    out = os.path.join(tmp_path, "out.sam")
    r = subprocess.run([EXE, "-map", os.path.join(GOLD, "se150.fq"), "-ufi", ufi, "-samout", out, "-batch", str(batch),
                        "-gpus", str(gpus), "-streams", str(streams)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300, env=_force_env())
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert _records(out) == [l for l in open(os.path.join(GOLD, "se150.sam"), "rb").read().split(b"\n") if l]
    assert f"({gpus} GPUs)".encode() in r.stderr


def test_cli_map2_on_two_devices_reproduces_reference_sam_and_tab(tmp_path):
    ufi = _golden_ufi(tmp_path)
    out, tab = os.path.join(tmp_path, "out.sam"), os.path.join(tmp_path, "out.tab")
    r = subprocess.run([EXE, "-map2", os.path.join(GOLD, "pe150_1.fq"), "-reverse", os.path.join(GOLD, "pe150_2.fq"), "-ufi", ufi,
                        "-samout", out, "-tabbedout", tab, "-batch", "60", "-gpus", "2", "-streams", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=_force_env())
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert _records(out) == [l for l in open(os.path.join(GOLD, "pe150.sam"), "rb").read().split(b"\n") if l]
    assert open(tab, "rb").read() == open(os.path.join(GOLD, "pe150.tab"), "rb").read()


def test_two_index_replicas_two_contexts_through_the_library(tmp_path):
    """urmapx_index_replicate + one context per replica: batches of the golden reads alternate between the two
    contexts and the records equal the golden SAM."""
    import torch
    from urmap_amd import api
    ufi = _golden_ufi(tmp_path)
    second = 1 if torch.cuda.device_count() > 1 else 0
    idx0 = api.Index.open(ufi).upload(0)
    idx1 = idx0.replicate(second)
    maps = [api.Mapper(idx0, device=0), api.Mapper(idx1, device=second)]
    labels, bases, offs, quals = api.read_fastq_arrays(os.path.join(GOLD, "se150.fq"))
    n = len(labels)
    sam = []
    for b, lo in enumerate(range(0, n, 37)):
        hi = min(n, lo + 37)
        o = offs[lo:hi + 1] - offs[lo]
        bb = bases[int(offs[lo]):int(offs[hi])]
        res, ops = maps[b % 2].map_se(bb, o)
        sam.append(idx0.sam_se(res, ops, labels[lo:hi], bb, o, quals[int(offs[lo]):int(offs[hi])]))
    got = [l for l in b"".join(sam).split(b"\n") if l]
    want = [l for l in open(os.path.join(GOLD, "se150.sam"), "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    assert got == want
    for m in maps:
        m.close()
    idx1.close()


def test_index_streamed_from_the_file_to_the_device_and_replicated_device_to_device(tmp_path):
    """urmapx_index_open_device (what the command line loads with): the .ufi's arrays go from the file to the device through
    page-locked buffers, no host copy is kept; the resident table validates like the one uploaded from host arrays, maps the
    golden reads to the golden SAM, and urmapx_index_replicate copies it device to device (no host arrays to upload from).
    A file cut short or with a damaged magic word is refused as urmapx_index_open refuses it."""
    import torch
    from urmap_amd import api
    ufi = _golden_ufi(tmp_path)
    second = 1 if torch.cuda.device_count() > 1 else 0
    idx_h = api.Index.open(ufi).upload(0)
    idx_d = api.Index.open_device(ufi, 0)
    ok_h, rep_h = idx_h.validate()
    ok_d, rep_d = idx_d.validate()
    assert ok_h and ok_d
    for k in ("slots", "heads", "positions", "used", "reached"):
        assert rep_h[k] == rep_d[k], k
    assert idx_d.chain_row_bytes() == idx_h.chain_row_bytes() > 0
    idx_r = idx_d.replicate(second)
    ok_r, rep_r = idx_r.validate()
    assert ok_r and rep_r["positions"] == rep_h["positions"] and idx_r.chain_row_bytes() == idx_h.chain_row_bytes()
    labels, bases, offs, quals = api.read_fastq_arrays(os.path.join(GOLD, "se150.fq"))
    want = [l for l in open(os.path.join(GOLD, "se150.sam"), "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    for idx, dev in ((idx_d, 0), (idx_r, second)):
        m = api.Mapper(idx, device=dev)
        res, ops = m.map_se(bases, offs)
        assert [l for l in idx.sam_se(res, ops, labels, bases, offs, quals).split(b"\n") if l] == want
        m.close()
    out = os.path.join(tmp_path, "f.sam")
    rep = api.map_files(idx_d, os.path.join(GOLD, "se150.fq"), samout=out, gpus=2 if second else 1, streams=2, batch=50, cmdline="t")
    assert rep["reads"] == len(want) and _records(out) == [l for l in open(os.path.join(GOLD, "se150.sam"), "rb").read().split(b"\n") if l]
    for i in (idx_r, idx_d, idx_h):
        i.close()
    raw = open(ufi, "rb").read()
    cut = os.path.join(tmp_path, "cut.ufi")
    open(cut, "wb").write(raw[: len(raw) // 2])
    bad = os.path.join(tmp_path, "bad.ufi")
    open(bad, "wb").write(raw[:-4] + b"XXXX")
    for p in (cut, bad):
        with pytest.raises(api.UrmapxError) as e:
            api.Index.open_device(p, 0)
        assert e.value.code == api.E_FORMAT
        with pytest.raises(api.UrmapxError) as e2:
            api.Index.open(p)
        assert e2.value.code == api.E_FORMAT


@pytest.mark.parametrize("broadcast", [False, True])
def test_bench_starts_its_own_two_ranks(broadcast):
    """`python bench.py --gpus 2` with no launcher around it: two child ranks, one JSON line with n_gpus 2, results of
    rank 0's last batch bit-identical to the oracle.  broadcast: the index reaches rank 1 through the collective
    broadcast of the resident arrays (the branch the ranks take over RCCL on a multi-GPU node) instead of a file."""
    env = dict(os.environ)
    if broadcast:
        env["URMAP_BENCH_BROADCAST"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genome-mbp", "40", "--reads-per-step", "20000",
                        "--steps", "2", "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks"]["world"] == 2
    assert d["parity"]["bit_identical_to_oracle"], d["parity"]
    assert d["value"] > 0 and d["scaling"] == "weak"
    import torch
    assert d["config"]["ranks"]["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    if broadcast:
        assert "broadcast of the resident arrays" in d["config"]["setup_s"]["how"]


def test_eight_ranks_on_the_one_device_as_the_scaling_run_launches_them():
    """BASELINE config 4's launch (`bench.py --gpus 8`: 8 ranks, index built by rank 0 and broadcast, reads sharded by rank) on a box
    with one GPU: all ranks share the device, the collectives go over gloo, the index takes the broadcast branch.  The line must say
    n_gpus 8, carry every rank's own step and set-up times, rank 0's results bit-identical to the oracle, and the file-to-file leg
    over the 8 'devices'.  Set-up per rank at this size bounds what the full-scale launch needs (printed for the record)."""
    env = dict(os.environ)
    env["URMAP_BENCH_BROADCAST"] = "1"
    env["URMAP_BENCH_E2E_READS"] = "160000"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--genome-mbp", "40", "--reads-per-step", "20000",
                        "--steps", "2", "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, env=env)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["ranks"]["world"] == 8 and d["scaling"] == "weak"
    assert d["parity"]["bit_identical_to_oracle"], d["parity"]
    assert d["config"]["index_validated"] is True
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and len(pr["setup_s"]) == 8 and 0 <= pr["slowest_rank"] < 8
    assert pr["ms_per_step_max"] == max(pr["ms_per_step"]) and pr["ms_per_step_min"] == min(pr["ms_per_step"]) > 0
    assert d["ms_per_step"] >= pr["ms_per_step_max"] * 0.999  # the line's step time is the slowest rank's (plus the barrier)
    assert d["value"] == pytest.approx(8 * d["steps"] * 20000 / (d["ms_per_step"] * d["steps"] * 1e-3), rel=1e-3)
    assert "broadcast of the resident arrays" in d["config"]["setup_s"]["how"]
    e2e = d["e2e"]
    assert "error" not in e2e and e2e["gpus"] == 8 and e2e["sam_records_identical_to_oracle"], e2e
    assert e2e["sharded"]["shards"] == 8 and e2e["sharded"]["cat_of_shards_equals_the_one_file"]
    print(f"8 ranks on one device: wall {wall:.0f} s, set-up per rank {pr['setup_s']}, ms per step per rank {pr['ms_per_step']}")


def test_broadcast_of_a_table_larger_than_one_piece():
    """The 8-GPU launch's index placement on a box with one GPU: rank 0 builds the table, the other rank
    waits in the collective, the resident arrays travel in 1 GiB pieces (ranks.broadcast_bytes) -- here at 1.2 Gbp (11.6 GB
    of index, 12 pieces; URMAP_TEST_FULLSCALE=1: the 3.1 Gbp genome of the bench, 31.6 GB, 29 pieces).  The line carries
    what a rank holds and how long the broadcast took; both ranks map their own reads bit-identically."""
    mbp = 3100 if os.environ.get("URMAP_TEST_FULLSCALE") else 1200
    env = dict(os.environ)
    env["URMAP_BENCH_BROADCAST"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genome-mbp", str(mbp), "--reads-per-step", "200000",
                        "--steps", "2", "--warmup", "1", "--no-e2e"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    rk = d["config"]["ranks"]
    assert d["n_gpus"] == 2 and rk["world"] == 2 and d["parity"]["bit_identical_to_oracle"], d["parity"]
    slots, bp = d["config"]["slots"], d["config"]["genome_bp"]
    assert rk["index_bytes_per_rank"] >= 5 * slots + bp + bp // 2  # slot table + sequence + its packed copy
    assert rk["broadcast_pieces"] == (5 * slots + 8 + (1 << 30) - 1) // (1 << 30) + (bp + 4096 + (1 << 30) - 1) // (1 << 30) >= (29 if mbp == 3100 else 12)
    assert rk["broadcast_s"] > 0
    assert "broadcast of the resident arrays" in d["config"]["setup_s"]["how"]


def test_map_files_library_call_on_a_resident_index(tmp_path):
    """urmapx_map_files (the command line's cmd_map as a library call) on an index that is already resident: golden SAM,
    the report's counters, and a FASTQ error coming back as a message instead of an exit."""
    from urmap_amd import api
    ufi = _golden_ufi(tmp_path)
    idx = api.Index.open(ufi).upload(0)
    out = os.path.join(tmp_path, "lib.sam")
    rep = api.map_files(idx, os.path.join(GOLD, "se150.fq"), samout=out, batch=90, streams=2, cmdline="test")
    assert _records(out) == [l for l in open(os.path.join(GOLD, "se150.sam"), "rb").read().split(b"\n") if l]
    assert rep["reads"] == 400 and rep["mapped_q"] + rep["mapped_lowq"] + rep["unmapped"] == 400 and rep["lanes"] == 2
    out2 = os.path.join(tmp_path, "lib2.sam")
    rep2 = api.map_files(idx, os.path.join(GOLD, "pe150_1.fq"), os.path.join(GOLD, "pe150_2.fq"), samout=out2, batch=64)
    assert _records(out2) == [l for l in open(os.path.join(GOLD, "pe150.sam"), "rb").read().split(b"\n") if l]
    assert rep2["reads"] == 600
    bad = os.path.join(tmp_path, "bad.fq")
    open(bad, "w").write("@a\nACGT\n+\nII\n")
    with pytest.raises(api.UrmapxError) as e:
        api.map_files(idx, bad, samout=os.path.join(tmp_path, "x.sam"))
    assert e.value.code == api.E_FORMAT
    idx.close()


def _cat(paths):
    return b"".join(open(p, "rb").read() for p in paths)


@pytest.mark.parametrize("gpus,shards,batch,host_text", [(2, 2, 64, False), (1, 3, 40, False), (2, 4, 33, False), (2, 2, 64, True)])
def test_cli_samshards_concatenate_to_the_one_file(tmp_path, gpus, shards, batch, host_text):
    """`-samout out.sam -samshards N` (round 4): N pipelines side by side, each over its part of the input with its own SAM
    file and writer; `cat out.sam.0 .. out.sam.N-1` is byte for byte the file a run without -samshards writes (which is the
    reference's golden SAM).  The reference appends every record to one file under one lock (output1.cpp:10-16)."""
    ufi = _golden_ufi(tmp_path)
    one, out = os.path.join(tmp_path, "one.sam"), os.path.join(tmp_path, "out.sam")
    env = _force_env()
    if host_text:
        env["URMAPX_HOST_TEXT"] = "1"
    base = [EXE, "-map", os.path.join(GOLD, "se150.fq"), "-ufi", ufi, "-batch", str(batch), "-gpus", str(gpus), "-streams", "2"]
    r1 = subprocess.run(base + ["-samout", one], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert r1.returncode == 0, r1.stderr.decode()[-2000:]
    r2 = subprocess.run(base + ["-samout", out, "-samshards", str(shards)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert r2.returncode == 0, r2.stderr.decode()[-2000:]
    parts = [f"{out}.{s}" for s in range(shards)]
    assert all(os.path.exists(p) for p in parts) and not os.path.exists(out)
    strip_pg = lambda b: b"\n".join(l for l in b.split(b"\n") if not l.startswith(b"@PG"))  # the command lines differ by the option itself
    assert strip_pg(_cat(parts)) == strip_pg(open(one, "rb").read())
    assert sum(os.path.getsize(p) > 0 for p in parts) >= min(shards, 2)  # the work really was split
    assert _records(one) == [l for l in open(os.path.join(GOLD, "se150.sam"), "rb").read().split(b"\n") if l]
    # the report counts every shard's reads
    reads = [l for l in r2.stderr.decode().split("\n") if "Reads (" in l]
    assert reads and reads == [l for l in r1.stderr.decode().split("\n") if "Reads (" in l]


def test_cli_samshards_pairs_and_tab(tmp_path):
    ufi = _golden_ufi(tmp_path)
    out, tab = os.path.join(tmp_path, "out.sam"), os.path.join(tmp_path, "out.tab")
    r = subprocess.run([EXE, "-map2", os.path.join(GOLD, "pe150_1.fq"), "-reverse", os.path.join(GOLD, "pe150_2.fq"), "-ufi", ufi,
                        "-samout", out, "-tabbedout", tab, "-batch", "60", "-gpus", "2", "-streams", "1", "-samshards", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=_force_env())
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    sam = _cat([out + ".0", out + ".1"])
    assert os.path.getsize(out + ".1") > 0
    assert [l for l in sam.split(b"\n") if l and not l.startswith(b"@PG")] == [l for l in open(os.path.join(GOLD, "pe150.sam"), "rb").read().split(b"\n") if l]
    assert _cat([tab + ".0", tab + ".1"]) == open(os.path.join(GOLD, "pe150.tab"), "rb").read()


def test_cli_samshards_of_input_that_cannot_be_cut(tmp_path):
    """.gz input has no record boundary to seek to: everything goes to shard 0, the other shards are empty files -- `cat` of the
    shards is still the one file."""
    ufi = _golden_ufi(tmp_path)
    gz = os.path.join(tmp_path, "r.fq.gz")
    with open(os.path.join(GOLD, "se150.fq"), "rb") as f, gzip.open(gz, "wb") as z:
        z.write(f.read())
    out = os.path.join(tmp_path, "out.sam")
    r = subprocess.run([EXE, "-map", gz, "-ufi", ufi, "-samout", out, "-samshards", "2", "-gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300, env=_force_env())
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert os.path.getsize(out + ".1") == 0
    assert _records(out + ".0") == [l for l in open(os.path.join(GOLD, "se150.sam"), "rb").read().split(b"\n") if l]


def test_lane_threads_are_placed_on_their_devices_numa_node(tmp_path, monkeypatch):
    """urmapx_map_files pins each lane's host thread to the CPUs of its device's NUMA node (read from sysfs by PCI bus id) and says
    where in the report.  On a box without NUMA information the devices are '@any'; with the node forced (test aid) the threads are
    pinned for real -- same records either way."""
    from urmap_amd import api
    ufi = _golden_ufi(tmp_path)
    idx = api.Index.open(ufi).upload(0)
    want = [l for l in open(os.path.join(GOLD, "se150.sam"), "rb").read().split(b"\n") if l]
    out = os.path.join(tmp_path, "a.sam")
    rep = api.map_files(idx, os.path.join(GOLD, "se150.fq"), samout=out, batch=90, streams=2, cmdline="test")
    assert rep["placement"].decode().startswith("gpu0@") and "reader+writer@" in rep["placement"].decode()
    assert _records(out) == want
    if os.path.exists("/sys/devices/system/node/node0/cpulist"):
        monkeypatch.setenv("URMAPX_TEST_NUMA_NODE", "0")
        rep = api.map_files(idx, os.path.join(GOLD, "se150.fq"), samout=out, batch=90, streams=2, cmdline="test", sam_shards=2)
        pl = rep["placement"].decode()
        assert pl.count("gpu0@node0") == 2 and pl.count("reader+writer@node0") == 2 and " | " in pl, pl
        assert [l for k in range(2) for l in _records(f"{out}.{k}")] == want
    monkeypatch.setenv("URMAPX_NO_NUMA_PIN", "1")
    rep = api.map_files(idx, os.path.join(GOLD, "se150.fq"), samout=out, batch=90, streams=2, cmdline="test")
    assert rep["placement"].decode() == "gpu0@any; reader+writer@any"
    idx.close()


def test_map_files_discard_and_device_stage_times(tmp_path):
    """urmapx_map_options.discard_sam: the text is made, copied back and dropped (what the device lanes sustain without an
    output medium); the report splits a lane's stream time into copy in / parse / map / SAM text / copy out."""
    from urmap_amd import api
    ufi = _golden_ufi(tmp_path)
    idx = api.Index.open(ufi).upload(0)
    out = os.path.join(tmp_path, "never.sam")
    rep = api.map_files(idx, os.path.join(GOLD, "se150.fq"), samout=out, batch=90, streams=2, cmdline="test", discard_sam=True)
    assert not os.path.exists(out)
    assert rep["medium"] == b"discarded" and rep["reads"] > 0 and rep["text_on_device"] == 1
    stages = [rep[k] for k in ("dev_h2d_s", "dev_parse_s", "dev_map_s", "dev_format_s", "dev_d2h_s")]
    assert all(x > 0 for x in stages) and sum(stages) <= rep["gpu_s"] * 1.05 + 1e-3, (stages, rep["gpu_s"])
    rep2 = api.map_files(idx, os.path.join(GOLD, "se150.fq"), samout=out, batch=90, streams=2, cmdline="test")
    assert {k: rep[k] for k in ("reads", "mapped_q", "mapped_lowq", "unmapped")} == {k: rep2[k] for k in ("reads", "mapped_q", "mapped_lowq", "unmapped")}
