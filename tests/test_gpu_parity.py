"""GPU parity tests: the HIP path (through the C ABI, liburmapx.so) against the CPU oracle on the same
seeded inputs.  Bit-exact: slots, tallies, positions, hit position/strand/score/MAPQ, alignment paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(small_case):
    from urmap_amd import api
    idx = api.Index.open(small_case["ufi"]).upload(0)
    m = api.Mapper(idx, device=0, method=6)
    assert m.arch.startswith("gfx950"), m.arch
    return {"index": idx, "mapper": m}


def mutate_edge_reads(reads, seed):
    """N's, lower case, IUPAC and 'u' letters sprinkled into a few reads."""
    rng = np.random.default_rng(seed)
    out = list(reads)
    for k in range(0, len(out), 17):
        lab, s, q = out[k]
        s = s.copy()
        s[int(rng.integers(0, len(s)))] = ord("N")
        out[k] = (lab, s, q)
    for k in range(3, len(out), 61):
        lab, s, q = out[k]
        out[k] = (lab, s | 0x20, q)
    for k in range(5, len(out), 97):
        lab, s, q = out[k]
        s = s.copy()
        s[int(rng.integers(0, len(s)))] = ord("R")
        s[int(rng.integers(0, len(s)))] = ord("u")
        out[k] = (lab, s, q)
    return out


def compare_results(gres, gops, ores, opaths):
    from urmap_amd import api
    assert (gres["status"] == 0).all(), f"status bits set: {np.unique(gres['status'])}"
    for name in ("dbpos", "seq_index", "coord", "score", "second", "mapq", "hit_count", "exit_phase"):
        a = gres[name].astype(np.int64)
        b = ores[name].astype(np.int64)
        bad = np.nonzero(a != b)[0]
        assert len(bad) == 0, f"{name}: {len(bad)} mismatches, first read {bad[0]}: gpu {a[bad[0]]} oracle {b[bad[0]]}"
    mapped = ores["dbpos"] != 0xFFFFFFFF
    assert (gres["plus"][mapped] == ores["plus"][mapped]).all()
    for i in np.nonzero(mapped)[0]:
        gp = api.decode_path(gops[int(gres["path_off"][i]): int(gres["path_off"][i]) + int(gres["path_nops"][i])])
        assert gp == opaths[i], f"read {i}: path gpu {gp!r} oracle {opaths[i]!r}"


def test_seed_probe_matches_oracle(small_case, gpu):
    """SetSlotsVec + GetBlob for every k-mer of both strands (state1.cpp:396-438, ufindex.h:184-187)."""
    import oracle_lib as ol
    from urmap_amd import synth
    from conftest import reads_to_arrays
    oi = small_case["oracle_index"]
    reads = mutate_edge_reads(synth.make_reads(7, small_case["genome"], 300, read_len=150), 1)
    reads += synth.make_reads(8, small_case["genome"], 50, read_len=250, sub=0.04)
    reads += synth.make_reads(9, small_case["genome"], 50, read_len=37)
    bases, offs = reads_to_arrays(reads)
    slots, tallies, positions = gpu["mapper"].seed_probe(bases, offs)
    blob = oi.blob()
    W = oi.word_length
    for r, (_, seq, _) in enumerate(reads):
        L = len(seq)
        rc = np.zeros(L, np.uint8)
        ol.lib().uo_revcomp(np.ascontiguousarray(seq).ctypes.data, L, rc.ctypes.data)
        for strand, s in ((0, seq), (1, rc)):
            want = oi.slots_vec(s)
            base = 2 * int(offs[r]) + strand * L
            got = slots[base: base + L - W + 1]
            assert (got == want).all(), f"read {r} strand {strand}: slots differ"
            valid = want != np.iinfo(np.uint64).max
            wt = np.zeros(len(want), np.uint8)
            wp = np.full(len(want), 0xFFFFFFFF, np.uint32)
            sv = want[valid].astype(np.int64)
            wt[valid] = blob[5 * sv]
            wp[valid] = (blob[5 * sv + 1].astype(np.uint32) | (blob[5 * sv + 2].astype(np.uint32) << 8)
                         | (blob[5 * sv + 3].astype(np.uint32) << 16) | (blob[5 * sv + 4].astype(np.uint32) << 24))
            assert (tallies[base: base + len(want)] == wt).all(), f"read {r} strand {strand}: tallies differ"
            gp = positions[base: base + len(want)]
            assert (gp[valid] == wp[valid]).all(), f"read {r} strand {strand}: positions differ"


@pytest.mark.parametrize("pair", [False, True])
def test_viterbi_matches_oracle(gpu, pair, monkeypatch):
    """State1::Viterbi + TraceBackBitMem (viterbi.cpp:11-261): score and full path, Left/Right variants.  pair: two problems
    per wavefront through VFlank (viterbi_dev.h), the interior row blocks of both in packed int16 -- what dp_kernel runs."""
    import oracle_lib as ol
    if pair:
        monkeypatch.setenv("URMAPX_VITERBI_PAIR", "1")
    else:
        monkeypatch.delenv("URMAPX_VITERBI_PAIR", raising=False)
    rng = np.random.default_rng(5)
    pairs, flags = [], []
    for k in range(400):
        la = int(rng.integers(1, 140))
        extra = int(rng.integers(0, 30))
        core = rng.integers(0, 4, size=la + extra)
        t = np.frombuffer(b"ACGT", np.uint8)[core]
        q = list(t[:la]) if k % 2 else list(t[extra:extra + la])
        # edits
        i = 0
        out = []
        while i < len(q):
            x = rng.random()
            if x < 0.05:
                out.append(int(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4)]))
            elif x < 0.08:
                pass
            elif x < 0.11:
                out.append(q[i]); out.append(int(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4)]))
            else:
                out.append(q[i])
            i += 1
        a = bytes(out)
        if len(a) == 0:
            a = b"A"
        lb = len(a) + 24 + (k % 2)
        tt = t.tobytes()
        b = (tt + bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=lb)]))[:lb]
        pairs.append((a, b))
        flags.append(1 if k % 2 == 0 else 2)
    # long flanks (many interior row blocks), adjacent problems of different length, heavy edits, a repeat-like target
    for k in range(120):
        la = int(rng.integers(60, 300))
        core = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=la + 40)]
        q = core[20:20 + la].copy() if k % 2 else core[:la].copy()
        nm = int(rng.integers(0, max(1, la // (4 + k % 9))))
        q[rng.integers(0, la, size=nm)] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=nm)]
        qq = list(q)
        for _ in range(k % 4):
            x = int(rng.integers(1, len(qq) - 1))
            if rng.random() < 0.5:
                del qq[x:x + int(rng.integers(1, 4))]
            else:
                qq[x:x] = list(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(rng.integers(1, 4)))])
        a = bytes(qq)
        lb = len(a) + 24
        if k % 2:
            b = (core[20 - 24:20 + la + 40].tobytes())[:lb] if 20 - 24 >= 0 else core.tobytes()[:lb]
        else:
            b = core.tobytes()[:lb]
        b = (b + bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=lb)]))[:lb]
        pairs.append((a, b))
        flags.append(1 if k % 2 else 2)
    # degenerate shapes
    pairs += [(b"ACGT", b""), (b"A", b"A"), (b"A", b"CCCCCCCCCCCCCCCCCCCCCCCCC"), (b"ACGTACGTAC", b"ACGTACGTAC"),
              (b"ACGTACGTAC" * 5, (b"ACGTACGTAC" * 5)[:30])]
    flags += [2, 1, 1, 3, 2]
    # bands wider than one wavefront: flank windows clipped at the end of the sequence store (alignhsp.cpp:143-145)
    # and the paired-end rescue's whole-read DP against a 1024+2*QL window (scan.cpp:14-39)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for la, lb in ((100, 20), (140, 3), (90, 60), (150, 1324), (150, 1200), (250, 1524), (64, 200)):
        t = acgt[rng.integers(0, 4, size=lb)]
        q = acgt[rng.integers(0, 4, size=la)]
        if lb > la + 50:
            st = int(rng.integers(0, lb - la))
            q = t[st:st + la].copy()
            q[rng.integers(0, la, size=la // 20)] = acgt[rng.integers(0, 4, size=la // 20)]
        for fl in (0, 1, 2, 3):
            pairs.append((q.tobytes(), t.tobytes()))
            flags.append(fl)
    # whole-read DP of the paired-end rescue: the alignment's last column swept across the 64-column chunk borders of
    # the wide path (regression: a cross-lane shift under a lane-dependent select lost the chunk's first column)
    t = acgt[rng.integers(0, 4, size=1024)]
    for st in list(range(170, 186)) + list(range(230, 250)):
        q = t[st:st + 120].copy()
        q[rng.integers(0, 120, size=8)] = acgt[rng.integers(0, 4, size=8)]
        pairs.append((q.tobytes(), t.tobytes()))
        flags.append(3)
    scores, status, paths = gpu["mapper"].viterbi_batch(pairs, flags)
    import itertools
    for k, ((a, b), fl) in enumerate(zip(pairs, flags)):
        s, p = ol.viterbi(a, b, bool(fl & 1), bool(fl & 2))
        assert float(scores[k]) == s, f"case {k}: score gpu {scores[k]} oracle {s}"
        runs = sum(1 for _ in itertools.groupby(p))
        if runs > 96:  # more runs than URMAPX_MAX_PATH_OPS: must be flagged, never truncated silently
            assert status[k] == 0x04, f"case {k}: {runs} runs, status {status[k]}"
            continue
        assert status[k] == 0, f"case {k}: status {status[k]}"
        assert paths[k] == p, f"case {k}: path gpu {paths[k]} oracle {p}"


@pytest.mark.parametrize("read_len,sub,indel,n", [(150, 0.01, 0.001, 3000), (250, 0.04, 0.01, 1500),
                                                   (100, 0.02, 0.004, 1500), (30, 0.0, 0.0, 500),
                                                   (300, 0.03, 0.008, 1000), (400, 0.02, 0.006, 600),
                                                   (512, 0.03, 0.004, 400), (700, 0.02, 0.004, 300),
                                                   (1024, 0.02, 0.003, 200)])  # kernel classes: 150/100/30: 192 bases, 250: 256, 300: 320, 400/512: 512, 700/1024: 1024
def test_map_se_matches_oracle(small_case, gpu, read_len, sub, indel, n):
    """State1::Search end to end (search1.cpp:7-24): top hit, scores, MAPQ, path -- bit-exact."""
    from urmap_amd import synth
    from conftest import reads_to_arrays
    reads = synth.make_reads(1000 + read_len, small_case["genome"], n, read_len=read_len, sub=sub, ins=indel / 2,
                             dele=indel / 2, random_frac=0.03)
    reads = mutate_edge_reads(reads, read_len)
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = small_case["oracle_index"].map_se(bases, offs, threads=4)
    gres, gops = gpu["mapper"].map_se(bases, offs)
    compare_results(gres, gops, ores, opaths)


@pytest.fixture(scope="module")
def dense_case(workdir):
    """An index at load factor 0.95: free slots are rare, so collision chains need the two-step long links
    (TALLY_LONG_MINE 253 / TALLY_LONG_OTHER 125: ufindex.cpp:256-300, walked by GetRow_Blob ufindex.cpp:905-925).  None
    of the other fixtures contains such a slot."""
    import os
    import oracle_lib as ol
    from urmap_amd import api, synth
    g = synth.make_genome(77, [400000, 150000], repeat_frac=0.15, n_families=6, n_run_frac=0.005)
    fa = os.path.join(workdir, "dense.fa")
    synth.write_fasta(fa, g)
    oi = ol.Index.build(fa, 578_041)
    ufi = os.path.join(workdir, "dense.ufi")
    oi.save(ufi)
    t0 = oi.blob().reshape(-1, 5)[:, 0]
    assert int((t0 == 253).sum()) >= 1000 and int((t0 == 125).sum()) >= 1000, "fixture lost its long links"
    idx = api.Index.open(ufi).upload(0)
    return {"genome": g, "ufi": ufi, "oracle_index": oi, "index": idx, "mapper": api.Mapper(idx, device=0, method=6)}


def _long_link_hops(oi, slots, tallies, limit):
    """Walk the chains of the first `limit` k-mers whose head is "mine" the way GetRow_Blob does and count the hops that
    go through a long link."""
    blob = oi.blob().reshape(-1, 5)
    N, max_ix = oi.slot_count, oi.max_ix
    hops = walked = 0
    for s, t in zip(slots.tolist(), tallies.tolist()):
        if s == 0xFFFFFFFFFFFFFFFF or not (t & 128) or t in (254, 255):
            continue
        walked += 1
        k = 0
        while True:
            k += 1
            t = int(blob[s, 0])
            if k == max_ix or t == 127 or t in (254, 255):
                break
            if t in (253, 125):
                pos = int(blob[s, 1:5].view("<u4")[0])
                s = (s + (pos & 0xFFFF) + (pos >> 16)) % N
                hops += 1
            else:
                s = (s + (t & 127)) % N
        if walked >= limit:
            break
    return hops, walked


@pytest.mark.parametrize("read_len,sub,indel,n", [(150, 0.01, 0.001, 3000), (250, 0.04, 0.01, 800)])
def test_map_se_on_dense_index_walks_long_links(dense_case, read_len, sub, indel, n):
    """Single-end parity on the dense index: the device's chain walk (walk_all) goes through TALLY_LONG_* slots."""
    from urmap_amd import synth
    from conftest import reads_to_arrays
    oi = dense_case["oracle_index"]
    reads = synth.make_reads(300 + read_len, dense_case["genome"], n, read_len=read_len, sub=sub, ins=indel / 2, dele=indel / 2,
                             random_frac=0.02)
    bases, offs = reads_to_arrays(reads)
    slots, tallies, _ = dense_case["mapper"].seed_probe(bases, offs)
    assert int((tallies == 253).sum()) > 200, "no read k-mer lands on a long-link head"
    hops, walked = _long_link_hops(oi, slots, tallies, 20000)
    assert hops > 300, (hops, walked)
    ores, opaths, _ = oi.map_se(bases, offs, threads=4)
    gres, gops = dense_case["mapper"].map_se(bases, offs)
    compare_results(gres, gops, ores, opaths)
    assert (ores["dbpos"] != 0xFFFFFFFF).mean() > 0.9


def test_pe_on_dense_index_walks_long_links(dense_case, tmp_path):
    """Paired-end twin (SearchPE_Pending's chain walks, kernels_pe.hip) on the dense index: SAM == the oracle's."""
    import os
    from urmap_amd import synth
    r1, r2 = synth.make_pairs(91, dense_case["genome"], 2500, read_len=150, sub1=0.01, sub2=0.03, ins=0.001, dele=0.001)
    f1, f2 = os.path.join(tmp_path, "r1.fq"), os.path.join(tmp_path, "r2.fq")
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    osam = os.path.join(tmp_path, "o.sam")
    dense_case["oracle_index"].map_file_pe(f1, f2, osam, threads=4)
    got = _map_pe_sam(dense_case["ufi"], f1, f2)
    want = open(osam, "rb").read()
    if got != want:
        g, w = got.split(b"\n"), want.split(b"\n")
        bad = [i for i in range(min(len(g), len(w))) if g[i] != w[i]]
        raise AssertionError(f"{len(bad)} differing records of {len(w)}, first: {g[bad[0]][:200]!r} vs {w[bad[0]][:200]!r}")


@pytest.mark.parametrize("lo,hi", [(24, 128), (24, 192), (24, 256), (24, 320), (300, 512), (500, 1024)])
def test_map_se_mixed_lengths_in_one_batch(small_case, gpu, lo, hi):
    """A batch is run by the kernel instance of its longest read: reads of every length from W up to the class limit in
    one batch, each class limit in turn."""
    from urmap_amd import synth
    from conftest import reads_to_arrays
    rng = np.random.default_rng(hi)
    reads = []
    for i, L in enumerate(rng.integers(lo, hi + 1, size=400).tolist() + [lo, hi, hi, lo + 1, hi - 1]):
        reads += synth.make_reads(7000 + 13 * i + hi, small_case["genome"], 1, read_len=int(L), sub=0.02, ins=0.002, dele=0.002)
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = small_case["oracle_index"].map_se(bases, offs, threads=4)
    gres, gops = gpu["mapper"].map_se(bases, offs)
    compare_results(gres, gops, ores, opaths)


def test_results_do_not_depend_on_scheduling(small_case, gpu):
    """Reads are handed to wavefronts by a ticket counter, so which wave maps which read changes from launch to launch;
    nothing else may.  The same batch three times, then once in reversed read order: identical results per read
    (paths compared after expansion, their offsets in the arena differ)."""
    from urmap_amd import synth
    from conftest import reads_to_arrays
    reads = synth.make_reads(4242, small_case["genome"], 4000, read_len=150, sub=0.02, ins=0.002, dele=0.002,
                             random_frac=0.03)
    bases, offs = reads_to_arrays(reads)
    fields = [f for f in gpu["mapper"].map_se(bases, offs)[0].dtype.names if f != "path_off"]

    def run(b, o):
        res, ops = gpu["mapper"].map_se(b, o)
        paths = [tuple(int(x) for x in ops[int(r["path_off"]):int(r["path_off"]) + int(r["path_nops"])]) for r in res]
        return [tuple(r[f] for f in fields) for r in res], paths
    first = run(bases, offs)
    for _ in range(2):
        assert run(bases, offs) == first
    rbases, roffs = reads_to_arrays(reads[::-1])
    rres, rpaths = run(rbases, roffs)
    assert (rres[::-1], rpaths[::-1]) == first


def test_bad_lengths_are_flagged(small_case, gpu):
    """Reads shorter than W (the reference underflows there) are reported, not silently mis-mapped; a read longer than the
    fast kernels' 1024 bases goes to the general kernel and is mapped (tests/test_gpu_slow.py compares those with the oracle)."""
    from urmap_amd import api
    seqs = [np.frombuffer(b"ACGTACGTACGTACGT", np.uint8), np.frombuffer(b"ACGT" * 300, np.uint8)]
    offs = np.array([0, 16, 1216], dtype=np.uint64)
    with pytest.raises(api.UrmapxError) as e:
        gpu["mapper"].map_se(np.concatenate(seqs), offs)
    assert e.value.code == api.E_UNSUPPORTED
    res, _ = gpu["mapper"].map_se(np.concatenate(seqs), offs, allow_unsupported=True)
    assert res["status"][0] == 0x10 and res["status"][1] == 0


@pytest.mark.parametrize("name", ["se150", "se250", "se_short"])
def test_cli_map_reproduces_reference_sam(tmp_path, name):
    """`urmap -map FQ -ufi UFI -samout SAM` (map.cpp:27-67) of this build vs the SAM the reference binary wrote
    for the same inputs (tests/golden): identical records, @PG excluded."""
    import gzip
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    exe = os.path.join(root, "urmap_amd", "urmap")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    ufi = os.path.join(tmp_path, "g.ufi")
    with gzip.open(os.path.join(gold, "g.ufi.gz"), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    out = os.path.join(tmp_path, "out.sam")
    r = subprocess.run([exe, "-map", os.path.join(gold, name + ".fq"), "-ufi", ufi, "-samout", out, "-batch", "128"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = [l for l in open(out, "rb").read().split(b"\n") if l and not l.startswith(b"@PG")]
    want = [l for l in open(os.path.join(gold, name + ".sam"), "rb").read().split(b"\n") if l]
    assert got == want
    assert any(l.startswith(b"@PG\tID:urmap") for l in open(out, "rb").read().split(b"\n"))
    if name == "se_short":  # "-" is standard input (OpenStdioFile, myutils.cpp:430-431), here a pipe
        out3 = os.path.join(tmp_path, "out_stdin.sam")
        with open(os.path.join(gold, name + ".fq"), "rb") as f:
            r = subprocess.run([exe, "-map", "-", "-ufi", ufi, "-samout", out3], input=f.read(), stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert [l for l in open(out3, "rb").read().split(b"\n") if l and not l.startswith(b"@PG")] == want
    if name == "se150":  # the same reads gzip-compressed, CRLF line ends, several host threads and one
        fqz = os.path.join(tmp_path, "r.fq.gz")
        with open(os.path.join(gold, name + ".fq"), "rb") as f, gzip.open(fqz, "wb") as z:
            z.write(f.read().replace(b"\n", b"\r\n"))
        for threads in ("1", "7"):
            out2 = os.path.join(tmp_path, f"out_{threads}.sam")
            r = subprocess.run([exe, "-map", fqz, "-ufi", ufi, "-samout", out2, "-batch", "100", "-threads", threads],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            assert [l for l in open(out2, "rb").read().split(b"\n") if l and not l.startswith(b"@PG")] == want


HITSTATS_CASES = [("se150", "map", "se150", "g", []), ("se250", "map", "se250", "g", []),
                  ("se_short", "map", "se_short", "g", []), ("se150_minq3", "map", "se150", "g", ["-minq", "3"]),
                  ("pe150", "map2", "pe150", "g", []), ("pe100_noisy", "map2", "pe100_noisy", "g", []),
                  ("pe100_noisy_minq3", "map2", "pe100_noisy", "g", ["-minq", "3"]),
                  ("pe120_rep_minq25", "map2", "pe120_rep", "r", ["-minq", "25"])]


@pytest.mark.parametrize("key,mode,name,g,extra", HITSTATS_CASES)
def test_cli_hitstats_report_equals_reference(tmp_path, key, mode, name, g, extra):
    """End-of-run report on stderr (State1::HitStats, state1.cpp:593-632; counters output1.cpp:20-30, output2.cpp:14-15):
    the Reads / Mapped Q>= / Mapped Q< / Unmapped lines equal the reference binary's (tests/golden/hitstats.json), incl.
    -minq being read by -map2 only (map2.cpp:76) and the "not used" warning under -map."""
    import gzip
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    exe = os.path.join(root, "urmap_amd", "urmap")
    want = json.load(open(os.path.join(gold, "hitstats.json")))[key]
    ufi = os.path.join(tmp_path, g + ".ufi")
    with gzip.open(os.path.join(gold, g + ".ufi.gz"), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    if mode == "map":
        args = ["-map", os.path.join(gold, name + ".fq")]
    else:
        args = ["-map2", os.path.join(gold, name + "_1.fq"), "-reverse", os.path.join(gold, name + "_2.fq")]
    r = subprocess.run([exe] + args + ["-ufi", ufi, "-samout", os.path.join(tmp_path, "o.sam"), "-batch", "96"] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    keep = ("  Reads (", "  Mapped Q>=", "  Mapped Q< ", "  Unmapped (", "WARNING: Option -minq")
    got = [ln for ln in r.stderr.decode().split("\n") if any(k in ln for k in keep)]
    assert got == want
    rq = subprocess.run([exe] + args + ["-ufi", ufi, "-samout", os.path.join(tmp_path, "q.sam"), "-quiet"],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert rq.returncode == 0 and b"Mapped" not in rq.stderr


@pytest.mark.parametrize("short,digit", [(2, "2"), (1, "1")])
def test_cli_map2_unequal_files_die_like_the_reference(tmp_path, short, digit):
    """map2.cpp:27-33: one record from each file per pair; when one file ends first: Die("Premature end of file in
    FASTQ<n>"), n = the file that ended.  Also a malformed record in the second file (read by its own thread)."""
    import gzip
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    exe = os.path.join(root, "urmap_amd", "urmap")
    ufi = os.path.join(tmp_path, "g.ufi")
    with gzip.open(os.path.join(gold, "g.ufi.gz"), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    f1, f2 = os.path.join(tmp_path, "a_1.fq"), os.path.join(tmp_path, "a_2.fq")
    for k, dst in ((1, f1), (2, f2)):
        lines = open(os.path.join(gold, f"pe150_{k}.fq"), "rb").read().split(b"\n")
        keep = 4 * 250 if k == short else 4 * 300
        open(dst, "wb").write(b"\n".join(lines[:keep]) + b"\n")
    r = subprocess.run([exe, "-map2", f1, "-reverse", f2, "-ufi", ufi, "-samout", os.path.join(tmp_path, "o.sam"), "-batch", "64"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 1 and f"Premature end of file in FASTQ{digit}".encode() in r.stderr, r.stderr.decode()[-500:]
    if short == 2:
        lines = open(os.path.join(gold, "pe150_2.fq"), "rb").read().split(b"\n")
        lines[4 * 100 + 1] = lines[4 * 100 + 1][:50] + b"*" + lines[4 * 100 + 1][51:]
        open(f2, "wb").write(b"\n".join(lines))
        open(f1, "wb").write(open(os.path.join(gold, "pe150_1.fq"), "rb").read())
        r = subprocess.run([exe, "-map2", f1, "-reverse", f2, "-ufi", ufi, "-samout", os.path.join(tmp_path, "o.sam"), "-batch", "64"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 1 and b"Invalid sequence letter '*' in FASTQ, line 402 file" in r.stderr, r.stderr.decode()[-500:]


def test_cli_errors_exit_1(tmp_path):
    """Die(): message on stderr, exit status 1 (myutils.cpp:915)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "urmap_amd", "urmap")
    r = subprocess.run([exe, "-map", "/nonexistent.fq", "-ufi", "/nonexistent.ufi", "-samout", os.path.join(tmp_path, "x.sam")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and b"Fatal error" in r.stderr


def test_mapq_formula_exhaustive(gpu, small_case):
    """CalcMAPQ6 (search1m6.cpp:9-33) is double arithmetic truncated to unsigned; the device evaluates the same
    expression.  Perfect, 1-mismatch and 2-hit reads of every length class are covered by the mapping tests; here
    the mapped MAPQs of a mixed set are compared once more as a histogram against the oracle's."""
    from urmap_amd import synth
    from conftest import reads_to_arrays
    reads = synth.make_reads(77, small_case["genome"], 2000, read_len=120, sub=0.03, ins=0.003, dele=0.003)
    bases, offs = reads_to_arrays(reads)
    ores, _, _ = small_case["oracle_index"].map_se(bases, offs, threads=4)
    gres, _ = gpu["mapper"].map_se(bases, offs)
    assert (np.bincount(gres["mapq"], minlength=41) == np.bincount(ores["mapq"].astype(np.int64), minlength=41)).all()
    assert (gres["mapq"] == ores["mapq"]).all()


def test_veryfast_method7_matches_oracle(small_case, tmp_path):
    """`-veryfast`: State1::SetMethod(7) constants (state1.cpp:166-179: mismatch -4, gaps -6/-2, x-drop 12, band
    radius 8, three-phase exits) on a MaxIx = 3 index (ufindexio.cpp:133-136)."""
    import os
    import oracle_lib as ol
    from urmap_amd import api, synth
    from conftest import reads_to_arrays
    ufi = os.path.join(tmp_path, "vf.ufi")
    api.make_ufi(small_case["fasta"], ufi, 524309, max_ix=3)
    oi = ol.Index.build(small_case["fasta"], 524309, max_ix=3)
    oi.save(os.path.join(tmp_path, "vf_o.ufi"))
    assert open(ufi, "rb").read() == open(os.path.join(tmp_path, "vf_o.ufi"), "rb").read()
    reads = synth.make_reads(4242, small_case["genome"], 2500, read_len=150, sub=0.02, ins=0.002, dele=0.002, random_frac=0.03)
    reads = mutate_edge_reads(reads, 7)
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = oi.map_se(bases, offs, method=7, threads=4)
    idx = api.Index.open(ufi).upload(0)
    m = api.Mapper(idx, device=0, method=7)
    gres, gops = m.map_se(bases, offs)
    compare_results(gres, gops, ores, opaths)


def _map_pe_sam(ufi, fq1, fq2, veryfast=False):
    from urmap_amd import api
    idx = api.Index.open(ufi).upload(0)
    m = api.Mapper(idx, device=0, method=6)
    if veryfast:
        m.set_pe_veryfast(True)
    labels, bases, offs, quals = api.interleave_pairs(api.read_fastq_arrays(fq1), api.read_fastq_arrays(fq2))
    res, ops = m.map_pe(bases, offs)
    return idx.sam_header_sq() + idx.sam_pe(res, ops, labels, bases, offs, quals)


@pytest.mark.parametrize("name", ["pe150", "pe100_noisy"])
def test_pe_reproduces_reference_golden_sam(tmp_path, name):
    """urmap -map2 (State2::Search4 ... SetSAM2): both records of every pair identical to the reference's SAM."""
    import gzip
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    ufi = os.path.join(tmp_path, "g.ufi")
    with gzip.open(os.path.join(gold, "g.ufi.gz"), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    got = _map_pe_sam(ufi, os.path.join(gold, name + "_1.fq"), os.path.join(gold, name + "_2.fq"))
    want = open(os.path.join(gold, name + ".sam"), "rb").read()
    if got != want:
        g, w = got.split(b"\n"), want.split(b"\n")
        bad = [i for i in range(min(len(g), len(w))) if g[i] != w[i]]
        raise AssertionError(f"{len(bad)} differing records, first: {g[bad[0]][:160]!r} vs {w[bad[0]][:160]!r}")


@pytest.mark.parametrize("rl,s1,s2,indel,n", [(150, 0.01, 0.02, 0.001, 3000), (120, 0.04, 0.08, 0.01, 2000),
                                                (250, 0.02, 0.04, 0.005, 800), (270, 0.02, 0.04, 0.005, 500)])  # 250 / 270: the 256- and 320-base kernel classes (pairs: <= 279 bp, byte QPos)
def test_pe_matches_oracle(small_case, tmp_path, rl, s1, s2, indel, n):
    """Fresh pairs on the 300 kbp genome (incl. one-mate-random pairs that go through ScanPair): SAM of the device
    path == SAM of the oracle's Search4 restatement."""
    import os
    from urmap_amd import synth
    r1, r2 = synth.make_pairs(500 + rl, small_case["genome"], n, read_len=rl, sub1=s1, sub2=s2, ins=indel, dele=indel)
    rng = np.random.default_rng(rl)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for k in range(3, n, 29):
        lab, s, q = r2[k]
        r2[k] = (lab, acgt[rng.integers(0, 4, size=len(s))], q)
    for k in range(5, n, 31):
        lab, s, q = r1[k]
        s = s.copy(); s[int(rng.integers(0, len(s)))] = ord("N"); r1[k] = (lab, s, q)
    f1, f2 = os.path.join(tmp_path, "r1.fq"), os.path.join(tmp_path, "r2.fq")
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    osam = os.path.join(tmp_path, "o.sam")
    small_case["oracle_index"].map_file_pe(f1, f2, osam, threads=4)
    got = _map_pe_sam(small_case["ufi"], f1, f2)
    want = open(osam, "rb").read()
    if got != want:
        g, w = got.split(b"\n"), want.split(b"\n")
        bad = [i for i in range(min(len(g), len(w))) if g[i] != w[i]]
        raise AssertionError(f"{len(bad)} differing records of {len(w)}, first: {g[bad[0]][:200]!r} vs {w[bad[0]][:200]!r}")


@pytest.mark.gpu
@pytest.mark.parametrize("w,rl", [(16, 150), (16, 192), (21, 190), (30, 150)])
def test_pairs_on_an_index_with_another_word_length(small_case, tmp_path, w, rl):
    """-make_ufi -wordlength W: pairs against indexes whose words are not 24 letters.  With W = 16 a 192-base mate has 177 k-mer starts -- more than the
    first pass's pending lists hold since round 6 (kernels_pe.hip: QMAX - 16) -- and the pair must come out of the general pair kernel with the oracle's
    records, not out of an overrun list."""
    import os
    import oracle_lib as ol
    from urmap_amd import synth
    idx = ol.Index.build(small_case["fasta"], 524309, word_length=w)
    ufi = os.path.join(tmp_path, f"w{w}.ufi")
    idx.save(ufi)
    r1, r2 = synth.make_pairs(900 + w + rl, small_case["genome"], 600, read_len=rl, sub1=0.01, sub2=0.02, ins=0.001, dele=0.001)
    f1, f2 = os.path.join(tmp_path, "r1.fq"), os.path.join(tmp_path, "r2.fq")
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    osam = os.path.join(tmp_path, "o.sam")
    idx.map_file_pe(f1, f2, osam, threads=4)
    got = _map_pe_sam(ufi, f1, f2)
    want = open(osam, "rb").read()
    assert got == want
    assert want.count(b"\t99\t") + want.count(b"\t83\t") > 100  # proper pairs were found


@pytest.mark.parametrize("name", ["pe150", "pe100_noisy"])
def test_cli_map2_reproduces_reference_sam(tmp_path, name):
    """`urmap -map2 R1 -reverse R2 -ufi UFI -samout SAM` (map2.cpp:39-90) vs the reference's golden SAM."""
    import gzip
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    exe = os.path.join(root, "urmap_amd", "urmap")
    ufi = os.path.join(tmp_path, "g.ufi")
    with gzip.open(os.path.join(gold, "g.ufi.gz"), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    out = os.path.join(tmp_path, "out.sam")
    r = subprocess.run([exe, "-map2", os.path.join(gold, name + "_1.fq"), "-reverse", os.path.join(gold, name + "_2.fq"),
                        "-ufi", ufi, "-samout", out, "-batch", "100"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = [l for l in open(out, "rb").read().split(b"\n") if l and not l.startswith(b"@PG")]
    want = [l for l in open(os.path.join(gold, name + ".sam"), "rb").read().split(b"\n") if l]
    assert got == want


def test_pe_veryfast_search5_matches_oracle(small_case, tmp_path):
    """`-map2 -veryfast`: State2::Search5 (search2m5.cpp:9-156), band radius 4 (map2.cpp:17-21)."""
    import os
    from urmap_amd import synth
    n = 2500
    r1, r2 = synth.make_pairs(909, small_case["genome"], n, read_len=150, sub1=0.02, sub2=0.05, ins=0.004, dele=0.004)
    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for k in range(3, n, 23):
        lab, s, q = r1[k]
        r1[k] = (lab, acgt[rng.integers(0, 4, size=len(s))], q)
    f1, f2 = os.path.join(tmp_path, "r1.fq"), os.path.join(tmp_path, "r2.fq")
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    osam = os.path.join(tmp_path, "o.sam")
    small_case["oracle_index"].map_file_pe(f1, f2, osam, threads=4, veryfast=True)
    got = _map_pe_sam(small_case["ufi"], f1, f2, veryfast=True)
    want = open(osam, "rb").read()
    assert got == want


def test_gather_microbench_reports_a_rate(gpu):
    """The roofline denominator bench.py reports (random slot reads over the resident table) is measurable."""
    rate = gpu["mapper"].gather_microbench(1 << 22)
    assert rate > 1e8  # slot reads per second; a few 1e10 on MI355X


@pytest.mark.parametrize("name,ufi_gz,with_sam", [("pe150", "g.ufi.gz", True), ("pe100_noisy", "g.ufi.gz", True),
                                                  ("pe120_rep", "r.ufi.gz", True), ("pe120_rep", "r.ufi.gz", False)])
def test_cli_map2_tabbedout_reproduces_reference(tmp_path, name, ufi_gz, with_sam):
    """`urmap -map2 ... [-samout SAM] -tabbedout TAB`: State2::OutputTab2's line per pair (outputtab2.cpp:85-120) equal
    to the reference's file byte for byte, incl. second pairs and the TL/Score info strings (pe120_rep)."""
    import gzip
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    exe = os.path.join(root, "urmap_amd", "urmap")
    ufi = os.path.join(tmp_path, "x.ufi")
    with gzip.open(os.path.join(gold, ufi_gz), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    sam, tab = os.path.join(tmp_path, "out.sam"), os.path.join(tmp_path, "out.tab")
    cmd = [exe, "-map2", os.path.join(gold, name + "_1.fq"), "-reverse", os.path.join(gold, name + "_2.fq"), "-ufi", ufi,
           "-tabbedout", tab, "-batch", "128"] + (["-samout", sam] if with_sam else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    want = open(os.path.join(gold, name + (".tab" if with_sam else "_nosam.tab")), "rb").read()
    got = open(tab, "rb").read()
    if got != want:
        g, w = got.split(b"\n"), want.split(b"\n")
        bad = [i for i in range(min(len(g), len(w))) if g[i] != w[i]]
        raise AssertionError(f"{len(bad)} differing lines of {len(w)}, first: {g[bad[0]]!r} vs {w[bad[0]]!r}")
    if with_sam:
        gots = [l for l in open(sam, "rb").read().split(b"\n") if l and not l.startswith(b"@PG")]
        assert gots == [l for l in open(os.path.join(gold, name + ".sam"), "rb").read().split(b"\n") if l]


def test_pair_info_through_the_library(tmp_path):
    """urmapx_ctx_set_pair_info / urmapx_ctx_get_pair_info / urmapx_tab_pe through the ctypes binding."""
    import gzip
    import os
    from urmap_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    ufi = os.path.join(tmp_path, "r.ufi")
    with gzip.open(os.path.join(gold, "r.ufi.gz"), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    idx = api.Index.open(ufi).upload(0)
    m = api.Mapper(idx, device=0, method=6)
    m.set_pair_info(True)
    labels, bases, offs, quals = api.interleave_pairs(api.read_fastq_arrays(os.path.join(gold, "pe120_rep_1.fq")),
                                                      api.read_fastq_arrays(os.path.join(gold, "pe120_rep_2.fq")))
    res, ops = m.map_pe(bases, offs)
    info = m.pair_info(len(res) // 2)
    assert (info["second_db"][:, 0] != 0xFFFFFFFF).sum() > 20  # this fixture has second pairs
    assert idx.tab_pe(res, info, labels, offs, sam_on=True) == open(os.path.join(gold, "pe120_rep.tab"), "rb").read()
    assert idx.tab_pe(res, info, labels, offs, sam_on=False) == open(os.path.join(gold, "pe120_rep_nosam.tab"), "rb").read()


def test_hsp_overflow_list_matches_oracle(tmp_path):
    """Reads in a high-copy repeat family collect more HSPs than fit LDS; the rest go to the block's global scratch
    (the reference's list is unbounded, state1.cpp:193-228).  With the LDS share lowered to 64 by the test aid, reads
    with up to ~180 HSPs run through that path: SAM must still equal the oracle's."""
    import os
    import subprocess
    import oracle_lib as ol
    from urmap_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "urmap_amd", "urmap")
    g = synth.make_genome(9, [600_000], repeat_frac=0.85, n_families=2, max_div=0.12, n_run_frac=0.0)
    fa, ufi, fq = (os.path.join(tmp_path, x) for x in ("g.fa", "g.ufi", "r.fq"))
    synth.write_fasta(fa, g)
    idx = ol.Index.build(fa, 1_000_003)
    idx.save(ufi)
    reads = synth.make_reads(3, g, 3000, read_len=150, sub=0.02)
    synth.write_fastq(fq, reads)
    b = np.concatenate([r[1] for r in reads])
    o = np.zeros(len(reads) + 1, dtype=np.uint64)
    o[1:] = np.cumsum([len(r[1]) for r in reads])
    res, _, _ = idx.map_se(b, o, threads=4)
    assert (res["hsp_count"] > 64).sum() > 50, "fixture no longer exercises the overflow list"
    assert (res["hit_count"] > 16).sum() > 20, "fixture no longer exercises the long hit list (test cap 16)"
    osam = os.path.join(tmp_path, "o.sam")
    idx.map_file_se(fq, osam, threads=4)
    for cap in ("64", None):
        env = dict(os.environ)
        if cap:
            env["URMAPX_TEST_HSP_LDS_CAP"] = cap
        out = os.path.join(tmp_path, f"gpu_{cap}.sam")
        r = subprocess.run([exe, "-map", fq, "-ufi", ufi, "-samout", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=300, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert ol.sam_records(out) == ol.sam_records(osam), f"LDS cap {cap}"


def test_pe_hsp_overflow_list_matches_oracle(tmp_path):
    """Paired-end twin of test_hsp_overflow_list_matches_oracle: mates in a high-copy repeat family, LDS share of the
    HSP lists lowered to 64, pairs re-mapped by the second-pass kernel with the lists continued in global scratch."""
    import os
    import subprocess
    import oracle_lib as ol
    from urmap_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "urmap_amd", "urmap")
    g = synth.make_genome(9, [600_000], repeat_frac=0.85, n_families=2, max_div=0.12, n_run_frac=0.0)
    fa, ufi, f1, f2 = (os.path.join(tmp_path, x) for x in ("g.fa", "g.ufi", "r1.fq", "r2.fq"))
    synth.write_fasta(fa, g)
    idx = ol.Index.build(fa, 1_000_003)
    idx.save(ufi)
    r1, r2 = synth.make_pairs(4, g, 1500, read_len=150, sub1=0.02, sub2=0.03)
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    b = np.concatenate([x[1] for pair in zip(r1, r2) for x in pair])
    o = np.zeros(2 * len(r1) + 1, dtype=np.uint64)
    o[1:] = np.cumsum([len(x[1]) for pair in zip(r1, r2) for x in pair])
    res, _, _ = idx.map_pe(b, o, threads=4)
    assert (res["hsp_count"] > 64).sum() > 20, "fixture no longer exercises the overflow list"
    assert (res["hit_count"] > 16).sum() > 10, "fixture no longer exercises the long hit list (test cap 16)"
    osam = os.path.join(tmp_path, "o.sam")
    idx.map_file_pe(f1, f2, osam, threads=4)
    for cap in ("64", None):
        env = dict(os.environ)
        if cap:
            env["URMAPX_TEST_HSP_LDS_CAP"] = cap
        out = os.path.join(tmp_path, f"gpu_{cap}.sam")
        r = subprocess.run([exe, "-map2", f1, "-reverse", f2, "-ufi", ufi, "-samout", out], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=300, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert ol.sam_records(out) == ol.sam_records(osam), f"LDS cap {cap}"


@pytest.mark.parametrize("name", ["g", "r"])
def test_make_ufi_gpu_reproduces_reference_index(tmp_path, name):
    """-make_ufi with the counting passes, head slots and overflow list made on the GPU (make_ufi_gpu.hip) and only the
    order-dependent inserts on the host: the .ufi the reference binary wrote (tests/golden), byte for byte -- through the
    library and through the command line (default slot count = the reference's GetPrime ladder)."""
    import gzip
    import os
    import subprocess
    import oracle_lib as ol
    from urmap_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    want = gzip.open(os.path.join(gold, name + ".ufi.gz"), "rb").read()
    ref = os.path.join(tmp_path, "ref.ufi")
    open(ref, "wb").write(want)
    w, maxix, _, slots = ol.ufi_header(ref)
    out = os.path.join(tmp_path, "gpu.ufi")
    api.make_ufi_gpu(0, os.path.join(gold, name + ".fa"), out, slots, word_length=w, max_ix=maxix)
    assert open(out, "rb").read() == want
    if name == "g":
        out2 = os.path.join(tmp_path, "cli.ufi")
        r = subprocess.run([os.path.join(root, "urmap_amd", "urmap"), "-make_ufi", os.path.join(gold, "g.fa"), "-output", out2],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0 and b"building on the host" not in r.stderr, r.stderr.decode()[-1000:]
        assert open(out2, "rb").read() == want


@pytest.mark.parametrize("load,maxix", [(0.6, 32), (0.95, 32), (1.0, 32), (0.8, 3)])
def test_make_ufi_gpu_matches_oracle_on_dense_tables(tmp_path, load, maxix):
    """Slot tables at load factors up to 1.0 (long links, truncated chains) and with -veryfast's MaxIx 3, lower-case and N
    runs in the FASTA: GPU-assisted build == the oracle's MakeIndex restatement (itself checked against the reference)."""
    import os
    import oracle_lib as ol
    from urmap_amd import api, synth
    g = synth.make_genome(int(load * 100) + maxix, [260000, 90000, 5000], repeat_frac=0.3, n_families=5, n_run_frac=0.01)
    fa = os.path.join(tmp_path, "g.fa")
    synth.write_fasta(fa, g, lowercase_frac=0.1)
    slots = int(355000 / load) | 1
    oi = ol.Index.build(fa, slots, max_ix=maxix)
    want = os.path.join(tmp_path, "o.ufi")
    oi.save(want)
    out = os.path.join(tmp_path, "gpu.ufi")
    api.make_ufi_gpu(0, fa, out, slots, max_ix=maxix)
    assert open(out, "rb").read() == open(want, "rb").read()
    # the array form (bench.py passes the sequence store resident on the device instead of a host array)
    blob = api.build_slots_gpu(0, slots, seqdata=oi.seqdata().copy(), max_ix=maxix)
    assert bytes(blob[:5 * slots]) == bytes(oi.blob())


def test_cli_long_single_end_reads(small_case, tmp_path):
    """600 and 1000 base reads through the command line (the 1024-base kernel class, FASTQ in, SAM with CIGAR out): the
    oracle's SAM, record for record.  The reference's own scratch holds reads up to ~30 kb (state1.h:113); this build's
    single-end device domain ends at 1024."""
    import os
    import subprocess
    import oracle_lib as ol
    from urmap_amd import synth
    reads = synth.make_reads(61, small_case["genome"], 150, read_len=600, sub=0.02, ins=0.002, dele=0.002) + \
        synth.make_reads(62, small_case["genome"], 100, read_len=1000, sub=0.01, ins=0.001, dele=0.002)
    fq = os.path.join(tmp_path, "long.fq")
    synth.write_fastq(fq, reads)
    osam = os.path.join(tmp_path, "o.sam")
    small_case["oracle_index"].map_file_se(fq, osam, threads=4)
    out = os.path.join(tmp_path, "gpu.sam")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "urmap_amd", "urmap")
    r = subprocess.run([exe, "-map", fq, "-ufi", small_case["ufi"], "-samout", out, "-batch", "64"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert ol.sam_records(out) == ol.sam_records(osam)
    assert sum(1 for l in ol.sam_records(out) if not l.startswith(b"@") and l.split(b"\t")[2] != b"*") > 200


@pytest.mark.gpu
@pytest.mark.parametrize("quiet", [False, True])
def test_cli_log_file_holds_rps_and_the_report(tmp_path, quiet):
    """-log FILE (State1::HitStats, state1.cpp:593-632): `@rps=` goes to the log only (Log), the report to the terminal
    and the log (ProgressLog); -quiet silences the terminal, not the log.  -trunclabels is accepted and changes nothing
    for -map (SetSAM cuts the read label itself)."""
    import gzip
    import json
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    ufi = os.path.join(tmp_path, "g.ufi")
    with gzip.open(os.path.join(gold, "g.ufi.gz"), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    out, log = os.path.join(tmp_path, "o.sam"), os.path.join(tmp_path, "run.log")
    r = subprocess.run([os.path.join(root, "urmap_amd", "urmap"), "-map", os.path.join(gold, "se150.fq"), "-ufi", ufi, "-samout", out,
                        "-log", log, "-trunclabels"] + (["-quiet"] if quiet else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    text = open(log).read()
    assert re.search(r"^@rps=\d+\.\d$", text, re.M), text
    assert "-map" in text.split("\n")[1] and "Started " in text and "Finished " in text
    want = json.load(open(os.path.join(gold, "hitstats.json")))["se150"]
    keep = ("  Reads (", "  Mapped Q>=", "  Mapped Q< ", "  Unmapped (")
    assert [ln for ln in text.split("\n") if any(k in ln for k in keep)] == want
    err = r.stderr.decode()
    assert ("Mapped Q>=" in err) == (not quiet) and "@rps=" not in err
    got = [l for l in open(out, "rb").read().split(b"\n") if l and not l.startswith(b"@PG")]
    assert got == [l for l in open(os.path.join(gold, "se150.sam"), "rb").read().split(b"\n") if l]


@pytest.fixture(scope="module")
def rescue_case(tmp_path_factory):
    """Pairs whose second mate lies in an exact 60-copy repeat (every k-mer of it is dropped by the index: more than MaxIx
    occurrences) next to a unique first mate: the only way to place the second mate is State2::ScanPair (state2.cpp:87-137)
    -> Scan / ScanSlots / ExtendScan (scan.cpp:14-39, scanslots.cpp:7-62, extendscan.cpp:51-187).  A third of the repeat
    mates have a substitution in each of the SCANK probe k-mers (query positions 0, 27, 54, 81), so ScanSlots finds nothing
    and the whole-read Viterbi rescue places them."""
    import os
    import oracle_lib as ol
    from urmap_amd import synth
    d = str(tmp_path_factory.mktemp("rescue"))
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: acgt[rng.integers(0, 4, n)]
    sub = lambda b: acgt[(int(np.where(acgt == b)[0][0]) + 1) % 4]
    rep = rnd(400)
    parts, starts = [], []
    for j in range(60):
        starts.append(sum(len(p) for p in parts))
        parts += [rnd(600), rep.copy(), rnd(300)]
    genome = [("chrR", np.concatenate(parts)), ("chrO", rnd(40000))]
    fa = os.path.join(d, "r.fa")
    synth.write_fasta(fa, genome)
    idx = ol.Index.build(fa, 524309)
    ufi = os.path.join(d, "r.ufi")
    idx.save(ufi)
    seq = genome[0][1]
    reads = []
    for j in range(60):
        s = starts[j]
        f = seq[s + 300:s + 450].copy()
        lo = s + 600 + 50 + (j % 5) * 10
        r = synth.revcomp(seq[lo:lo + 150])
        if j % 3 == 0:
            for p in (10, 35, 60, 90, 139, 114, 89, 59):  # every probe k-mer of either strand's coordinates
                r[p] = sub(r[p])
        else:
            for t in range(j % 4):
                r[120 + 7 * t] = sub(r[120 + 7 * t])
        if j % 2:
            f, r = r, f
        q = np.full(150, ord("I"), np.uint8)
        reads += [(f"p{j}/1", f, q), (f"p{j}/2", r, q)]
    return {"dir": d, "ufi": ufi, "oracle_index": idx, "reads": reads}


@pytest.mark.gpu
def test_pe_rescue_scan_produces_hits(rescue_case):
    """a21: the rescue path must PRODUCE hits on the device, not merely run: the oracle's counters say how many
    hits Scan added (through ExtendScan and through the whole-read Viterbi), and every mate's result equals the oracle's."""
    from conftest import reads_to_arrays
    from urmap_amd import api
    c = rescue_case
    bases, offs = reads_to_arrays(c["reads"])
    ores, opaths, cnt = c["oracle_index"].map_pe(bases, offs)
    assert cnt["n_scan"] >= 60 and cnt["n_extscan"] >= 100 and cnt["n_scan_vit"] >= 15 and cnt["n_scan_hits"] >= 55, cnt
    assert cnt["n_dpcells"] > 0  # the pair path's DP cells are counted too
    assert (ores["dbpos"] != 0xFFFFFFFF).sum() >= 115
    m = api.Mapper(api.Index.open(c["ufi"]).upload(0), device=0)
    g, gops = m.map_pe(bases, offs)
    assert (g["status"] == 0).all()
    for name in ("dbpos", "seq_index", "coord", "score", "second", "mapq"):
        assert (g[name].astype(np.int64) == ores[name].astype(np.int64)).all(), name
    mapped = ores["dbpos"] != 0xFFFFFFFF
    assert (g["plus"][mapped] == ores["plus"][mapped]).all()
    gapped = 0
    for i in np.nonzero(mapped)[0]:
        o = int(g["path_off"][i])
        assert api.decode_path(gops[o:o + int(g["path_nops"][i])]) == opaths[i], i
        gapped += int(g["path_nops"][i]) > 0


@pytest.mark.gpu
def test_chain_rows_on_and_off_give_the_oracle(dense_case, tmp_path, monkeypatch):
    """chain_rows.hip: GetRow_Blob's rows (ufindex.cpp:883-943) laid out beside the resident table; the search kernels look a
    row up instead of walking the chain.  On the dense index (thousands of long-link slots, truncated chains): the layout is
    built by default, both settings give the oracle's results for single reads and pairs."""
    from conftest import reads_to_arrays
    from urmap_amd import api, synth
    c = dense_case
    reads = synth.make_reads(4711, c["genome"], 2500, read_len=150, sub=0.01, ins=0.001, dele=0.001)
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = c["oracle_index"].map_se(bases, offs, threads=4)
    long_reads = synth.make_reads(4713, c["genome"], 800, read_len=250, sub=0.03, ins=0.004, dele=0.004)  # another read-length class of the kernels
    lb, lo = reads_to_arrays(long_reads)
    lres, _, _ = c["oracle_index"].map_se(lb, lo, threads=4)
    r1, r2 = synth.make_pairs(4712, c["genome"], 1200, read_len=150, sub1=0.01, sub2=0.02, ins=0.001, dele=0.001)
    pairs = [x for ab in zip(r1, r2) for x in ab]
    pb, po = reads_to_arrays(pairs)
    pres, ppaths, _ = c["oracle_index"].map_pe(pb, po, threads=4)
    sizes, sums = {}, {}
    # what a device with room for everything runs -- built in one go (round 6: slot16 is its own scratch) or in two steps through the per-slot
    # info entries (URMAPX_TWO_STEP_LAYOUT, the round-5 build) --, one with room for the rows only, one with room for neither
    for mode in ("slot16", "slot16_two_step", "rows", "walk"):
        off = mode == "walk"
        if mode == "slot16_two_step":
            monkeypatch.setenv("URMAPX_TWO_STEP_LAYOUT", "1")
        if mode == "rows":
            monkeypatch.delenv("URMAPX_TWO_STEP_LAYOUT")
            monkeypatch.setenv("URMAPX_NO_SLOT16", "1")  # search_se_kernel<.., ROWS 1>, the pair kernel's info-entry lookup
        if off:
            monkeypatch.delenv("URMAPX_NO_SLOT16")
            monkeypatch.setenv("URMAPX_NO_CHAIN_ROWS", "1")
        idx = api.Index.open(c["ufi"]).upload(0)
        assert (idx.chain_row_bytes() == 0) == off
        sizes[mode] = idx.chain_row_bytes()
        sums[mode] = idx.layout_checksum()
        m = api.Mapper(idx, device=0)
        g, gops = m.map_se(bases, offs)
        for name in ("dbpos", "seq_index", "coord", "score", "second", "mapq", "exit_phase", "hit_count"):
            assert (g[name].astype(np.int64) == ores[name].astype(np.int64)).all(), (mode, name)
        for i in np.nonzero(ores["dbpos"] != 0xFFFFFFFF)[0]:
            o = int(g["path_off"][i])
            assert api.decode_path(gops[o:o + int(g["path_nops"][i])]) == opaths[i]
        g, _ = m.map_se(lb, lo)
        for name in ("dbpos", "seq_index", "coord", "score", "second", "mapq", "exit_phase", "hit_count"):
            assert (g[name].astype(np.int64) == lres[name].astype(np.int64)).all(), (mode, "250", name)
        g, gops = m.map_pe(pb, po)
        for name in ("dbpos", "seq_index", "coord", "score", "second", "mapq"):
            assert (g[name].astype(np.int64) == pres[name].astype(np.int64)).all(), (mode, name)
        m.close()
        idx.close()
    assert sizes["slot16"] > sizes["rows"] > 0  # (the 16-byte table is there by default, and is what URMAPX_NO_SLOT16 leaves out)
    assert sizes["slot16"] == sizes["slot16_two_step"]  # the same layouts resident either way (slot16 + rows; the info entries are dropped)
    assert sums["slot16"] == sums["slot16_two_step"] and sums["slot16"][0] != 0 and sums["slot16"][1] != 0  # and the same BYTES in them
    assert sums["rows"][0] == 0 and sums["rows"][1] == sums["slot16"][1] and sums["walk"] == (0, 0)


@pytest.mark.gpu
def test_phase6_launches_path_arena_dense_and_round_times(dense_case):
    """Phase 6 as launches of its own (kernels.hip: dp_round_lists_kernel, dp_kernel, finalize_se_kernel): reads of a genome with
    repeat families park with many HSPs each.  The results are the oracle's; the path arena is dense (finalize_se_kernel hands the
    paths of the reads a block finishes to the arena many at a time: every gapped read's runs are there exactly once, nothing
    between them); the launches are timed one by one (urmapx_ctx_round_ms) and add up to the stage times."""
    from urmap_amd import synth
    from conftest import reads_to_arrays
    reads = synth.make_reads(4242, dense_case["genome"], 4000, read_len=250, sub=0.04, ins=0.005, dele=0.005, random_frac=0.02)
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = dense_case["oracle_index"].map_se(bases, offs, threads=4)
    m = dense_case["mapper"]
    gres, gops = m.map_se(bases, offs)
    compare_results(gres, gops, ores, opaths)
    st = m.dp_stats()
    assert st[0] > st[1] > 500 and st[2] > 0, st  # reads were parked with their HSPs as jobs, and the ordered replay looked at DPs
    nops = gres["path_nops"].astype(np.int64)
    have = nops > 0
    order = np.argsort(gres["path_off"][have])
    o, l = gres["path_off"][have][order].astype(np.int64), nops[have][order]
    assert (o[1:] >= o[:-1] + l[:-1]).all() and o[-1] + l[-1] <= len(gops)  # no two paths share a run
    # nothing in the arena but paths -- except what a read mapped again by the second pass left behind from its first
    assert int(nops.sum()) >= len(gops) - 96 * max(8, len(reads) // 100), (int(nops.sum()), len(gops))
    rounds = m.round_ms()
    stage = m.stage_ms()
    assert len(rounds) == 4 and all(d >= 0 and f >= 0 for d, f in rounds)  # 250-base reads: four rounds (kernels.h: DpBounds)
    assert m.dp_rounds() == "HSPs [0,2), [2,8), [8,32), [32,...) of a read"
    assert abs(sum(d for d, _ in rounds) - stage[1]) < 0.05 + 0.02 * stage[1]
    assert abs(sum(f for _, f in rounds) - stage[2]) < 0.05 + 0.02 * stage[2]


@pytest.mark.gpu
@pytest.mark.parametrize("bounds,n", [("0,2,16", 3), ("0,1,4,16", 4), ("0", 1), ("0,64", 2)])
def test_phase6_round_boundaries_do_not_change_results(dense_case, monkeypatch, bounds, n):
    """the rounds only decide WHEN a read's HSPs are aligned and which are gated before their DP; the ordered replay sees the same jobs"""
    from urmap_amd import synth
    from conftest import reads_to_arrays
    reads = synth.make_reads(4243, dense_case["genome"], 1500, read_len=250, sub=0.04, ins=0.005, dele=0.005)
    reads += synth.make_reads(4244, dense_case["genome"], 10, read_len=150, sub=0.02)  # (batch class stays the 256-base one)
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = dense_case["oracle_index"].map_se(bases, offs, threads=4)
    monkeypatch.setenv("URMAPX_DP_BOUNDS", bounds)
    m = dense_case["mapper"]
    gres, gops = m.map_se(bases, offs)
    compare_results(gres, gops, ores, opaths)
    assert len(m.round_ms()) == n


@pytest.mark.gpu
@pytest.mark.parametrize("lo,hi", [(60, 128), (130, 192), (200, 256)])
def test_short_flanks_with_a_gap_next_to_the_read_ends(small_case, gpu, lo, hi):
    """A one-base insertion or deletion 6..20 bases from an end of the read: the ungapped extension stops at it and AlignHSP's flank
    DP runs on a handful of rows, most of its lanes left of column 0 or right of column LB.  The row blocks address the window's
    bytes as base + immediate with a base below the window's start for those lanes; dp_kernel once depended on what the compiler
    had placed in front of the window in LDS (one 14-row flank in the whole suite met it)."""
    from urmap_amd import synth
    from conftest import reads_to_arrays
    rng = np.random.default_rng(1000 + hi)
    reads = []
    for si, (name, seq) in enumerate(small_case["genome"]):
        for _ in range(400):
            L = int(rng.integers(lo, hi + 1))
            p0 = int(rng.integers(0, len(seq) - L - 4))
            frag = seq[p0:p0 + L + 2].copy()
            if (frag == ord("N")).any():
                continue
            p = int(rng.integers(6, 21))
            if rng.random() < 0.5:
                p = L - p
            if rng.random() < 0.5:
                r = np.delete(frag, p)[:L]  # a base of the reference the read does not have
            else:
                r = np.insert(frag, p, synth.ACGT[(int(np.where(synth.ACGT == (frag[p] & 0xDF))[0][0]) + 1) % 4])[:L]
            if rng.random() < 0.3:  # and a substitution somewhere else
                q = int(rng.integers(0, L))
                r[q] = synth.ACGT[(int(np.where(synth.ACGT == (r[q] & 0xDF))[0][0]) + 2) % 4]
            if rng.random() < 0.5:
                r = synth.revcomp(r)
            reads.append((f"e{len(reads)}", np.ascontiguousarray(r), np.full(len(r), ord("I"), np.uint8)))
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = small_case["oracle_index"].map_se(bases, offs, threads=4)
    assert sum(("I" in p or "D" in p) for p in opaths) > len(reads) // 3  # the gaps are aligned, not clipped away
    gres, gops = gpu["mapper"].map_se(bases, offs)
    compare_results(gres, gops, ores, opaths)


@pytest.mark.gpu
@pytest.mark.parametrize("w,hi", [(24, 151), (24, 152), (30, 157), (30, 158), (16, 143), (16, 144)])
def test_two_chunk_instance_up_to_128_kmer_starts(small_case, tmp_path, monkeypatch, w, hi):
    """Round 6: a batch whose longest read has at most 128 k-mer starts (150 bases at W = 24) is mapped by search_se_kernel<3, .., KCH = 2>: slot entries,
    prefix array and chain groups for two chunks of 64 starts instead of three, the row store in LDS.  Either side of the limit for three word lengths
    (hi - W + 1 = 128: the two-chunk instance; 129: the three-chunk one), every read length from W up in the batch, N / lower case / IUPAC reads among them:
    the oracle's results, and the same results with the instance switched off (URMAPX_NO_K2)."""
    import os
    import oracle_lib as ol
    from urmap_amd import api, synth
    from conftest import reads_to_arrays
    if w == 24:
        oi, ufi = small_case["oracle_index"], small_case["ufi"]
    else:
        oi = ol.Index.build(small_case["fasta"], 524309, word_length=w)
        ufi = os.path.join(tmp_path, f"w{w}.ufi")
        oi.save(ufi)
    rng = np.random.default_rng(100 * w + hi)
    reads = []
    for i, L in enumerate(rng.integers(w, hi + 1, size=300).tolist() + [hi] * 300 + [w, hi - 1]):
        reads += synth.make_reads(9000 + 17 * i + hi, small_case["genome"], 1, read_len=int(L), sub=0.02, ins=0.002, dele=0.002)
    reads = mutate_edge_reads(reads, hi)
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = oi.map_se(bases, offs, threads=4)
    idx = api.Index.open(ufi).upload(0)
    m = api.Mapper(idx, device=0, method=6)
    gres, gops = m.map_se(bases, offs)
    compare_results(gres, gops, ores, opaths)
    monkeypatch.setenv("URMAPX_NO_K2", "1")
    gres3, gops3 = m.map_se(bases, offs)
    compare_results(gres3, gops3, ores, opaths)
    m.close()
    idx.close()


@pytest.mark.gpu
def test_a_genome_of_more_than_64_sequences(tmp_path):
    """PosToCoordL (ufindex.cpp:729-755) on the device: up to 64 sequences every lane tests one (round 6), more than that the binary search over the
    offsets runs -- 70 sequences of 4 kbp, single reads and pairs against the oracle (sequence index and coordinate are compared per read)."""
    import os
    import oracle_lib as ol
    from urmap_amd import api, synth
    from conftest import reads_to_arrays
    g = synth.make_genome(707, [4000] * 70, repeat_frac=0.2, n_families=6)
    fa = os.path.join(tmp_path, "many.fa")
    synth.write_fasta(fa, g)
    oi = ol.Index.build(fa, 524309)
    ufi = os.path.join(tmp_path, "many.ufi")
    oi.save(ufi)
    reads = synth.make_reads(708, g, 2000, read_len=150, sub=0.01, ins=0.001, dele=0.001, random_frac=0.02)
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = oi.map_se(bases, offs, threads=4)
    assert len(np.unique(ores["seq_index"][ores["dbpos"] != 0xFFFFFFFF])) > 64
    idx = api.Index.open(ufi).upload(0)
    m = api.Mapper(idx, device=0, method=6)
    gres, gops = m.map_se(bases, offs)
    compare_results(gres, gops, ores, opaths)
    m.close()
    idx.close()
    r1, r2 = synth.make_pairs(709, g, 600, read_len=150, sub1=0.01, sub2=0.02, ins=0.001, dele=0.001)
    f1, f2 = os.path.join(tmp_path, "r1.fq"), os.path.join(tmp_path, "r2.fq")
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    osam = os.path.join(tmp_path, "o.sam")
    oi.map_file_pe(f1, f2, osam, threads=4)
    assert _map_pe_sam(ufi, f1, f2) == open(osam, "rb").read()
