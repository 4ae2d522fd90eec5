"""The HOST side of the drop-in path under sanitizers.

`make -C urmap_amd/csrc san` compiles pipeline.cpp (urmapx_map_files: reader -> lanes -> writer threads, chunk and shard cutters,
the SAM writer), sam.cpp (FASTQ reader, SAM / tab text), pgzip.cpp (parallel gzip reader) and make_ufi.cpp (host index builder)
UNCHANGED with g++ -fsanitize=address,undefined and again with -fsanitize=thread, linked against san/stub_device.cpp instead of
the HIP translation units: a lane "maps" a chunk on the host with a pure function of the read.  So cmd_map / cmd_map2 as the
library runs them -- queues, page-locked buffer pool, offsets, hand-backs to the host reader, shards -- run here under ASan +
UBSan and under TSan, on well-formed input (the SAM must not depend on chunk size, lanes, 'devices', shards or the road a chunk
took) and on damaged input (an error code or a clean run, never a sanitizer report, never a signal).

The reference's own memory story is RCE_MALLOC, off (myutils.h:13); malformed FASTQ must Die with its messages
(fastqseqsource.cpp:44-105), which tests/test_fastq_cpu.py checks on the production library."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "urmap_amd", "csrc")
ASAN = os.path.join(CSRC, "build_san", "urmap_san_asan")
TSAN = os.path.join(CSRC, "build_san", "urmap_san_tsan")
SAN_EXIT = 99
ENV = dict(os.environ, ASAN_OPTIONS=f"exitcode={SAN_EXIT}:detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS=f"exitcode={SAN_EXIT}:print_stacktrace=1:halt_on_error=1",
           TSAN_OPTIONS=f"exitcode={SAN_EXIT}:halt_on_error=1", OMP_NUM_THREADS="4")


@pytest.fixture(scope="module")
def san():
    if not shutil.which("g++"):
        pytest.skip("no g++")
    r = subprocess.run(["make", "-s", "-j3", "-C", CSRC, "san"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return {"asan": ASAN, "tsan": TSAN}


def run(exe, args, env=None, stdin=None):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, env=dict(ENV, **(env or {})), stdin=stdin, timeout=600)
    out = r.stdout.decode("latin-1") + r.stderr.decode("latin-1")
    assert r.returncode != SAN_EXIT and r.returncode >= 0, f"sanitizer report or signal (rc {r.returncode}) from {args}:\n{out[-3000:]}"
    assert "Sanitizer" not in out and "runtime error" not in out, out[-3000:]
    return r.returncode, out


def make_reads(n, seed, fixed=0):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        L = fixed or int(rng.integers(25, 220))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        q = bytes(rng.integers(33, 74, L, dtype=np.uint8))
        recs.append((b"read%d extra words/1" % i, s, q))
    return recs


def fastq_text(recs, eol=b"\n"):
    return b"".join(b"@" + l + eol + s + eol + b"+" + eol + q + eol for l, s, q in recs)


def unmapped_sam(recs):
    """SetSAM_Unmapped (setsam.cpp:12-44) with output1.cpp:13's arguments: the label cut at white space, '/1' dropped, FLAG 4"""
    out = []
    for l, s, q in recs:
        name = l[:-2] if l.endswith((b"/1", b"/2")) else l
        name = name.split(b" ")[0].split(b"\t")[0]
        out.append(name + b"\t4\t*\t0\t0\t*\t*\t0\t0\t" + s + b"\t" + q)
    return out


def records(path):
    return [l for l in open(path, "rb").read().split(b"\n") if l and not l.startswith(b"@")]


@pytest.fixture(scope="module")
def reads(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    recs = make_reads(30000, 11)
    fq = os.path.join(d, "r.fq")
    open(fq, "wb").write(fastq_text(recs))
    return {"dir": str(d), "recs": recs, "fq": fq}


VARIANTS = [
    ("small_chunks", ["-batch", 300, "-streams", 3], {}),
    ("two_devices", ["-batch", 1000, "-gpus", 2, "-streams", 2], {}),
    ("one_lane", ["-batch", 5000, "-streams", 1, "-threads", 1], {}),
    ("sam_cap_refetch", ["-batch", 700], {"URX_STUB_FORCE_SAM_CAP": "2"}),
    ("no_deferred_copy", ["-batch", 700], {"URMAPX_NO_DEFERRED_COPY": "1"}),
    ("host_text", ["-batch", 900], {"URMAPX_HOST_TEXT": "1"}),
    ("no_lane_pool", ["-batch", 900], {"URMAPX_NO_LANE_POOL": "1"}),
    ("mmap_writer", ["-batch", 900], {"URMAPX_SAM_WRITE": "mmap"}),
    # no -batch: the library's own chunk sizes, ramped up at the start of the file and down at its end (pipeline.cpp, round 6), three lanes
    ("ramped_chunks", ["-streams", 3], {"URMAPX_TEST_CHUNK_READS": "1600"}),
    ("ramp_off", ["-streams", 2], {"URMAPX_TEST_CHUNK_READS": "1600", "URMAPX_NO_CHUNK_RAMP": "1"}),
]


@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_map_files_same_sam_whatever_the_road(san, reads, which):
    exe, d = san[which], reads["dir"]
    want = unmapped_sam(reads["recs"])
    for name, args, env in VARIANTS:
        sam = os.path.join(d, f"{which}_{name}.sam")
        rc, out = run(exe, ["map", reads["fq"], "-o", sam] + args, env)
        assert rc == 0 and f"reads={len(want)} " in out, out
        assert records(sam) == want, name
    # the stand-in search switched on: mapped records (RNAME with spaces, minus strand text) -- the roads must still agree
    base = None
    for name, args, env in VARIANTS[:3] + VARIANTS[5:6]:
        sam = os.path.join(d, f"{which}_{name}.mapped.sam")
        rc, out = run(exe, ["map", reads["fq"], "-o", sam] + args, dict(env, URX_STUB_MAP="1"))
        assert rc == 0, out
        got = open(sam, "rb").read()
        base = base or got
        assert got == base, name
    assert b"\t16\t" in base or b"\t0\t" in base


@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_shards_pairs_tab_and_compressed_input(san, reads, which):
    exe, d, fq = san[which], reads["dir"], reads["fq"]
    one = os.path.join(d, f"{which}_one.sam")
    rc, out = run(exe, ["map", fq, "-o", one, "-batch", 2000], {"URX_STUB_MAP": "1"})
    assert rc == 0
    # three shards on 'three devices', five shards on one: cat == the one file
    for shards, gpus in ((3, 3), (5, 1)):
        sh = os.path.join(d, f"{which}_sh{shards}.sam")
        rc, out = run(exe, ["map", fq, "-o", sh, "-batch", 1500, "-shards", shards, "-gpus", gpus], {"URX_STUB_MAP": "1"})
        assert rc == 0 and f"shards={shards}" in out, out
        assert b"".join(open(f"{sh}.{k}", "rb").read() for k in range(shards)) == open(one, "rb").read()
    # gzip stream (parallel inflater), BGZF, standard input: the same SAM
    data = open(fq, "rb").read()
    gz, bg = os.path.join(d, "r.fq.gz"), os.path.join(d, "rb.fq.gz")
    open(gz, "wb").write(gzip.compress(data, 4))
    import bench
    open(bg, "wb").write(bench.bgzf_bytes(data))
    # (gzip at three chunk sizes: smaller than a segment's text -- through the reader's own buffer; room for two or three of a round's four segments -- a smaller
    # round written straight into the chunk, then a short read; room for whole rounds)
    for name, src, env, batch in (("gz", gz, {"URMAPX_PGZIP_SEGMENT": "65536"}, 1200), ("gz_fit", gz, {"URMAPX_PGZIP_SEGMENT": "65536"}, 3200),
                                  ("gz_rounds", gz, {"URMAPX_PGZIP_SEGMENT": "65536"}, 20000), ("bgzf", bg, {}, 1200)):
        sam = os.path.join(d, f"{which}_{name}.sam")
        rc, out = run(exe, ["map", src, "-o", sam, "-batch", batch], dict(env, URX_STUB_MAP="1"))
        assert rc == 0, out
        assert open(sam, "rb").read() == open(one, "rb").read(), name
    sam = os.path.join(d, f"{which}_stdin.sam")
    with open(fq, "rb") as f:
        rc, out = run(exe, ["map", "-", "-o", sam, "-batch", 1200], {"URX_STUB_MAP": "1"}, stdin=f)
    assert rc == 0 and open(sam, "rb").read() == open(one, "rb").read(), out
    # pairs with -tabbedout, text road against host road, one file against two shards
    recs2 = make_reads(len(reads["recs"]), 12)
    fq2 = os.path.join(d, "r2.fq")
    open(fq2, "wb").write(fastq_text([(l[:-1] + b"2", s, q) for l, s, q in recs2]))
    outs = {}
    for name, args, env in (("text", ["-batch", 1000], {}), ("host", ["-batch", 1000], {"URMAPX_HOST_TEXT": "1"}), ("shards", ["-batch", 800, "-shards", 2], {})):
        sam, tab = os.path.join(d, f"{which}_pe_{name}.sam"), os.path.join(d, f"{which}_pe_{name}.tab")
        rc, out = run(exe, ["map", fq, "-2", fq2, "-o", sam, "-tab", tab] + args, dict(env, URX_STUB_MAP="1"))
        assert rc == 0 and f"reads={2 * len(recs2)} " in out, out
        if name == "shards":
            outs[name] = (b"".join(open(f"{sam}.{k}", "rb").read() for k in range(2)), b"".join(open(f"{tab}.{k}", "rb").read() for k in range(2)))
        else:
            outs[name] = (open(sam, "rb").read(), open(tab, "rb").read())
    assert outs["text"] == outs["host"] == outs["shards"]
    assert outs["text"][0].count(b"\n") == 2 * len(recs2) + 4 and outs["text"][1].count(b"\n") == len(recs2)


def test_odd_but_legal_fastq_takes_the_host_reader(san, reads):
    """'\\r\\n' line ends, no final newline, blank lines at the end: the device parser hands such a chunk back and the host reader
    continues at its first byte (pipeline.cpp's resume) -- same records as the clean file"""
    exe, d = san["asan"], reads["dir"]
    recs = reads["recs"][:5000]
    want = unmapped_sam(recs)
    clean = fastq_text(recs)
    cases = {"crlf_from_the_middle": fastq_text(recs[:2500]) + fastq_text(recs[2500:], eol=b"\r\n"), "no_final_newline": clean[:-1],
             "blank_lines_at_the_end": clean + b"\n\n\n", "all_crlf": fastq_text(recs, eol=b"\r\n")}
    for name, data in cases.items():
        p, sam = os.path.join(d, name + ".fq"), os.path.join(d, name + ".sam")
        open(p, "wb").write(data)
        for exe_ in (san["asan"], san["tsan"]):
            rc, out = run(exe_, ["map", p, "-o", sam, "-batch", 400])
            assert rc == 0 and f"reads={len(recs)} " in out, (name, out)
            assert records(sam) == want, name


def test_fastq_reader_and_gunzip_under_asan(san, tmp_path):
    """urmapx_fastq_* and urmapx_gunzip_file (what tests/test_fastq_cpu.py and tests/test_pgzip_cpu.py check on the production library)
    in the instrumented build: the same records and bytes, and the stream shapes zlib writes"""
    import zlib
    import test_pgzip_cpu as tp
    exe = san["asan"]
    recs = make_reads(6000, 3)
    data = fastq_text(recs)
    p = os.path.join(tmp_path, "a.fq")
    open(p, "wb").write(data)
    rc, out = run(exe, ["fastq", p, 777])
    assert rc == 0 and f"records={len(recs)} bases={sum(len(s) for _, s, _ in recs)} " in out, out
    digest = out.split("digest=")[1].split()[0]
    open(p + ".gz", "wb").write(gzip.compress(data))
    rc, out = run(exe, ["fastq", p + ".gz", 5])
    assert rc == 0 and digest in out
    text = data * 3
    for case in sorted(tp.CASES):
        gz = tp.CASES[case](text)
        q = os.path.join(tmp_path, case + ".gz")
        open(q, "wb").write(gz)
        rc, out = run(exe, ["gunzip", q, q + ".out", 4], {"URMAPX_PGZIP_SEGMENT": "32768"})
        want = gzip.decompress(gz) if case != "trailing_garbage" else text
        assert rc == 0 and open(q + ".out", "rb").read() == want, (case, out)
    rc, out = run(san["tsan"], ["gunzip", q, q + ".out", 4], {"URMAPX_PGZIP_SEGMENT": "32768"})
    assert rc == 0


def _damage(data, rng, kind):
    b = bytearray(data)
    n = len(b)
    if kind == "flip":
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8))
    elif kind == "truncate":
        del b[int(rng.integers(1, n)):]
    elif kind == "zero":
        a = int(rng.integers(0, n - 1))
        k = int(rng.integers(1, min(5000, n - a) + 1))
        b[a:a + k] = bytes(k)
    elif kind == "random":
        a = int(rng.integers(0, n - 1))
        k = int(rng.integers(1, min(3000, n - a)))
        b[a:a + k] = rng.integers(0, 256, k, dtype=np.uint8).tobytes()
    elif kind == "drop":
        a = int(rng.integers(0, n - 1))
        del b[a:a + int(rng.integers(1, min(400, n - a)))]
    return bytes(b)


def test_corrupt_input_ends_in_an_error_code_never_a_report(san, tmp_path):
    """bit flips, truncations, zeroed and randomised ranges, dropped bytes -- in plain FASTQ, in a gzip stream, in a BGZF file -- through
    urmapx_map_files (both mates' roads), urmapx_gunzip_file and urmapx_fastq_*: rc 0 (the damage left a legal file) or rc 1 (refused with an
    error code), never a sanitizer report, a signal or a hang.  Seeded: a failure names its case"""
    import bench
    exe = san["asan"]
    recs = make_reads(3000, 21)
    data = fastq_text(recs)
    sources = {"plain": (data, ".fq"), "gzip": (gzip.compress(data, 6), ".fq.gz"), "bgzf": (bench.bgzf_bytes(data, block=20000), ".fq.gz")}
    rng = np.random.default_rng(2026)
    n_refused = n_ok = 0
    for src, (blob, suffix) in sources.items():
        for kind in ("flip", "truncate", "zero", "random", "drop"):
            for rep in range(5):
                bad = _damage(blob, rng, kind)
                p = os.path.join(tmp_path, f"{src}_{kind}_{rep}{suffix}")
                open(p, "wb").write(bad)
                sam = p + ".sam"
                rc, out = run(exe, ["map", p, "-o", sam, "-batch", 500], {"URMAPX_PGZIP_SEGMENT": "16384"})
                assert rc in (0, 1), (p, out)
                n_ok += rc == 0
                n_refused += rc == 1
                if rc == 1:
                    assert "rc=-" in out and "err=" in out
                if rep == 0:
                    rc2, out2 = run(exe, ["map", p, "-2", p, "-o", sam, "-batch", 500])
                    assert rc2 in (0, 1), (p, out2)
                    rc3, out3 = run(exe, ["fastq", p, 100])
                    assert rc3 in (0, 1)
                    if suffix.endswith(".gz"):
                        rc4, out4 = run(exe, ["gunzip", p, p + ".out", 4], {"URMAPX_PGZIP_SEGMENT": "16384"})
                        assert rc4 in (0, 1)
                os.remove(p)
                if os.path.exists(sam):
                    os.remove(sam)
    assert n_refused > 20  # most damage must be noticed
    # the thread sanitizer on a sample of the same
    for src, (blob, suffix) in sources.items():
        bad = _damage(blob, rng, "random")
        p = os.path.join(tmp_path, f"tsan_{src}{suffix}")
        open(p, "wb").write(bad)
        rc, out = run(san["tsan"], ["map", p, "-o", p + ".sam", "-batch", 500, "-shards", 2 if src == "plain" else 0])
        assert rc in (0, 1), out


def test_host_index_builder_under_asan(san, tmp_path):
    """urmapx_make_ufi (make_ufi.cpp, the host builder of -make_ufi) in the instrumented build: its .ufi is the oracle's"""
    import oracle_lib as ol
    from urmap_amd import synth
    g = synth.make_genome(5, [50000, 20000], repeat_frac=0.3, n_families=5)
    fa = os.path.join(tmp_path, "g.fa")
    synth.write_fasta(fa, g, lowercase_frac=0.05)
    ufi = os.path.join(tmp_path, "san.ufi")
    rc, out = run(san["asan"], ["makeufi", fa, ufi, 131101])
    assert rc == 0, out
    ol.Index.build(fa, 131101).save(os.path.join(tmp_path, "oracle.ufi"))
    assert open(ufi, "rb").read() == open(os.path.join(tmp_path, "oracle.ufi"), "rb").read()
