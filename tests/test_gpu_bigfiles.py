"""File offsets past 2^32 through the file-to-file path: BASELINE config 4 is 100 M reads -- 31.5 GB of FASTQ
in, 34 GB of SAM out -- and even one device's share of it (12.5 M reads) writes 4.3 GB, while every other module of the suite
moves files of at most 3.5 GB.  Here 14.2 M 150-base reads (4.47 GB of FASTQ -> 4.9 GB of SAM) go through urmapx_map_files
(cmd_map, map.cpp:43-61; the reader of linereader.cpp:54-113) against a small index, once into one file and once into two
shards, so that the reader's cut points, the writer's pwrite offsets, the shard cutter's record search and the line counts
all pass 4 GiB:
  * every record of the one file is there, in input order (labels r00000000 .. by position: a piece written at a wrapped
    offset would land on top of the head of the file and leave a hole behind 4 GiB);
  * the head, the 100 k records that straddle byte 2^32 of the SAM file, the 100 k reads that straddle byte 2^32 of the
    FASTQ file and the last 100 k records are the oracle's, byte for byte;
  * `cat` of the two shards is the one file.
A library whose writer cuts its file offset to 32 bits (make EXTRA=-DURX_FAULT_OFF32) fails the first of these
(profiles/r6/fault_off32.txt).

The genome is the bench's generator at 40 Mbp; its checksum on the device must be the one tests/test_genome_cpu.py pins on
the CPU (the generator is device-independent since round 6), and the library's checksum must be the numpy restatement's."""
import json
import os
import shutil
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_READS = int(os.environ.get("URMAP_TEST_BIG_READS", 14_200_000))
L = 150
REC = 11 + L + 3 + L + 1  # "@r%08d\n" bases "\n+\n" quals "\n"
SLICE = 100_000
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_genome.json")


@pytest.fixture(scope="module")
def big():
    import torch
    import bench
    import oracle_lib as ol
    from urmap_amd import api, ranks
    dev = torch.device("cuda", 0)
    R = ranks.Ranks().init(torch)
    d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(40e6), dev)
    genome_checksum = api.checksum_device(0, d_seq.data_ptr(), int(d_seq.numel()))
    slots, fasta_bytes = bench.default_slot_count(lens, labels)
    index, blob_np, seq_np, d_seq, info = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
    oi = ol.Index.wrap(24, 32, slots, blob_np, seq_np, lens, offs, labels)
    d = tempfile.mkdtemp(prefix="urmap_big_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    # the reads, a slab at a time (the generator's index arrays are 8 bytes per base), straight into the FASTQ file
    fq = os.path.join(d, "big.fq")
    slab = 2_000_000
    with open(fq, "wb") as f:
        for lo in range(0, N_READS, slab):
            n = min(slab, N_READS - lo)
            r = bench.make_reads_torch(torch, 9000 + lo // slab, d_seq, lens, offs, n, L, 0.01, 0.001, dev).cpu().numpy()
            a = np.empty((n, REC), dtype=np.uint8)
            a[:, 0] = ord("@"); a[:, 1] = ord("r")
            idx = np.arange(lo, lo + n, dtype=np.int64)
            for k in range(8):
                a[:, 2 + k] = ((idx // 10 ** (7 - k)) % 10 + ord("0")).astype(np.uint8)
            a[:, 10] = ord("\n")
            a[:, 11:11 + L] = r.reshape(n, L)
            a[:, 11 + L] = ord("\n"); a[:, 12 + L] = ord("+"); a[:, 13 + L] = ord("\n")
            a[:, 14 + L:14 + 2 * L] = ord("I")
            a[:, REC - 1] = ord("\n")
            a.tofile(f)
            del a, r
    yield {"torch": torch, "bench": bench, "api": api, "index": index, "oi": oi, "dir": d, "fq": fq, "cores": bench.host_cores(),
           "genome_checksum": genome_checksum, "seq": seq_np, "blob": blob_np, "slots": slots}
    shutil.rmtree(d, ignore_errors=True)
    index.close()
    del d_seq
    torch.cuda.empty_cache()


def test_the_genome_on_the_device_is_the_recorded_one(big):
    """same generator, same seed, other device: the store the GPU built is the store tests/test_genome_cpu.py pins on the CPU"""
    g = json.load(open(GOLD))["40"]
    assert f"{big['genome_checksum']:016x}" == g["checksum"]
    assert big["slots"] == g["slots"]
    table, seq = big["index"].checksum()  # urmapx_index_checksum over the resident arrays == the numpy restatement on the host arrays
    assert seq == big["genome_checksum"] == big["bench"].array_checksum(big["seq"])
    assert table == big["bench"].array_checksum(np.asarray(big["blob"][: 5 * big["slots"]]))
    assert f"{table:016x}" == g["slot_table_checksum"]  # the GPU-assisted builder's table == the host builder's, recorded in the build container


def test_checksum_of_odd_sizes_and_a_misaligned_pointer(big):
    """urmapx_checksum_device: the documented sum for sizes around the word and the block boundaries; a pointer that is not 8-byte aligned is refused"""
    torch, api, bench = big["torch"], big["api"], big["bench"]
    g = torch.Generator(device="cuda").manual_seed(3)
    for n in (1, 7, 8, 9, 2047, 2048, 2049, (1 << 20) + 5, 40_000_003):
        t = torch.randint(0, 256, (n + 16,), dtype=torch.uint8, generator=g, device="cuda")
        assert t.data_ptr() % 8 == 0
        assert api.checksum_device(0, t.data_ptr(), n) == bench.array_checksum(t[:n].cpu().numpy()), n
    with pytest.raises(api.UrmapxError):
        api.checksum_device(0, t.data_ptr() + 4, 64)


def _sam_lines(path):
    """(offset of the first record, offsets of every record's first byte, file size): newline positions, a piece at a time"""
    size = os.path.getsize(path)
    starts = [np.zeros(1, np.int64)]
    piece = 256 << 20
    with open(path, "rb") as f:
        for lo in range(0, size, piece):
            b = np.frombuffer(f.read(piece), dtype=np.uint8)
            starts.append(np.flatnonzero(b == 10).astype(np.int64) + (lo + 1))
    starts = np.concatenate(starts)
    assert starts[-1] == size, "the file does not end with a newline"
    starts = starts[:-1]
    # header lines begin with '@'
    with open(path, "rb") as f:
        n_hdr = 0
        while f.read(1) == b"@":
            n_hdr += 1
            f.seek(int(starts[n_hdr]))
    return starts[n_hdr:], size


def _records(path, starts, size, lo, hi):
    a, b = int(starts[lo]), int(starts[hi]) if hi < len(starts) else size
    with open(path, "rb") as f:
        f.seek(a)
        return f.read(b - a).split(b"\n")[:-1]


def _oracle_records(big, lo, hi, name):
    """the oracle's SAM records for reads [lo, hi) of the FASTQ file"""
    d = big["dir"]
    part, out = os.path.join(d, name + ".fq"), os.path.join(d, name + ".oracle.sam")
    with open(big["fq"], "rb") as f, open(part, "wb") as g:
        f.seek(lo * REC)
        g.write(f.read((hi - lo) * REC))
    big["oi"].map_file_se(part, out, threads=big["cores"])
    want = [l for l in open(out, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    os.remove(part); os.remove(out)
    return want


def test_one_file_and_two_shards_past_four_gib(big):
    api, d, fq = big["api"], big["dir"], big["fq"]
    assert os.path.getsize(fq) == N_READS * REC
    full_size = N_READS >= 14_000_000
    if full_size:
        assert os.path.getsize(fq) > 2 ** 32
    sam = os.path.join(d, "one.sam")
    rep = api.map_files(big["index"], fq, samout=sam, first_gpu=0, gpus=1, streams=2, cmdline="test big files")
    assert rep["reads"] == N_READS and rep["text_on_device"] == 1
    starts, size = _sam_lines(sam)
    if full_size:
        assert size > 2 ** 32
    assert len(starts) == N_READS
    # every record in its place: the 8 digits behind the 'r' of record k spell k
    with open(sam, "rb") as f:
        step = 1_000_000
        for lo in range(0, N_READS, step):
            hi = min(N_READS, lo + step)
            a, b = int(starts[lo]), int(starts[hi]) if hi < N_READS else size
            f.seek(a)
            buf = np.frombuffer(f.read(b - a), dtype=np.uint8)
            rel = (starts[lo:hi] - a).astype(np.int64)
            assert (buf[rel] == ord("r")).all()
            num = np.zeros(hi - lo, np.int64)
            for k in range(8):
                num = num * 10 + (buf[rel + 1 + k].astype(np.int64) - ord("0"))
            assert (num == np.arange(lo, hi)).all(), f"records {lo}..{hi} are not the reads {lo}..{hi} in order"
    # slices against the oracle: head, across byte 2^32 of the SAM file, across byte 2^32 of the FASTQ file, tail
    k_sam = int(np.searchsorted(starts, 2 ** 32)) if size > 2 ** 32 else N_READS // 2
    k_fq = (2 ** 32) // REC if N_READS * REC > 2 ** 32 else N_READS // 3
    cuts = {"head": 0, "sam_4gib": max(0, min(N_READS - SLICE, k_sam - SLICE // 2)), "fastq_4gib": max(0, min(N_READS - SLICE, k_fq - SLICE // 2)),
            "tail": N_READS - SLICE}
    for name, lo in cuts.items():
        want = _oracle_records(big, lo, lo + SLICE, name)
        got = _records(sam, starts, size, lo, lo + SLICE)
        assert len(want) == SLICE
        assert got == want, f"slice {name} (reads {lo}..{lo + SLICE}) differs from the oracle's SAM"
    # two pipelines, two files: the cutter's record search and the second shard's reader start behind 2 GiB, its writer
    # ends behind it; cat == the one file
    rep2 = api.map_files(big["index"], fq, samout=sam + ".sh", first_gpu=0, gpus=1, streams=2, cmdline="test big files", sam_shards=2)
    assert rep2["reads"] == N_READS and rep2["shards"] == 2
    parts = [sam + ".sh.0", sam + ".sh.1"]
    assert all(os.path.getsize(p) > 0 for p in parts)
    assert big["bench"].files_equal_concat(sam, parts)
    for p in parts + [sam]:
        os.remove(p)


def test_mates_files_past_four_gib_each_side_of_the_cut(big):
    """-map2 over the same file as both mates' files (each 4.47 GB): the mate file's cut points are line counts looked up
    past 2^32 (count_newlines / line_offsets / skip_lines of pipeline.cpp), in two shards; record count and the last pair's
    labels are checked, and the tail slice against the oracle"""
    api, d, fq = big["api"], big["dir"], big["fq"]
    n_pairs = N_READS
    sam = os.path.join(d, "pairs.sam")
    rep = api.map_files(big["index"], fq, fq, samout=sam, first_gpu=0, gpus=1, streams=2, cmdline="test big files", sam_shards=2)
    assert rep["reads"] == 2 * n_pairs and rep["shards"] == 2
    parts = [sam + ".0", sam + ".1"]
    tail_n = 20_000
    # the last records of the second shard are the last pairs: mate 1 and mate 2 of read N-1 carry its label
    size = os.path.getsize(parts[1])
    with open(parts[1], "rb") as f:
        f.seek(max(0, size - 2 * tail_n * 700))
        lines = f.read().split(b"\n")[:-1][-2 * tail_n:]
    assert len(lines) == 2 * tail_n
    labels = [int(l.split(b"\t", 1)[0][1:]) for l in lines]
    assert labels == [N_READS - tail_n + k // 2 for k in range(2 * tail_n)]
    # the oracle on the same tail
    part, out = os.path.join(d, "ptail.fq"), os.path.join(d, "ptail.oracle.sam")
    with open(fq, "rb") as f, open(part, "wb") as g:
        f.seek((N_READS - tail_n) * REC)
        g.write(f.read(tail_n * REC))
    big["oi"].map_file_pe(part, part, out, threads=big["cores"])
    want = [l for l in open(out, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    assert lines == want
    total = 0
    for p in parts:
        with open(p, "rb") as f:
            while True:
                b = f.read(256 << 20)
                if not b:
                    break
                total += b.count(b"\n")
        os.remove(p)
    hdr = 24 + 1  # @SQ per sequence + @PG
    assert total == 2 * n_pairs + hdr
