"""Parity at the headline configuration's scale: a 3.1 Gbp genome, 5 392 814 809 slots (more than 2^32: bit 32 of a slot number,
byte offsets up to 27 GB, positions above 2^31), the index built once per module by the product's -make_ufi passes.

Every other module of the suite works on tables of at most 2.0 G slots; a slot number cut to 32 bits, a 32-bit byte offset or
a launch of more than 2^32 work-items passes there and fails here.  The table is checked by the device's UFIndex::Validate
pass first, then 20 k SE 150 bp reads, 10 k pairs and 10 k SE 250 bp reads go against the oracle (which reads the same table on
the host), with the chain-row layout on and off, and one file-to-file run writes two SAM shards.

URMAP_TEST_FULLSCALE_MBP shrinks the genome (rehearsals on a small box); the assertions about 2^32 then do not apply."""
import os
import shutil
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MBP = float(os.environ.get("URMAP_TEST_FULLSCALE_MBP", 3100))
FULL = MBP >= 2600  # GetPrime(3.1 Gbp FASTA / 0.6) = 5 392 814 809 > 2^32


@pytest.fixture(scope="module")
def full():
    import torch
    import bench
    import oracle_lib as ol
    from urmap_amd import api, ranks
    dev = torch.device("cuda", 0)
    R = ranks.Ranks().init(torch)
    assert R.world == 1
    d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(MBP * 1e6), dev)
    slots, fasta_bytes = bench.default_slot_count(lens, labels)
    index, blob_np, seq_np, d_seq, info = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
    oi = ol.Index.wrap(24, 32, slots, blob_np, seq_np, lens, offs, labels)
    d = tempfile.mkdtemp(prefix="urmap_full_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    yield {"torch": torch, "bench": bench, "api": api, "dev": dev, "d_seq": d_seq, "lens": lens, "offs": offs, "labels": labels, "slots": slots,
           "index": index, "blob": blob_np, "seq": seq_np, "oi": oi, "dir": d, "cores": bench.host_cores(), "info": info}
    shutil.rmtree(d, ignore_errors=True)
    index.close()
    del d_seq
    torch.cuda.empty_cache()


def test_the_table_is_past_2_to_the_32_and_validates(full):
    """UFIndex::Validate (ufindex.cpp:611-658) on the resident table: every stored position re-hashed, every chain followed"""
    if FULL:
        assert full["slots"] == 5392814809
        assert 5 * full["slots"] > 2 ** 34
    ok, rep = full["index"].validate()
    assert ok, rep
    assert rep["slots"] == full["slots"] and rep["used"] == rep["reached"]
    # the table's own counts on the host: used slots, heads (numpy over a 27 GB array: tallies only, in pieces)
    blob, n = full["blob"], full["slots"]
    used = heads = 0
    step = 1 << 28
    for lo in range(0, n, step):
        t = np.asarray(blob[5 * lo: 5 * min(n, lo + step): 5])
        used += int(np.count_nonzero(t))
        heads += int(np.count_nonzero(t >= 128))
    assert (rep["used"], rep["heads"]) == (used, heads)
    if FULL:  # the upper fifth of the table (slot numbers with bit 32 set) holds its share of the positions
        t = np.asarray(blob[5 * (1 << 32): 5 * n: 5])
        assert np.count_nonzero(t) > 0.3 * len(t)
        assert rep["positions"] > 2 ** 31


def _run(full, pe, L, sub, indel, n, seed, mappers):
    b = full["bench"]
    wl = b.Workload(full["torch"], full["api"], full["dev"], full["d_seq"], full["lens"], full["offs"], pe, L, sub, indel, n, 1, seed,
                    streams=len(mappers))
    wl.timed(mappers, 1, 0)
    par, cnt, _ = wl.check(full["oi"], n, full["cores"])
    return wl, par


CASES = [("se150", False, 150, 0.01, 0.001, 20000), ("pe2x150", True, 150, 0.01, 0.001, 20000), ("se250", False, 250, 0.04, 0.01, 10000)]


@pytest.mark.parametrize("name,pe,L,sub,indel,n", CASES)
def test_reads_against_the_oracle(full, name, pe, L, sub, indel, n):
    """every field SAM is made of (position, strand, scores, MAPQ, path) for reads over the whole table; the k-mers of the
    batch land on both sides of slot 2^32"""
    api = full["api"]
    m = api.Mapper(full["index"], device=0)
    wl, par = _run(full, pe, L, sub, indel, n, 4242, [m])
    assert par["bit_identical_to_oracle"], par
    assert par["mapped_frac"] > 0.8
    if not pe:
        assert par["exit_phase_equal"]
    if FULL and name == "se150":
        hb = wl.last[: 200 * L].cpu().numpy()
        ho = np.arange(201, dtype=np.uint64) * L
        slots, tallies, positions = m.seed_probe(hb, ho)
        real = slots[slots != np.iinfo(np.uint64).max]
        assert (real >= 2 ** 32).sum() > 0.1 * len(real) and (real < 2 ** 32).sum() > 0.5 * len(real)
        assert int(real.max()) < full["slots"]
        # ... and those slots hold what the table holds there
        big = np.nonzero((slots != np.iinfo(np.uint64).max) & (slots >= 2 ** 32))[0][:2000]
        sv = slots[big].astype(np.int64)
        blob = full["blob"]
        assert (tallies[big] == np.array([blob[5 * int(s)] for s in sv], np.uint8)).all()
        want = np.array([int.from_bytes(bytes(blob[5 * int(s) + 1: 5 * int(s) + 5]), "little") for s in sv], np.uint32)
        assert (positions[big] == want).all()
        assert (want >= 2 ** 31).any()


def test_chain_rows_off_gives_the_same(full, monkeypatch):
    """the link-by-link walk (what an index whose rows do not fit keeps) over the same table: a second replica without rows"""
    api = full["api"]
    assert full["index"].chain_row_bytes() > 4 * full["slots"]
    monkeypatch.setenv("URMAPX_NO_CHAIN_ROWS", "1")
    idx2 = api.Index.wrap_host(24, 32, full["slots"], full["blob"], full["seq"], full["lens"], full["offs"], full["labels"]).upload(0)
    try:
        assert idx2.chain_row_bytes() == 0
        m = api.Mapper(idx2, device=0)
        for name, pe, L, sub, indel, n in CASES:
            _, par = _run(full, pe, L, sub, indel, n // 2, 777, [m])
            assert par["bit_identical_to_oracle"], (name, par)
        del m
    finally:
        idx2.close()


def test_two_sam_shards_file_to_file(full):
    """urmapx_map_files (= urmap -map -samshards 2) on the resident table: `cat` of the shards = the oracle's records"""
    b, api, torch = full["bench"], full["api"], full["torch"]
    n, L = 20000, 150
    reads = b.make_reads_torch(torch, 31337, full["d_seq"], full["lens"], full["offs"], n, L, 0.01, 0.001, full["dev"]).cpu().numpy()
    fq, sam, sam_o = (os.path.join(full["dir"], x) for x in ("r.fq", "out.sam", "oracle.sam"))
    b.write_fastq_fixed(fq, reads, n, L)
    rep = api.map_files(full["index"], fq, samout=sam, first_gpu=0, gpus=1, streams=2, batch=4096, sam_shards=2, cmdline="test")
    assert rep["reads"] == n and rep["shards"] == 2 and rep["unsupported"] == 0
    full["oi"].map_file_se(fq, sam_o, threads=full["cores"])
    want = [l for l in open(sam_o, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    got = []
    for k in range(2):
        got += [l for l in open(f"{sam}.{k}", "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    assert len(got) == n and got == want
