"""The bench's genome is a pure function of (seed, size): SURVEY.md 8(d) asks for a seeded in-repo generator.

Rounds 1-5 scattered overlapping repeat copies in one index-put (write order of duplicate indices undefined on the GPU) and drew
from the device's generator: every run mapped a slightly different genome.  bench.make_genome_torch now derives every
random value from a counter-based integer hash and lays overlapping copies in a defined order; this module pins the 40 Mbp store
(the size the GPU suite's small fixtures use) and the recorded values of the 3.1 Gbp store (tests/golden/bench_genome.json; the
GPU modules compare what they build on the device with the same file)."""
import hashlib
import json
import os

import numpy as np

import bench

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_genome.json")


def _gold():
    return json.load(open(GOLD))


def test_generator_twice_gives_equal_arrays_and_the_recorded_store():
    import torch
    a = bench.make_genome_torch(torch, 20260101, int(40e6), torch.device("cpu"))
    b = bench.make_genome_torch(torch, 20260101, int(40e6), torch.device("cpu"))
    assert torch.equal(a[0], b[0])
    assert (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[3] == b[3] and a[4] == b[4]
    s = a[0].numpy()
    g = _gold()["40"]
    assert s.size == g["bytes"]
    assert hashlib.sha256(s.tobytes()).hexdigest() == g["sha256"]
    assert f"{bench.array_checksum(s):016x}" == g["checksum"]
    slots, fasta = bench.default_slot_count(a[1], a[3])
    assert (slots, fasta) == (g["slots"], g["fasta_bytes"])
    # the layout -make_ufi writes: 32 '-' between sequences, nothing but ACGTN- anywhere
    assert set(np.unique(s).tolist()) <= set(b"ACGTN-")
    for i in range(len(a[1]) - 1):
        gap = s[int(a[2][i]) + int(a[1][i]): int(a[2][i]) + int(a[1][i]) + 32]
        assert (gap == ord("-")).all()


def test_another_seed_is_another_genome():
    import torch
    a = bench.make_genome_torch(torch, 1, int(2e6), torch.device("cpu"))[0]
    b = bench.make_genome_torch(torch, 2, int(2e6), torch.device("cpu"))[0]
    assert a.shape == b.shape and float((a != b).float().mean()) > 0.5


def _murmur64(h):
    m = (1 << 64) - 1
    h ^= h >> 33
    h = (h * 0xFF51AFD7ED558CCD) & m
    h ^= h >> 33
    h = (h * 0xC4CEB9FE1A85EC53) & m
    return h ^ (h >> 33)


def test_array_checksum_is_the_documented_sum():
    """include/urmapx.h: sum over the little-endian 64-bit words w_i (last one zero-padded) of murmur64(w_i + (i + 1) * golden)"""
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 8, 9, 4096, 100003):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        pad = bytes(a) + b"\0" * (-n % 8)
        want = 0
        for i in range(len(pad) // 8):
            w = int.from_bytes(pad[8 * i: 8 * i + 8], "little")
            want = (want + _murmur64((w + (i + 1) * 0x9E3779B97F4A7C15) & ((1 << 64) - 1))) & ((1 << 64) - 1)
        assert bench.array_checksum(a) == want, n


def test_recorded_full_scale_values_are_present():
    g = _gold()["3100"]
    assert g["bytes"] > 3_000_000_000 and g["slots"] == 5392814809
    assert len(g["checksum"]) == 16 and len(g["sha256"]) == 64
