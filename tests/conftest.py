import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_finish(session):
    """A process that uses both torch and liburmapx must let torch initialise the GPU first (the other order can
    leave torch without a device -- "No HIP GPUs are available").  The full-scale module builds its genome with torch; whichever
    GPU module the selection starts with, torch goes first."""
    if any(item.get_closest_marker("gpu") is not None for item in session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.zeros(1, device="cuda")
        except Exception:
            pass


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("urmap"))


@pytest.fixture(scope="session")
def small_case(workdir):
    """A 300 kbp, 3-sequence synthetic genome with repeats and N runs, its index built by the ORACLE
    (byte-identical to the reference's -make_ufi, see test_oracle_golden.py), and mixed reads."""
    import oracle_lib as ol
    from urmap_amd import synth

    g = synth.make_genome(101, [180000, 90000, 30000], repeat_frac=0.4, n_families=12)
    fa = os.path.join(workdir, "small.fa")
    synth.write_fasta(fa, g, lowercase_frac=0.05)
    slots = 524309  # prime
    idx = ol.Index.build(fa, slots)
    ufi = os.path.join(workdir, "small.ufi")
    idx.save(ufi)
    return {"genome": g, "fasta": fa, "ufi": ufi, "oracle_index": idx, "dir": workdir}


def reads_to_arrays(reads):
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r[1]) for r in reads])
    bases = np.concatenate([r[1] for r in reads]) if reads else np.zeros(0, np.uint8)
    return np.ascontiguousarray(bases, dtype=np.uint8), offs
