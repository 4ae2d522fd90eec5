"""Guards the shipped kernels' register allocation against the toolchain: the search kernels' speed rests
on how many waves fit a SIMD and on how little they spill, and a compiler bump or a changed flag can move both silently.

The figures are read out of the gfx950 code objects inside the built liburmapx.so (tests/tools/kernel_meta.py: the
NT_AMDGPU_METADATA notes), i.e. of the binary that runs on the GPU box -- nothing is recompiled.  Ceilings are what the
round's profiled build has (profiles/r6/kernel_resources.txt), with no slack on occupancy and about 5 % on spill counts."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
SO = os.path.join(ROOT, "urmap_amd", "liburmapx.so")

# kernel: (min waves per SIMD, max VGPRs, max spilled VGPRs, max spilled SGPRs, max scratch bytes per lane, max LDS bytes)
CEILINGS = {
    # (template arguments: byte chunks, second pass, diagnostic, row layout (2 = slot16), phase-3 part, chunks of k-mer starts)
    "search_se_kernel<3, false, false, 2, 0, 2>": (4, 128, 34, 246, 112, 10000),  # 150-base reads, the headline kernel since round 6: two chunks of k-mer starts, row store in LDS
    "search_se_kernel<3, false, false, 2, 0, 3>": (4, 128, 62, 252, 168, 10048),  # reads of 152-192 bases at W = 24 (and URMAPX_NO_K2): three chunks, row store in global scratch
    "search_se_kernel<4, false, false, 2, 0, 4>": (3, 168, 4, 276, 16, 12560),    # 250-base reads
    "search_se_kernel<2, false, false, 2, 0, 2>": (4, 128, 28, 230, 112, 9808),   # reads of up to 128 bases (row store in LDS: 7 760 + 2 048 B)
    "search_se_kernel<3, false, false, 1, 0, 3>": (4, 128, 56, 270, 168, 10048),  # the same on an index without slot16 (row layout only)
    "search_se_kernel<3, false, false, 1, 1, 3>": (4, 128, 88, 292, 64, 10048),   # URMAPX_PARK_PHASE3=1 (opt-in, measured slower): first launch (no banded DP inside)
    "search_se_kernel<3, false, false, 1, 2, 3>": (4, 120, 0, 150, 0, 10048),     # ... second launch (the reads parked at phase 3)
    "search_pe_kernel<3, 0>": (4, 128, 80, 302, 304, 10240),                    # 2 x 150 pairs; round 6: the LDS diet (kernels_pe.hip: URX_PE_DIET 2) -- FOUR waves per SIMD, 16 blocks x 10 240 B = a CU's 160 KB
    "search_pe_kernel<2, 0>": (4, 128, 75, 296, 272, 8256),                     # pairs of reads of up to 128 bases, same diet
    "dp_kernel<3>": (6, 80, 0, 65, 8, 3456),
    "dp_kernel<4>": (6, 80, 0, 65, 8, 4352),
    "finalize_se_kernel<3, false>": (8, 64, 0, 30, 0, 4480),
    "finalize_se_kernel<4, false>": (8, 64, 0, 30, 0, 4480),
    "seed_probe_kernel<3>": (8, 64, 0, 0, 0, 0),
    "validate_kernel": (8, 64, 0, 0, 0, 64),
    "slot16_kernel": (8, 32, 0, 0, 0, 0),
    # the text stage of the file-to-file lanes (text_gpu.hip)
    "sam_kernel": (5, 88, 0, 100, 44, 30992),        # 64 heads per wavefront in LDS: 4 x 64 x 96 B + the 1 600-byte buffers of long heads
    "sam_len_kernel": (8, 48, 0, 0, 0, 10256),
    "copy_bases_kernel": (8, 32, 0, 0, 0, 0),
    "record_kernel": (8, 32, 0, 0, 0, 0),
}


@pytest.fixture(scope="module")
def table():
    import kernel_meta
    if not os.path.exists(SO):
        pytest.fail(f"{SO} is missing: run __graft_entry__.build() first")
    return kernel_meta.kernel_table(SO)


@pytest.mark.parametrize("kernel", sorted(CEILINGS))
def test_shipped_kernel_stays_inside_its_recorded_resources(table, kernel):
    assert kernel in table, f"{kernel} is not in liburmapx.so; it holds: {sorted(k for k in table if k.split('<')[0] == kernel.split('<')[0])}"
    waves, vgpr, vspill, sspill, scratch, lds = CEILINGS[kernel]
    r = table[kernel]
    assert r["waves_per_simd"] >= waves, (kernel, r)
    assert r["vgpr"] + r["agpr"] <= vgpr, (kernel, r)
    assert r["vgpr_spill"] <= vspill, (kernel, r)
    assert r["sgpr_spill"] <= sspill, (kernel, r)
    assert r["scratch"] <= scratch, (kernel, r)
    assert r["lds"] <= lds, (kernel, r)


def test_every_kernel_of_the_library_is_a_wave64_gfx950_kernel(table):
    assert len(table) >= 60
    for k, r in table.items():
        assert r["max_threads"] % 64 == 0 or r["max_threads"] <= 64, (k, r)


def test_lds_per_block_allows_the_occupancy_the_registers_allow(table):
    """160 KB of LDS per CU, 4 SIMDs: blocks of one wave need waves_per_simd * 4 * lds <= 160 KB"""
    for k in ("search_se_kernel<3, false, false, 2, 0, 2>", "search_se_kernel<3, false, false, 2, 0, 3>", "search_pe_kernel<3, 0>", "search_se_kernel<4, false, false, 2, 0, 4>"):
        r = table[k]
        assert r["max_threads"] == 64
        assert r["waves_per_simd"] * 4 * r["lds"] <= 160 * 1024, (k, r)
