"""Register / spill / scratch / LDS figures of every gfx950 kernel inside the shipped liburmapx.so (test infrastructure).

The library's .hip_fatbin section holds one clang offload bundle per translation unit; each bundle's gfx950 entry is an ELF
code object whose NT_AMDGPU_METADATA note lists, per kernel, what the compiler allocated.  Nothing is compiled here: the
numbers are those of the binary that runs."""
import os
import re
import struct
import subprocess
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = {"vgpr": ".vgpr_count", "agpr": ".agpr_count", "sgpr": ".sgpr_count", "vgpr_spill": ".vgpr_spill_count", "sgpr_spill": ".sgpr_spill_count",
          "scratch": ".private_segment_fixed_size", "lds": ".group_segment_fixed_size", "max_threads": ".max_flat_workgroup_size"}


def code_objects(so_path):
    """bytes of every gfx950 code object bundled in the shared library"""
    raw = open(so_path, "rb").read()
    out, i = [], 0
    while True:
        i = raw.find(MAGIC, i)
        if i < 0:
            break
        n = struct.unpack_from("<Q", raw, i + 24)[0]
        p = i + 32
        if 0 < n < 16:
            for _ in range(n):
                off, size, ts = struct.unpack_from("<QQQ", raw, p)
                p += 24
                triple = raw[p:p + ts]
                p += ts
                if b"gfx950" in triple and size:
                    out.append(raw[i + off:i + off + size])
        i += len(MAGIC)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return r.stdout.split("\n")[:len(names)]


def waves_per_simd(vgpr, agpr=0):
    """gfx950: 512 VGPRs per SIMD lane (unified with AGPRs), allocated in blocks of 8, at most 8 waves per SIMD"""
    tot = max(1, -(-(vgpr + agpr) // 8) * 8)
    return max(1, min(8, 512 // tot))


def kernel_table(so_path):
    """{demangled kernel name without its argument list: {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, max_threads, waves_per_simd}}"""
    table = {}
    for co in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n  - \.agpr_count:", "\n" + txt)[1:]:
            blk = "    .agpr_count:" + blk
            m = re.search(r"\.name:\s+(\S+)", blk)
            if not m:
                continue
            row = {}
            for k, f_ in FIELDS.items():
                mm = re.search(re.escape(f_) + r":\s+(\d+)", blk)
                row[k] = int(mm.group(1)) if mm else 0
            row["waves_per_simd"] = waves_per_simd(row["vgpr"], row["agpr"])
            table[m.group(1)] = row
    names = list(table)
    out = {}
    for mangled, nice in zip(names, demangle(names)):
        # "void urx::search_se_kernel<3, false, false, true>(urx::DevIndex, ...)" -> "search_se_kernel<3, false, false, true>"
        nice = re.sub(r"^void\s+", "", nice).replace("(anonymous namespace)::", "")  # (before the cut at the argument list's parenthesis)
        depth, cut = 0, len(nice)
        for j, ch in enumerate(nice):
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = j
                break
        nice = nice[:cut].replace("urx::", "").replace("(anonymous namespace)::", "")
        out[nice] = table[mangled]
    return out


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "urmap_amd", "liburmapx.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    t = kernel_table(so)
    print(f"{'kernel':64s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s} {'waves':>5s}")
    for k in sorted(t):
        if pat in k:
            r = t[k]
            print(f"{k:64s} {r['vgpr']:5d} {r['sgpr']:5d} {r['vgpr_spill']:6d} {r['sgpr_spill']:6d} {r['scratch']:7d} {r['lds']:6d} {r['waves_per_simd']:5d}")
