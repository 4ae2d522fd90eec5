"""File-to-file experiments (urmapx_map_files) on one GPU: a synthetic genome, its index built on the GPU, N reads written as
FASTQ to /dev/shm, then the pipeline under different settings.  Prints one JSON line per setting.
  python scripts/e2e_probe.py --genome-mbp 400 --reads 4000000 --set text:2:262144 --set host:2:262144 ..."""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=400)
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--set", action="append", default=[], help="mode:streams:batch[:env=val,...]  mode = text | host")
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    import torch
    from urmap_amd import api, ranks
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    R = ranks.Ranks().init(torch)
    total_bp = int(args.genome_mbp * 1e6)
    d_seq, seq_lengths, seq_offsets, labels, _ = bench.make_genome_torch(torch, 20260101, total_bp, device)
    slots, _ = bench.default_slot_count(seq_lengths, labels)
    index, blob_np, seq_np, d_seq, t_index = bench.place_index(R, torch, api, device, d_seq, slots, seq_lengths, seq_offsets, labels)
    d = tempfile.mkdtemp(prefix="urmap_probe_", dir="/dev/shm")
    try:
        L = args.read_len
        reads = bench.make_reads_torch(torch, 777, d_seq, seq_lengths, seq_offsets, args.reads, L, 0.01, 0.002, device).cpu().numpy()
        fq = os.path.join(d, "r.fq")
        fq_bytes = bench.write_fastq_fixed(fq, reads, args.reads, L)
        del reads
        sums = {}
        for spec in args.set or ["text:2:262144", "host:2:262144"]:
            parts = spec.split(":")
            mode, streams, batch = parts[0], int(parts[1]), int(parts[2])
            env = dict(kv.split("=") for kv in parts[3].split(",")) if len(parts) > 3 else {}
            for k in ("URMAPX_HOST_TEXT", "URMAPX_SAM_WRITE", "URMAPX_NO_PIN", "URMAPX_WRITE_THREADS"):
                os.environ.pop(k, None)
            if mode == "host":
                os.environ["URMAPX_HOST_TEXT"] = "1"
            os.environ.update(env)
            sam = os.path.join(d, "out.sam")
            best = None
            all_s = []
            for _ in range(args.repeat):
                if os.path.exists(sam):
                    os.unlink(sam)
                t0 = time.time()
                rep = api.map_files(index, fq, samout=sam, first_gpu=0, gpus=1, streams=streams, batch=batch, cmdline="probe")
                wall = time.time() - t0
                all_s.append(round(rep["seconds"], 3))
                if best is None or rep["seconds"] < best["seconds"]:
                    best = dict(rep)
                    best["wall_call_s"] = round(wall, 3)
            h = hashlib.md5()
            with open(sam, "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            sums[spec] = h.hexdigest()
            print(json.dumps({"set": spec, "reads_per_s": round(best["reads"] / best["seconds"], 1), "fastq_GB": round(fq_bytes / 1e9, 3),
                              "sam_GB": round(os.path.getsize(sam) / 1e9, 3), "md5": sums[spec][:12], "runs_s": all_s,
                              **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in best.items()}}), flush=True)
        print(json.dumps({"all_sam_identical": len(set(sums.values())) == 1}))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
