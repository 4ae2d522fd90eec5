"""CPU, world_size 2 (gloo): the multi-GPU path is "read batches sharded by rank, index replicated, no collective on
the data path" (the reference's unit is an OpenMP thread, map.cpp:58-61).  bench.py starts its own ranks
through urmap_amd.ranks.launch_ranks when run as plain `python bench.py --gpus N`; this drives the same launcher and the
same init / barrier / reductions / broadcast with a CPU script."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from urmap_amd import ranks  # noqa: E402

PROBE = os.path.join(ROOT, "tests", "tools", "rank_probe.py")


def _launch(n, n_reads, batch):
    env = dict(os.environ)
    env["URMAP_RANK_DEVICES"] = "0"
    port = ranks.free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), PROBE, str(n_reads), str(batch)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    return json.loads(lines[0])


def test_two_ranks_cover_every_read_once_and_reduce():
    d = _launch(2, 100003, 4096)
    assert d["world"] == 2 and d["backend"] == "gloo"
    assert d["total"] == 100003 and d["covered_once"] and d["broadcast_ok"]
    assert d["slowest"] == 2.0   # max over ranks, as the bench contract times a step
    assert d["gathered"] == [[0.0, 10.0], [1.0, 11.0]]  # every rank's own numbers, in rank order


def test_eight_ranks_as_the_drivers_scaling_run_launches_them():
    """config 4's world size (8 ranks, here without devices): sharding, the reductions and the per-rank gather"""
    d = _launch(8, 1000003, 65536)
    assert d["world"] == 8 and d["total"] == 1000003 and d["covered_once"] and d["broadcast_ok"]
    assert d["slowest"] == 8.0 and d["gathered"] == [[float(r), 10.0 + r] for r in range(8)]


def test_launch_ranks_starts_children_and_returns_their_code(tmp_path):
    """ranks.launch_ranks is what `python bench.py --gpus N` calls before touching the GPU."""
    ok = tmp_path / "ok.py"
    ok.write_text("import os, sys\nassert os.environ['WORLD_SIZE'] == '2' and os.environ['URMAP_RANK_DEVICES'] == '0'\n"
                  "open(sys.argv[1] + os.environ['RANK'], 'w').write('x')\n")
    assert ranks.launch_ranks(str(ok), [str(tmp_path / "seen")], 2, n_devices=0, timeout=300) == 0
    assert (tmp_path / "seen0").exists() and (tmp_path / "seen1").exists()
    bad = tmp_path / "bad.py"
    bad.write_text("import sys\nsys.exit(3)\n")
    assert ranks.launch_ranks(str(bad), [], 2, n_devices=0, timeout=300) != 0


def test_batches_cover_everything_single_rank():
    b = ranks.batches_for_rank(10, 4, 0, 1)
    assert b == [(0, 4), (4, 8), (8, 10)]
    assert ranks.batches_for_rank(10, 4, 1, 3) == [(4, 8)]
    assert ranks.batches_for_rank(0, 4, 0, 2) == []


def test_device_assignment_rules(monkeypatch):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "URMAP_RANK_DEVICES", "URMAP_BENCH_FORCE_DEVICE"):
        monkeypatch.delenv(k, raising=False)
    assert not ranks.launched()
    r = ranks.Ranks()
    assert (r.rank, r.world, r.device_index, r.shared) == (0, 1, 0, False)
    monkeypatch.setenv("RANK", "5"); monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("LOCAL_RANK", "5")
    assert ranks.launched()
    r = ranks.Ranks()
    assert (r.device_index, r.shared) == (5, False)          # one GPU per rank: RCCL
    monkeypatch.setenv("URMAP_RANK_DEVICES", "2")
    r = ranks.Ranks()
    assert (r.device_index, r.shared) == (1, True)           # ranks outnumber devices: share, gloo
