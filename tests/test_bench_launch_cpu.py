"""`python bench.py --gpus N` must start N ranks itself (the reference's parallel mode is N worker processes,
launch/oxford/eval/utils/start_workers:54-57) and refuse to print a line for a different world size. Runs on a CPU-only host
through bench.py's dry-run leg: gloo instead of RCCL and a stub step, everything else (launcher, sequence sharding, barriers,
max-over-ranks time, SUM/MAX reduction, one JSON line from rank 0) is the code the GPU run uses."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_launch_argv():
    sys.path.insert(0, ROOT)
    import bench
    argv = bench.rank_launch_argv(8, 29511, ["--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in argv and argv[argv.index("--nproc-per-node") + 1] == "8"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert os.path.basename(argv[-7]) == "bench.py"


def test_gpus_2_spawns_two_ranks_and_reduces():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--sequences", "7",
                          "--dry-run"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # one line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["dry_run"] is True and "INVALID" in r
    assert r["config"]["sequences_per_gpu"] == 7 and r["config"]["sweeps_per_step"] == 14
    assert r["config"]["first_sequences_rank0"] == [0, 2, 4, 6]  # dist.shard_sequences: global sequence q on rank q % world
    # value = scans of all ranks / max-over-ranks seconds: 2 x 7 x 5 scans in ~5 ms of stub steps
    assert abs(r["value"] - 70.0 / (r["ms_per_step"] * 5 / 1e3)) < 1e-6 * r["value"]


def test_world_size_mismatch_is_refused():
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_single_rank_dry_run_needs_no_launcher():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--sequences", "5", "--dry-run"], cwd=ROOT, env=_env(),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["config"]["sequences_per_gpu"] == 5


def test_eight_rank_preflight_dry_run():
    """The shape of the round-end scaling run (BASELINE configs[3]: 8 ranks, one per GPU): launcher -> 8 gloo ranks -> sharding ->
    barrier / max-over-ranks timing -> SUM / MAX reduction -> one line from rank 0, on a CPU-only host."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--sequences", "3",
                          "--dry-run"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["config"]["sequences_per_gpu"] == 3 and r["config"]["sweeps_per_step"] == 24
    assert r["config"]["first_sequences_rank0"] == [0, 8, 16]
    assert abs(r["value"] - 8 * 3 * 3 / (r["ms_per_step"] * 3 / 1e3)) < 1e-6 * r["value"]
