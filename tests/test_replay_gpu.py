"""Recorded-sequence replay (rosbag v2.0 / Oxford PNG -> device odometry -> KITTI trajectory files) against the oracle fuser."""
import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import kitti, readers, replay, synth

pytestmark = pytest.mark.gpu
RR = np.float32(0.0595238)


def test_bag_replay_matches_oracle_and_writes_kitti_files(oracle, tmp_path):
    imgs, gt = synth.world_sequence(6, seed=51)
    w = readers.BagWriter(tmp_path / "seq.bag", compression="bz2")
    for i in range(6):
        t = 1547131046000000000 + i * 250000000
        w.write("/gt", "nav_msgs/Odometry", t, readers.encode_odometry(gt[i] + [10.0, -3.0, 0.0], t, seq=i))  # arbitrary world offset
        w.write("/Navtech/Polar", "sensor_msgs/Image", t + 500, readers.encode_image(imgs[i], t + 500, seq=i))
    w.close()
    out = replay.main(["--bag", str(tmp_path / "seq.bag"), "--est_directory", str(tmp_path / "est"), "--range-res", "0.0595238", "--res", "3.0",
                       "--z-min", "60", "--submap_scan_size", "4", "--weight_option", "4"])
    assert out["frames"] == 6 and "drift" in out
    est = kitti.read_kitti(tmp_path / "est" / "est_00.txt")
    gtk = kitti.read_kitti(tmp_path / "est" / "gt_00.txt")
    assert est.shape == (6, 4, 4) and gtk.shape == (6, 4, 4)
    assert np.allclose(gtk[0], np.eye(4), atol=1e-6)  # relative to the first ground-truth pose
    fu = oracle.Fuser(oracle.default_params(range_res=RR, z_min=60.0, res=3.0, submap_scan_size=4, weight_opt=4, weight_intensity=1, compensate=1,
                                            radar_ccw=0, cost=1, loss=1))
    for t in range(6):
        exp = fu.process_polar(imgs[t])
        got = np.array([est[t, 0, 3], est[t, 1, 3], np.arctan2(est[t, 1, 0], est[t, 0, 0])])
        assert np.all(np.abs(got[:2] - exp[:2]) < 1e-4 + 5e-7) and abs(got[2] - exp[2]) < 1e-5 + 2e-6
    assert np.linalg.norm(est[-1, :2, 3] - gtk[-1, :2, 3]) < 0.5  # known answer: follows the synthetic ground truth


def test_oxford_png_directory_replay(oracle, tmp_path):
    rr = np.float32(0.0438)
    imgs, gt = synth.world_sequence(3, A=400, R=3768, range_res=rr, seed=52)
    d = tmp_path / "radar"
    d.mkdir()
    for i in range(3):
        ts = 1547131046353776 + i * 250000
        readers.write_png_gray8(d / ("%d.png" % ts), readers.oxford_png_rows(imgs[i], ts + np.arange(400) * 625), filter_type=2)
    out = replay.main(["--oxford_png_dir", str(d), "--est_directory", str(tmp_path / "est"), "--z-min", "60", "--res", "3.0"])
    assert out["frames"] == 3 and "drift" not in out
    fu = oracle.Fuser(oracle.default_params(range_res=rr, z_min=60.0, res=3.0, submap_scan_size=3, weight_opt=0, weight_intensity=1, compensate=1,
                                            radar_ccw=0, cost=1, loss=1))
    for t in range(3):
        exp = fu.process_polar(imgs[t])
    assert np.all(np.abs(np.array(out["final_pose"]) - exp) < [1e-4, 1e-4, 1e-5])


def test_bench_two_ranks_through_the_launcher_on_one_gpu():
    """`python bench.py --gpus 2` end to end with real device work: the launcher starts two ranks, each generates its own streams,
    runs its resident sequences through the HIP path and the throughput is reduced (SUM scans, MAX seconds) into one JSON line
    from rank 0. No second GPU in the test box: --share-gpu puts both ranks on cuda:0 and the reduction on gloo (the line
    says so); the 8-GPU run differs in the device index and the RCCL backend only."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "6", "--warmup", "2", "--sequences", "64",
                          "--unique", "2", "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["sweeps_per_step"] == 128 and "INVALID" in r
    assert len(r["per_rank_scans_per_s"]) == 2 and all(v > 0 for v in r["per_rank_scans_per_s"])
    assert abs(r["value"] - 2 * 64 * 6 / (r["ms_per_step"] * 6 / 1e3)) < 1e-6 * r["value"]
    assert r["state"]["replicas_bit_identical"] is True


def test_bench_eight_ranks_through_the_launcher_on_one_gpu():
    """BASELINE configs[3]'s shape - eight ranks, one sequence shard each - with real device work: `python bench.py --gpus 8` starts
    the eight ranks itself; --share-gpu puts them all on cuda:0 and the reduction on gloo. The day an 8-GPU node runs it the only
    new variables are the device index and the RCCL backend. The line must carry the per-rank rates and roofline fractions."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--share-gpu", "--steps", "4", "--warmup", "2", "--sequences", "256",
                          "--unique", "8", "--repeats", "2", "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["config"]["sequences_per_gpu"] == 256 and r["config"]["sweeps_per_step"] == 8 * 256 and "INVALID" in r
    assert r["scaling"] == "weak" and r["steps"] == 4
    assert len(r["per_rank_scans_per_s"]) == 8 and all(v > 0 for v in r["per_rank_scans_per_s"])
    assert len(r["per_rank_filter_frac_of_hbm_peak"]) == 8 and all(0 < v < 1 for v in r["per_rank_filter_frac_of_hbm_peak"])
    # value = the units all ranks processed / the slowest rank's time
    assert abs(r["value"] - 8 * 256 * 4 / (r["ms_per_step"] * 4 / 1e3)) < 1e-6 * r["value"]
    assert r["value"] <= sum(r["per_rank_scans_per_s"]) * (1 + 1e-9)
    assert r["state"]["replicas_bit_identical"] is True and r["config"]["keyframes_at_first_timed_step_min"] == 4


@pytest.mark.parametrize("B,persistent_max", [(1, 256), (1, 0), (3, 256), (3, 0)])
def test_replay_host_equals_step_host(B, persistent_max):
    """cfear_odometry_replay_host (chunks copied and filtered ahead on a second stream; a persistent workgroup per sequence or the
    two launches per sweep of the batched step) gives bit-identical poses, iteration counts and cell counts to one
    cfear_odometry_step_host per sweep, for one sequence and for several in lockstep, in one call or in pieces, and its state
    carries over to the per-sweep entry points."""
    from cfear_radarodometry_code_public_amd import capi, synth
    RR = np.float32(0.0595238)
    T = 70  # more than one chunk of 64 sweeps
    imgs, _ = synth.world_sequence(T // 2, seed=31, world_seed=55)
    imgs = np.concatenate([imgs, imgs[::-1]])  # forwards, then backwards: a consistent trajectory twice as long
    frames = np.stack([np.roll(imgs, q, axis=1) if q else imgs for q in range(B)], axis=1)  # [T, B, A, R]: sequence q sees the world rotated
    p = capi.default_params(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, submap_scan_size=4)
    ctx = capi.Context(p, 400, 3360)
    ctx.tune(capi.TUNE_REPLAY_PERSISTENT_MAX, persistent_max)
    a, b, c = ctx.odometry(B), ctx.odometry(B), ctx.odometry(B)
    exp = []
    for t in range(T):
        a.step_host(frames[t])
        P = a.poses()
        exp.append([(P[q].copy(),) + tuple(int(v) for v in (a.summary(q)[0].outer_iterations, a.summary(q)[0].num_residuals, a.summary(q)[2], a.summary(q)[1])) for q in range(B)])
    pinned = ctx.pinned(frames.shape)
    pinned[:] = frames
    rec = b.replay_host(pinned)  # one call
    rec2 = np.concatenate([c.replay_host(frames[:5]), c.replay_host(frames[5:6]), c.replay_host(pinned[6:T - 1])])  # pieces, pageable and pinned
    c.step_host(frames[T - 1])  # the state carries over
    for t in range(T):
        for q in range(B):
            e = exp[t][q]
            for r in ([rec[t, q]] + ([rec2[t, q]] if t < T - 1 else [])):
                assert np.array_equal(r["pose"], e[0]), (t, q)
                assert (int(r["outer_iterations"]), int(r["num_residuals"]), int(r["n_keyframes"]), int(r["n_cells"])) == e[1:], (t, q)
    assert np.array_equal(c.poses(), np.array([exp[T - 1][q][0] for q in range(B)]))
    assert np.array_equal(b.poses(), c.poses())
    ctx.close()


def test_replay_device_is_asynchronous_and_equals_replay_host():
    """cfear_odometry_replay_device: the same records from device-resident sweeps, written into a device buffer, nothing waited for
    inside the call (the records are read after a synchronisation of the context)."""
    import torch
    from cfear_radarodometry_code_public_amd import capi, synth
    RR = np.float32(0.0595238)
    T, B = 70, 2
    imgs, _ = synth.world_sequence(T // 2, seed=33, world_seed=56)
    imgs = np.concatenate([imgs, imgs[::-1]])
    frames = np.stack([imgs, np.roll(imgs, 5, axis=1)], axis=1)  # [T, B, A, R]
    p = capi.default_params(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, submap_scan_size=4)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = capi.Context(p, 400, 3360, stream=st.cuda_stream)
        a, b = ctx.odometry(B), ctx.odometry(B)
        rec_h = a.replay_host(frames)
        d_frames = torch.from_numpy(frames).cuda()
        d_rec = torch.zeros((T, B, capi.SWEEP_RECORD_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        b.replay_device(d_frames, T, d_rec)
        d_frames.zero_()  # queued behind the last filter: must not disturb the replay
        ctx.synchronize(); torch.cuda.synchronize()
        rec_d = d_rec.cpu().numpy().view(capi.SWEEP_RECORD_DTYPE).reshape(T, B)
        for f in ("pose", "final_cost", "outer_iterations", "num_residuals", "n_keyframes", "n_cells", "inner_iterations"):
            assert np.array_equal(rec_d[f], rec_h[f]), f
        assert np.array_equal(a.poses(), b.poses())
        ctx.close()


def test_rccl_backend_runs_the_throughput_reduction_on_the_device():
    """The collective of the N-GPU run - {scans: SUM, seconds: MAX} as two 8-byte all-reduces of float64 device tensors over the
    "nccl" backend (= RCCL on ROCm) - on the one GPU of the test box: a process group of world size 1, the same
    init_process_group(device_id=...) call, the same dist.reduce_throughput and the all_gather of bench.py's per-rank figures.
    Everything of the 8-GPU run that does not need a second device."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, torch, torch.distributed as dist\n"
        "from cfear_radarodometry_code_public_amd.dist import reduce_throughput, shard_sequences\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "dev = torch.device('cuda', 0)\n"
        "t = torch.tensor([2.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)\n"
        "s = torch.tensor([7.0], dtype=torch.float64, device=dev); dist.all_reduce(s, op=dist.ReduceOp.SUM)\n"
        "g = [torch.zeros(2, dtype=torch.float64, device=dev)]; dist.all_gather(g, torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev))\n"
        "dist.barrier(); torch.cuda.synchronize()\n"
        "assert float(t.item()) == 2.5 and float(s.item()) == 7.0 and g[0].tolist() == [1.0, 2.0]\n"
        "assert shard_sequences(5, 0, 1) == [0, 1, 2, 3, 4]\n"
        "print('backend', dist.get_backend(), 'ok')\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "backend nccl ok" in out.stdout
