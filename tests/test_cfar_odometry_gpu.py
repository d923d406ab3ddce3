"""CA-CFAR as the stage-1 filter of the device fuser (radarDriver::Process with filter_type "CA-CFAR", radar_driver.cpp:52-56, in
front of OdometryKeyframeFuser::pointcloudCallback): cfear_params.filter_type = CFEAR_FILTER_CACFAR makes cfear_odometry_step_* /
cfear_odometry_replay_* run AzimuthCACFAR -> Compensate -> MapPointNormal -> Register on the device; cfear_odometry_step_cloud_device
takes any clouds. Checked against the oracle's fuser fed with the oracle's CA-CFAR clouds (cfo_cfar, cfar.cpp:27-87), every sweep:
keyframe / iteration / residual / cell counts and the pose (1e-4 m, 1e-5 rad).

Settings: launch/oxford/eval/params/kstrong_vs_cfar/oxford-cfear-3-ca-cfar:13-30 - P2P, four keyframes, res 3, Huber 0.1, unweighted,
z_min 20 (static threshold), 10 guard cells, window 40 (first of its sweep), false-alarm rate 0.01."""
import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import capi, synth

pytestmark = pytest.mark.gpu

A, R, RR = 400, 3360, np.float32(0.0595238)
CFAR = dict(window_size=40, nb_guard_cells=10, false_alarm_rate=0.01)
PRESET = dict(range_res=RR, z_min=20.0, cost=0, submap_scan_size=4, res=3.0, loss=1, loss_limit=0.1, weight_intensity=0, weight_opt=0, compensate=1,
              radar_ccw=0, regularization=1.0, covar_scale=1.0)


def _frames(T, B, kind, seed0=0):
    frames = np.empty((T, B, A, R), dtype=np.uint8)
    for q in range(B):
        for t0, chunk in synth.drive_chunks(T, kind, 30 + seed0 + q, 40 + seed0 + q, A, R, RR, ccw=False):
            frames[t0:t0 + len(chunk), q] = chunk
    return frames


def _hip_params(kw, max_points=0, **cfar):
    c = dict(CFAR, **cfar)
    return capi.default_params(filter_type=capi.FILTER_CACFAR, cfar_window_size=c["window_size"], cfar_nb_guard_cells=c["nb_guard_cells"],
                               cfar_false_alarm_rate=c["false_alarm_rate"], cfar_max_points=max_points, **kw)


def _expect(oracle, fu, img, kw, cfar, prefix=False):
    cloud = oracle.cfar(img, float(np.float32(kw["range_res"])), float(kw["z_min"]), 2.5, prefix=prefix, **cfar)
    pose = fu.process_cloud(cloud)
    S = fu.last_summary()
    no = max(int(S.outer_iterations), 0)
    return pose, (int(S.outer_iterations), [int(v) for v in S.inner_iterations[:min(no, 8)]], int(S.num_residuals), int(fu.num_keyframes), len(fu.last_cells())), len(cloud)


def _rec_tuple(r):
    no = max(int(r["outer_iterations"]), 0)
    return (int(r["outer_iterations"]), [int(v) for v in r["inner_iterations"][:min(no, 8)]], int(r["num_residuals"]), int(r["n_keyframes"]), int(r["n_cells"]))


@pytest.mark.parametrize("route,kind,extra", [
    ("step", "blocks", {}), ("replay_batched", "canyon", {}), ("replay_persistent", "blocks", {}),
    ("replay_persistent", "canyon", dict(cost=1, submap_scan_size=1)),     # CFEAR-1's registration behind the detector
    ("step", "canyon", dict(cost=2, regularization=0.1, compensate=0)),    # P2D, compensation off
])
def test_cfar_fuser_matches_oracle_at_every_sweep(oracle, route, kind, extra):
    T, B = 45, 3  # (the oracle's literal CA-CFAR - a window sum per bin - is what takes the time here: ~70 ms per sweep)
    kw = dict(PRESET, **extra)
    frames = _frames(T, B, kind)
    fus = [oracle.Fuser(oracle.default_params(**kw)) for _ in range(B)]
    ctx = capi.Context(_hip_params(kw), A, R)
    ctx.tune(capi.TUNE_REPLAY_PERSISTENT_MAX, 256 if route == "replay_persistent" else 0)
    odo = ctx.odometry(B)
    recs = odo.replay_host(frames) if route != "step" else None
    npts = []
    for t in range(T):
        if route == "step":
            odo.step_host(frames[t])
            got = odo.poses()
        for q in range(B):
            exp, e, n = _expect(oracle, fus[q], frames[t, q], kw, CFAR)
            npts.append(n)
            if route == "step":
                S, nc, nk = odo.summary(q)
                no = max(int(S.outer_iterations), 0)
                g = (int(S.outer_iterations), [int(v) for v in S.inner_iterations[:min(no, 8)]], int(S.num_residuals), nk, nc)
                pose = got[q]
            else:
                g, pose = _rec_tuple(recs[t, q]), recs[t, q]["pose"]
            if t > 0:
                assert g == e, (t, q, g, e)
            assert np.all(np.abs(pose[:2] - exp[:2]) < 1e-4) and abs(pose[2] - exp[2]) < 1e-5, (t, q, pose, exp)
    assert min(npts) > 500 and e[3] == kw["submap_scan_size"] and e[2] > 50, (min(npts), max(npts), e)
    odo.release()
    ctx.close()


@pytest.mark.parametrize("window,pfa,route", [(500, 0.0001, "replay_persistent"), (150, 0.001, "replay_batched"), (500, 0.0001, "replay_batched")])
def test_cfar_fuser_long_windows_of_the_reference_sweep(oracle, window, pfa, route):
    """the far corner and the middle of the reference's CA-CFAR grid (launch/oxford/eval/params/kstrong_vs_cfar/oxford-cfear-3-ca-cfar:31-32: window 40 ... 500
    x false-alarm rate 0.1 ... 0.0001, 10 guard cells) through cfear_odometry_replay_host, 60 sweeps, against the oracle's fuser at every sweep. The oracle's
    detector here is its prefix-sum twin (cfo_cfar_prefix; the literal window loop takes a second per sweep at window 500) - checked against the literal one on
    the first sweep of every sequence, and bin for bin in tests/test_cfar_cpu.py"""
    T, B = 60, 2
    cfar = dict(window_size=window, nb_guard_cells=10, false_alarm_rate=pfa)
    kw = dict(PRESET)
    frames = _frames(T, B, "blocks", seed0=window)
    for q in range(B):
        rr, zz = float(np.float32(kw["range_res"])), float(kw["z_min"])
        assert np.array_equal(oracle.cfar(frames[0, q], rr, zz, 2.5, **cfar), oracle.cfar(frames[0, q], rr, zz, 2.5, prefix=True, **cfar))
    fus = [oracle.Fuser(oracle.default_params(**kw)) for _ in range(B)]
    ctx = capi.Context(_hip_params(kw, **cfar), A, R)
    ctx.tune(capi.TUNE_REPLAY_PERSISTENT_MAX, 256 if route == "replay_persistent" else 0)
    odo = ctx.odometry(B)
    recs = odo.replay_host(frames)
    npts = []
    for t in range(T):
        for q in range(B):
            exp, e, n = _expect(oracle, fus[q], frames[t, q], kw, cfar, prefix=True)
            npts.append(n)
            g, pose = _rec_tuple(recs[t, q]), recs[t, q]["pose"]
            if t > 0:
                assert g == e, (t, q, g, e)
            assert np.all(np.abs(pose[:2] - exp[:2]) < 1e-4) and abs(pose[2] - exp[2]) < 1e-5, (t, q, pose, exp)
    assert min(npts) > 300 and e[3] == kw["submap_scan_size"] and e[2] > 30, (min(npts), max(npts), e)
    odo.release()
    ctx.close()


def test_cfar_fuser_large_clouds_take_the_general_feature_path(oracle):
    """a low static threshold and a high false-alarm rate: > 4864 detections per sweep (beyond the compact feature path), still the
    oracle's result at every sweep; and the step from caller-made clouds (cfear_odometry_step_cloud_device) agrees with the built-in
    filter bit for bit"""
    import torch
    T, B = 30, 2
    cfar = dict(window_size=40, nb_guard_cells=10, false_alarm_rate=0.1)
    kw = dict(PRESET, z_min=10.0)
    frames = _frames(T, B, "canyon", seed0=5)
    fus = [oracle.Fuser(oracle.default_params(**kw)) for _ in range(B)]
    ctx = capi.Context(_hip_params(kw, max_points=60000, **cfar), A, R)
    odo, odo2 = ctx.odometry(B), ctx.odometry(B)
    dev = torch.device("cuda:0")
    d_xyi = torch.empty((B, 60000, 3), dtype=torch.float32, device=dev)
    d_n = torch.empty((B,), dtype=torch.int32, device=dev)
    nmax = 0
    for t in range(T):
        odo.step_host(frames[t])
        got = odo.poses()
        d_img = torch.from_numpy(frames[t]).to(dev)
        torch.cuda.synchronize()
        ctx.filter_cfar_batch(d_img.data_ptr(), B, d_xyi.data_ptr(), 60000, d_n.data_ptr(), **cfar)
        odo2.step_cloud_device(d_xyi.data_ptr(), 60000, d_n.data_ptr())
        assert np.array_equal(odo2.poses(), got)
        for q in range(B):
            exp, e, n = _expect(oracle, fus[q], frames[t, q], kw, cfar)
            nmax = max(nmax, n)
            S, nc, nk = odo.summary(q)
            no = max(int(S.outer_iterations), 0)
            g = (int(S.outer_iterations), [int(v) for v in S.inner_iterations[:min(no, 8)]], int(S.num_residuals), nk, nc)
            if t > 0:
                assert g == e, (t, q, g, e)
            assert np.all(np.abs(got[q][:2] - exp[:2]) < 1e-4) and abs(got[q][2] - exp[2]) < 1e-5, (t, q, got[q], exp)
    assert nmax > 4864, nmax
    odo.release(); odo2.release()
    ctx.close()


def test_cfar_fuser_point_capacity_is_loud(oracle):
    """more detections than cfar_max_points: the reading calls fail with CFEAR_ERR_CAPACITY and say which limit"""
    frames = _frames(3, 1, "blocks")
    ctx = capi.Context(_hip_params(PRESET, max_points=300), A, R)
    odo = ctx.odometry(1)
    for t in range(3):
        odo.step_host(frames[t])
    with pytest.raises(capi.CfearError, match=r"rc=-6.*more points than the 300.*cfar_max_points"):
        odo.poses()
    odo.release()
    ctx.close()
    # and k-strongest objects refuse a changed filter type
    ctx = capi.Context(capi.default_params(**PRESET), A, R)
    odo = ctx.odometry(1)
    ctx.set_params(_hip_params(PRESET))
    with pytest.raises(capi.CfearError, match="filter_type changed"):
        odo.step_host(frames[0])
    odo.release()
    ctx.close()


def test_cloud_step_on_a_kstrongest_object_matches_the_oracle(oracle):
    """cfear_odometry_step_cloud_device is not tied to the CA-CFAR filter: an object created for k-strongest takes clouds of up to A * k points
    from any producer - here the oracle's own k-strongest clouds, uploaded - and follows the oracle's fuser fed with the same clouds"""
    import torch
    T, B, k = 40, 2, 12
    kw = dict(PRESET, z_min=60.0, cost=1, weight_intensity=1, weight_opt=4, k_strongest=k)
    frames = _frames(T, B, "blocks", seed0=9)
    fus = [oracle.Fuser(oracle.default_params(**kw)) for _ in range(B)]
    ctx = capi.Context(capi.default_params(**kw), A, R)
    odo = ctx.odometry(B)
    dev = torch.device("cuda:0")
    cap = A * k
    for t in range(T):
        clouds = [oracle.cloud(oracle.filter_polar(frames[t, q], 60, k), float(RR), 2.5) for q in range(B)]
        h = np.zeros((B, cap, 3), dtype=np.float32)
        for q, c in enumerate(clouds):
            h[q, :len(c)] = c
        d_xyi = torch.from_numpy(h).to(dev)
        d_n = torch.tensor([len(c) for c in clouds], dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        odo.step_cloud_device(d_xyi.data_ptr(), cap, d_n.data_ptr())
        got = odo.poses()
        for q in range(B):
            exp = fus[q].process_cloud(clouds[q])
            S = fus[q].last_summary()
            Sg, nc, nk = odo.summary(q)
            if t > 0:
                assert (int(Sg.outer_iterations), int(Sg.num_residuals), nk, nc) == (int(S.outer_iterations), int(S.num_residuals), int(fus[q].num_keyframes), len(fus[q].last_cells())), (t, q)
            assert np.all(np.abs(got[q][:2] - exp[:2]) < 1e-4) and abs(got[q][2] - exp[2]) < 1e-5, (t, q, got[q], exp)
    with pytest.raises(capi.CfearError, match="capacity .* exceeds"):
        odo.step_cloud_device(d_xyi.data_ptr(), cap + 1, d_n.data_ptr())
    odo.release()
    ctx.close()
