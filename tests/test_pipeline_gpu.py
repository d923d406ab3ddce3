"""GPU parity of stages 1.5-3 (through the C ABI) against the CPU oracle.

Tolerances (BASELINE.json north_star): k-strongest bit-exact (test_kstrongest_gpu.py); clouds
bit-exact up to 1 float ulp on compensated points; cell statistics 1e-9; SE(2) poses within
1e-4 m / 1e-5 rad at equal outer/inner iteration counts."""
import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import capi, synth

pytestmark = pytest.mark.gpu

RR = np.float32(0.0595238)
POS_TOL, ROT_TOL = 1e-4, 1e-5


def mk_params(mod, **kw):
    base = dict(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4,
                submap_scan_size=4, compensate=1, radar_ccw=0)
    base.update(kw)
    return mod.default_params(**base)


@pytest.fixture(scope="module")
def seq():
    imgs, gt = synth.world_sequence(14, seed=3)
    return imgs, gt


def test_param_struct_layouts_match(oracle):
    import ctypes as C
    assert C.sizeof(capi.Params) == C.sizeof(oracle.Params) + 24  # + the CA-CFAR knobs of the batched objects (tests/test_abi_cpu.py)
    assert C.sizeof(capi.Cell) == C.sizeof(oracle.Cell)
    assert C.sizeof(capi.RegSummary) == C.sizeof(oracle.RegSummary)


def test_cloud_matches_oracle(oracle, seq):
    imgs, _ = seq
    p = mk_params(capi)
    ctx = capi.Context(p, 400, 3360)
    for t in (0, 5):
        c, cp = ctx.filter_polar(imgs[t])
        slots = oracle.filter_polar(imgs[t], 60, 12)
        exp = oracle.cloud(slots, RR, 2.5, peaks=False)
        expp = oracle.cloud(slots, RR, 2.5, peaks=True)
        got, gotp = c.download(), cp.download()
        assert got.shape == exp.shape and gotp.shape == expp.shape
        assert np.array_equal(got, exp)  # host-libm trig table -> bit exact
        assert np.array_equal(gotp, expp)
        mot = [0.93, -0.04, 0.021]
        for ccw in (0, 1):
            c2 = ctx.cloud_upload(exp)
            ctx.compensate(c2, mot, ccw)
            g2 = c2.download()
            e2 = oracle.compensate(exp, mot, ccw)
            ulp = np.spacing(np.abs(e2[:, :2]).astype(np.float32))
            assert np.all(np.abs(g2[:, :2] - e2[:, :2]) <= ulp)
            assert np.mean(g2[:, :2] != e2[:, :2]) < 1e-3
            # both clouds of the sweep in one launch (cfear_compensate_pair): what two single calls give, bit for bit - through the clouds' host mirrors
            # (the filter's own clouds) and through the copy route (an uploaded one)
            a, b = ctx.filter_polar(imgs[t])
            a1, b1 = ctx.filter_polar(imgs[t])
            ctx.compensate_pair(a, b, mot, ccw)
            ctx.compensate(a1, mot, ccw); ctx.compensate(b1, mot, ccw)
            assert np.array_equal(a.download(), a1.download()) and np.array_equal(b.download(), b1.download())
            assert np.array_equal(a.download(), g2)
            u = ctx.cloud_upload(expp)
            ctx.compensate_pair(c2, u, mot, ccw)  # (c2 a second time: the motion applies again)
            ctx.compensate(a1, mot, ccw)
            assert np.array_equal(c2.download(), a1.download()) and np.array_equal(u.download(), b1.download())
    ctx.close()


def cells_close(a, b, tol=1e-9):
    assert len(a) == len(b), (len(a), len(b))
    for f in ("mean", "cov", "normal", "lambda_min", "lambda_max", "scale", "sum_intensity", "avg_intensity"):
        assert np.allclose(a[f], b[f], rtol=tol, atol=tol), f
    assert np.array_equal(a["nsamples"], b["nsamples"])


@pytest.mark.parametrize("res,wi,df", [(3.0, 1, 1.0), (3.5, 0, 1.0), (3.0, 1, 2.0)])
def test_features_match_oracle(oracle, seq, res, wi, df):
    imgs, _ = seq
    po = mk_params(oracle, res=res, weight_intensity=wi, downsample_factor=df)
    pg = mk_params(capi, res=res, weight_intensity=wi, downsample_factor=df)
    ctx = capi.Context(pg, 400, 3360)
    for t in (1, 7):
        slots = oracle.filter_polar(imgs[t], 60, 12)
        xyi = oracle.compensate(oracle.cloud(slots, RR, 2.5), [1.0, 0.01, 0.02], 0)
        so = oracle.Scan(xyi, po)
        sg = ctx.scan_create(ctx.cloud_upload(xyi))
        co, cg = so.cells(), sg.cells()
        assert len(co) > 50
        cells_close(cg, co)
        # GetClosestIdx
        rng = np.random.default_rng(t)
        q = co["mean"][rng.integers(0, len(co), 400)] + rng.normal(0, 1.5, (400, 2))
        for d in (2.0, 4.0):
            got = sg.closest(q, d)
            exp = np.array([so.closest(x, y, d) for x, y in q])
            assert np.array_equal(got, exp)
    ctx.close()


@pytest.mark.parametrize("case", ["fractional_intensities", "more_points_than_compact", "huge_voxel_grid", "small_leaf", "many_voxels"])
def test_general_feature_path_matches_oracle(oracle, seq, case):
    """Clouds outside the limits of the compact (two workgroups per compute unit) feature path take the general path in
    global arrays: intensities that are not integers 0..255, more than 4864 points, a voxel grid beyond 32768 voxels. Same
    results as the oracle either way; "small_leaf" stays on the compact path with more than three voxel rows per radius;
    "many_voxels" stays on it with more occupied voxels (> 2431) than its per-wave sort counters hold, so that the sort runs on
    one counter set and ranks the members of a voxel (up to ~100 of them here) by counting - the path a block also takes if
    it ever finds its ordered scatter out of order."""
    imgs, _ = seq
    res, df = 3.0, 1.0
    slots = oracle.filter_polar(imgs[2], 60, 12)
    xyi = oracle.compensate(oracle.cloud(slots, RR, 2.5), [1.0, 0.01, 0.02], 0)
    if case == "fractional_intensities":
        xyi = xyi.copy(); xyi[:, 2] += np.float32(0.25)
    elif case == "more_points_than_compact":
        xyi2 = oracle.compensate(oracle.cloud(oracle.filter_polar(imgs[3], 60, 12), RR, 2.5), [0.5, 0.0, 0.01], 0)
        xyi = np.concatenate([xyi, xyi2 + np.float32([0.37, -0.21, 0])])
        assert len(xyi) > 4864
    elif case == "huge_voxel_grid":
        xyi = xyi.copy(); xyi[0, :2] = [-900.0, -700.0]; xyi[1, :2] = [800.0, 650.0]  # 1700 m x 1350 m / 3 m = 255 k voxels
    elif case == "small_leaf":
        df = 2.5  # leaf 1.2 m: five voxel rows inside the radius
    elif case == "many_voxels":
        rng = np.random.default_rng(11)
        keep = xyi[rng.permutation(len(xyi))[:2100]]  # the scan's own clusters (cells come from these)
        gx, gy = np.meshgrid(np.arange(-26, 26), np.arange(-25, 25))  # 2600 lone points, one per voxel of a 3 m lattice
        lone = np.stack([gx.ravel() * 3.0 + 1.1 + rng.uniform(-0.9, 0.9, gx.size), gy.ravel() * 3.0 + 1.3 + rng.uniform(-0.9, 0.9, gx.size),
                         rng.integers(61, 200, gx.size)], axis=1).astype(np.float32)
        xyi = np.concatenate([keep, lone])[rng.permutation(2100 + gx.size)]
        assert len(xyi) <= 4864
    po = mk_params(oracle, res=res, weight_intensity=1, downsample_factor=df)
    pg = mk_params(capi, res=res, weight_intensity=1, downsample_factor=df)
    ctx = capi.Context(pg, 400, 3360)
    so = oracle.Scan(xyi, po)
    sg = ctx.scan_create(ctx.cloud_upload(xyi))
    co, cg = so.cells(), sg.cells()
    assert len(co) > 50
    cells_close(cg, co)
    rng = np.random.default_rng(5)
    q = co["mean"][rng.integers(0, len(co), 200)] + rng.normal(0, 1.5, (200, 2))
    assert np.array_equal(sg.closest(q, 2.0), np.array([so.closest(x, y, 2.0) for x, y in q]))
    ctx.close()


def test_empty_cloud_fails_loudly(hip_lib):
    ctx = capi.Context(mk_params(capi), 400, 3360)
    with pytest.raises(capi.CfearError):
        ctx.scan_create(ctx.cloud_upload(np.zeros((0, 3), dtype=np.float32)))
    ctx.close()


@pytest.mark.parametrize("cost,loss,wopt", [(1, 1, 4), (2, 1, 4), (0, 1, 4), (1, 2, 0), (1, 0, 2), (1, 3, 1), (2, 5, 3), (1, 4, 4)])
def test_register_matches_oracle(oracle, seq, cost, loss, wopt):
    imgs, gt = seq
    kw = dict(cost=cost, loss=loss, weight_opt=wopt, regularization=0.1, covar_scale=1.0)
    po, pg = mk_params(oracle, **kw), mk_params(capi, **kw)
    ctx = capi.Context(pg, 400, 3360)
    frames = [0, 2, 4, 6, 7]
    so, sg = [], []
    for t in frames:
        xyi = oracle.cloud(oracle.filter_polar(imgs[t], 60, 12), RR, 2.5)
        so.append(oracle.Scan(xyi, po))
        sg.append(ctx.scan_create(ctx.cloud_upload(xyi)))
    poses = gt[frames].copy()
    poses[-1] = gt[6] + np.array([0.3, -0.2, 0.01])  # perturbed guess for the current scan
    for n in (2, 5):
        reto, Po, covo, So = oracle.register(so[-n:], poses[-n:], po)
        retg, Pg, covg, Sg = ctx.register(sg[-n:], poses[-n:])
        assert So.outer_iterations == Sg.outer_iterations
        assert list(So.inner_iterations[:8]) == list(Sg.inner_iterations[:8])
        assert So.num_residuals == Sg.num_residuals
        assert bool(reto) == retg
        assert np.all(np.abs(Pg[:, :2] - Po[:, :2]) < POS_TOL)
        assert np.all(np.abs(Pg[:, 2] - Po[:, 2]) < ROT_TOL)
        assert np.allclose(Sg.final_cost, So.final_cost, rtol=1e-9)
        assert np.allclose(covg, covo, rtol=1e-6, atol=1e-12)
        # recovered pose close to ground truth as a known-answer check
        if cost != 0 and loss in (1, 2):
            assert np.linalg.norm(Pg[-1, :2] - gt[7, :2]) < 0.25
    ctx.close()


@pytest.mark.parametrize("cost", [1, 2])
def test_batched_odometry_matches_oracle_fuser(oracle, seq, cost):
    imgs, gt = seq
    kw = dict(cost=cost)
    po, pg = mk_params(oracle, **kw), mk_params(capi, **kw)
    B = 3
    imgs2, _ = synth.world_sequence(14, seed=5, t0=30)
    streams = [imgs, imgs2, imgs[:, ::-1].copy()]  # three different sequences
    fus = [oracle.Fuser(po) for _ in range(B)]
    ctx = capi.Context(pg, 400, 3360)
    odo = ctx.odometry(B)
    for t in range(imgs.shape[0]):
        batch = np.stack([s[t] for s in streams])
        odo.step_host(batch)
        got = odo.poses()
        for q in range(B):
            exp = fus[q].process_polar(streams[q][t])
            S, nc, nk = odo.summary(q)
            So = fus[q].last_summary()
            assert nk == fus[q].num_keyframes
            assert nc == len(fus[q].last_cells())
            assert S.outer_iterations == So.outer_iterations, (t, q)
            assert list(S.inner_iterations[:8]) == list(So.inner_iterations[:8]), (t, q)
            assert np.all(np.abs(got[q, :2] - exp[:2]) < POS_TOL), (t, q, got[q], exp)
            assert abs(got[q, 2] - exp[2]) < ROT_TOL
    # known answer: the first stream follows the synthetic ground truth
    assert np.linalg.norm(got[0, :2] - gt[-1, :2]) < 0.5
    odo.release()
    ctx.close()


@pytest.mark.parametrize("k,cost,res,submap", [(40, 0, 3.0, 4), (40, 1, 3.0, 4), (12, 1, 3.5, 3), (25, 2, 3.0, 2)])
def test_reference_presets_through_device_fuser(oracle, k, cost, res, submap):
    """The reference's own presets: CFEAR-3 (P2P, k=40, r=3, s=4) runs the big-cloud
    path (400*40 points: sort keys and voxel lists in global memory instead of LDS)."""
    imgs, gt = synth.world_sequence(7, seed=9)
    kw = dict(k_strongest=k, cost=cost, res=res, submap_scan_size=submap, z_min=60.0)
    po, pg = mk_params(oracle, **kw), mk_params(capi, **kw)
    fu = oracle.Fuser(po)
    ctx = capi.Context(pg, 400, 3360)
    odo = ctx.odometry(1)
    for t in range(imgs.shape[0]):
        odo.step_host(imgs[t][None])
        got = odo.poses()[0]
        exp = fu.process_polar(imgs[t])
        S, nc, nk = odo.summary(0)
        So = fu.last_summary()
        assert nc == len(fu.last_cells()) and nk == fu.num_keyframes
        assert [S.outer_iterations] + list(S.inner_iterations[:8]) == [So.outer_iterations] + list(So.inner_iterations[:8]), t
        assert np.all(np.abs(got[:2] - exp[:2]) < POS_TOL) and abs(got[2] - exp[2]) < ROT_TOL, (t, got, exp)
    odo.release()
    ctx.close()


def test_filter_ahead_streams_give_identical_results(oracle):
    """The batched odometry can run the filter of a sweep on a low-priority stream of its own, one sweep ahead, and the features /
    registration kernels of n ranges of the sequences on n high-priority streams (double-buffered slots; cfear_tune
    ODOMETRY_OVERLAP = n). It must not change any result, for
    device-resident input (free-running streams) and for host input (staging buffer reuse), with reads in between."""
    import torch
    imgs, _ = synth.world_sequence(8, seed=13)
    imgs2, _ = synth.world_sequence(8, seed=14, t0=20)
    B = 6
    streams = [imgs, imgs2, imgs[:, ::-1].copy(), imgs2[:, ::-1].copy(), imgs, imgs2]
    batches = np.stack([np.stack([s[t] for s in streams]) for t in range(8)])  # [T][B][A][R]
    d_batches = torch.from_numpy(batches).cuda()
    torch.cuda.synchronize()
    pg = mk_params(capi)
    results = {}
    for overlap in (0, 1, 3):
        ctx = capi.Context(pg, 400, 3360)
        odo = ctx.odometry(B, overlap=overlap)
        mid = None
        for t in range(8):
            if t % 3 == 1:
                h = batches[t].copy()
                odo.step_host(h)
                h[:] = 0  # the host buffer belongs to the caller again when step_host returns
            else:
                odo.step_device(d_batches[t].data_ptr())
            if t == 4:
                mid = odo.poses()  # a reading call in the middle of the run joins the streams
        poses = odo.poses()
        summ = [odo.summary(q) for q in range(B)]
        results[overlap] = (poses, mid, [(s[0].outer_iterations, list(s[0].inner_iterations[:8]), s[1], s[2]) for s in summ])
        odo.release()
        ctx.close()
    for n in (1, 3):
        assert np.array_equal(results[0][0], results[n][0])
        assert np.array_equal(results[0][1], results[n][1])
        assert results[0][2] == results[n][2]
    assert np.all(np.isfinite(results[1][0])) and np.abs(results[1][0][:, :2]).max() > 1.0  # the sequences moved
    # against the oracle's fuser, sweep by sweep order
    po = mk_params(oracle)
    for q in (0, 3):
        fu = oracle.Fuser(po)
        for t in range(8):
            exp = fu.process_polar(streams[q][t])
        got = results[3][0][q]
        assert np.all(np.abs(got[:2] - exp[:2]) < POS_TOL) and abs(got[2] - exp[2]) < ROT_TOL, (q, got, exp)


def _fuser_parity(oracle, imgs, A, R, **kw):
    po, pg = mk_params(oracle, **kw), mk_params(capi, **kw)
    fu = oracle.Fuser(po)
    ctx = capi.Context(pg, A, R)
    odo = ctx.odometry(1)
    for t in range(imgs.shape[0]):
        odo.step_host(imgs[t][None])
        got = odo.poses()[0]
        exp = fu.process_polar(imgs[t])
        S, nc, nk = odo.summary(0)
        So = fu.last_summary()
        assert nk == fu.num_keyframes and nc == len(fu.last_cells())
        assert S.outer_iterations == So.outer_iterations and list(S.inner_iterations[:8]) == list(So.inner_iterations[:8]), t
        assert np.all(np.abs(got[:2] - exp[:2]) < POS_TOL) and abs(got[2] - exp[2]) < ROT_TOL, (t, got, exp)
    odo.release()
    ctx.close()
    return got


def test_baseline_config1_oxford_shape_pair(oracle):
    """BASELINE configs[0] / SURVEY 8(d) Config 1 with the synthetic-world substitute: a 400 x 3768 sweep pair at
    range_res 0.0438 (Oxford Navtech geometry), k=12, z_min=60, r=3.0, P2L + Huber 0.1; frame 1 registers against
    exactly one keyframe (odometrykeyframefuser.cpp:171-177)."""
    rr = np.float32(0.0438)
    imgs, gt = synth.world_sequence(2, A=400, R=3768, range_res=rr, seed=41)
    got = _fuser_parity(oracle, imgs, 400, 3768, range_res=rr, cost=1, loss=1, loss_limit=0.1, submap_scan_size=3, res=3.0)
    assert np.linalg.norm(got[:2] - gt[-1, :2]) < 0.5


def test_baseline_config3_p2d(oracle):
    """SURVEY 8(d) Config 3: the configs[1] stream with cost P2D, regularization 0.1, covar_scale 1 (params/baseline_p2d)."""
    imgs, gt = synth.world_sequence(6, seed=43, ccw=True)
    got = _fuser_parity(oracle, imgs, 400, 3360, cost=2, regularization=0.1, covar_scale=1.0, radar_ccw=1, min_keyframe_dist=1.5)
    assert np.linalg.norm(got[:2] - gt[-1, :2]) < 0.5


def test_blank_sweep_in_one_sequence_does_not_disturb_the_others(oracle):
    """A sequence whose sweep has no returns (the reference prints 'error, cloud empty' and exits, pointnormal.cpp:72-75) coasts on
    its constant-velocity guess (odometrykeyframefuser.cpp:164-168) and stays finite; its batch neighbours are unaffected; the
    sequence recovers on the next real sweep."""
    imgs, _ = synth.world_sequence(5, seed=71)
    blank = np.zeros_like(imgs[0])
    pg, po = mk_params(capi), mk_params(oracle)
    ctx = capi.Context(pg, 400, 3360)
    odo = ctx.odometry(2)
    fu = oracle.Fuser(po)
    for t in range(5):
        batch = np.stack([imgs[t], blank if t == 2 else imgs[t]])
        odo.step_host(batch)
        got = odo.poses()
        exp = fu.process_polar(imgs[t])
        assert np.all(np.abs(got[0, :2] - exp[:2]) < POS_TOL) and abs(got[0, 2] - exp[2]) < ROT_TOL, t
        assert np.all(np.isfinite(got[1]))
        if t == 2:  # no registration possible: the pose is the prediction from the previous motion
            assert np.linalg.norm(got[1, :2] - got[0, :2]) < 0.3 and abs(got[1, 2] - got[0, 2]) < 0.02
    assert np.linalg.norm(got[1, :2] - got[0, :2]) < 1.0  # back on track after the gap
    odo.release()
    ctx.close()


def test_reset_replays_identically_and_contexts_are_independent(oracle):
    imgs, _ = synth.world_sequence(4, seed=72)
    ctx_a = capi.Context(mk_params(capi), 400, 3360)
    ctx_b = capi.Context(mk_params(capi, cost=2, res=3.5, submap_scan_size=2), 400, 3360)  # different parameters side by side
    oa, ob = ctx_a.odometry(1), ctx_b.odometry(1)
    first = []
    for t in range(4):
        oa.step_host(imgs[t][None]); ob.step_host(imgs[t][None])
        first.append((oa.poses()[0].copy(), ob.poses()[0].copy()))
    assert not np.allclose(first[-1][0], first[-1][1], atol=1e-6)  # P2L r=3.0 vs P2D r=3.5 differ, i.e. no shared state
    oa.reset()
    for t in range(4):
        oa.step_host(imgs[t][None])
        assert np.array_equal(oa.poses()[0], first[t][0])
    fb = oracle.Fuser(mk_params(oracle, cost=2, res=3.5, submap_scan_size=2))
    for t in range(4):
        exp = fb.process_polar(imgs[t])
    assert np.all(np.abs(first[-1][1][:2] - exp[:2]) < POS_TOL)
    oa.release(); ob.release(); ctx_a.close(); ctx_b.close()


def dense_clouds(n_frames, n_points=4800, seed=5):
    """Clouds with many surface points: a lattice of wall segments seen from poses on a slow arc (float32 [N, 3] each)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    segs = []
    for i in range(-6, 7):  # walls every 10 m, both directions, with gaps
        for j in range(-6, 6):
            if rng.uniform() < 0.7:
                segs.append(((10.0 * i, 10.0 * j + 1.0), (10.0 * i, 10.0 * j + 8.0)))
            if rng.uniform() < 0.7:
                segs.append(((10.0 * j + 1.0, 10.0 * i), (10.0 * j + 8.0, 10.0 * i)))
    segs = np.asarray(segs)
    t = rng.uniform(size=(40000, 1))
    s = segs[rng.integers(0, len(segs), size=40000)]
    world = s[:, 0] * (1 - t) + s[:, 1] * t
    poses, clouds = [], []
    for f in range(n_frames):
        x, y, th = 0.8 * f, 0.1 * f, 0.015 * f
        c, sn = np.cos(th), np.sin(th)
        d = world - np.array([x, y])
        loc = np.stack([c * d[:, 0] + sn * d[:, 1], -sn * d[:, 0] + c * d[:, 1]], axis=1)
        loc = loc + rng.normal(scale=0.03, size=loc.shape)
        r = np.hypot(loc[:, 0], loc[:, 1])
        loc = loc[(r > 3.0) & (r < 70.0)]
        pick = rng.permutation(len(loc))[:n_points]
        inten = rng.integers(70, 200, size=len(pick)).astype(np.float32)
        clouds.append(np.concatenate([loc[pick], inten[:, None]], axis=1).astype(np.float32))
        poses.append([x, y, th])
    return clouds, np.asarray(poses)


@pytest.mark.parametrize("cost,res,rev", [(1, 1.5, False), (2, 1.5, False), (1, 1.0, False), (1, 1.5, True)])
def test_register_many_cells(oracle, cost, res, rev):
    """Scans with more surface points than the registration workgroup has threads (blocks of source cells, matches parked
    between the two passes) and more residual blocks than the LDS match array holds (global match arrays); reversed, the
    source scan has more than four blocks of cells (contiguous pair ranges per thread)."""
    clouds, gt = dense_clouds(5)
    if rev:
        clouds, gt = clouds[::-1], gt[::-1].copy()
        assert len(oracle.Scan(clouds[-1], mk_params(oracle, res=res)).cells()) > 1024
    kw = dict(cost=cost, res=res, regularization=0.1, covar_scale=1.0)
    po, pg = mk_params(oracle, **kw), mk_params(capi, **kw)
    ctx = capi.Context(pg, 400, 3360)
    so = [oracle.Scan(c, po) for c in clouds]
    sg = [ctx.scan_create(ctx.cloud_upload(c)) for c in clouds]
    assert len(so[-1].cells()) > 256, len(so[-1].cells())
    poses = gt.copy()
    poses[-1] += np.array([0.25, -0.15, 0.008])
    for n in (2, 5):
        reto, Po, covo, So = oracle.register(so[-n:], poses[-n:], po)
        retg, Pg, covg, Sg = ctx.register(sg[-n:], poses[-n:])
        assert So.num_residual_blocks == Sg.num_residual_blocks and So.num_residuals == Sg.num_residuals
        if n == 5:
            assert Sg.num_residual_blocks > 636  # beyond the LDS match array
        assert So.outer_iterations == Sg.outer_iterations
        assert list(So.inner_iterations[:8]) == list(Sg.inner_iterations[:8])
        assert bool(reto) == retg
        assert np.all(np.abs(Pg[:, :2] - Po[:, :2]) < POS_TOL)
        assert np.all(np.abs(Pg[:, 2] - Po[:, 2]) < ROT_TOL)
        assert np.allclose(Sg.final_cost, So.final_cost, rtol=1e-9)
        assert np.linalg.norm(Pg[-1, :2] - gt[-1, :2]) < 0.1  # known answer
    ctx.close()


def test_profile_hooks_time_every_stage(oracle):
    """cfear_odometry_profile / _profile_read / _profile_read_stages (the HIP-event timing bench.py's roofline leg reads):
    one filter launch and one features + registration pair per profiled step, positive durations, results untouched."""
    imgs, _ = synth.world_sequence(5, seed=21)
    pg = mk_params(capi)
    ctx = capi.Context(pg, 400, 3360)
    odo, ref = ctx.odometry(2), ctx.odometry(2)
    batch = lambda t: np.stack([imgs[t], imgs[t]])
    odo.step_host(batch(0)); ref.step_host(batch(0))
    odo.profile(True)
    for t in range(1, 5):
        odo.step_host(batch(t)); ref.step_host(batch(t))
    tf, nf = odo.profile_read()
    tfeat, treg, ns = odo.profile_read_stages()
    assert nf == 4 and ns == 4
    assert 0 < tf < 1.0 and 0 < tfeat < 1.0 and 0 < treg < 1.0
    assert np.array_equal(odo.poses(), ref.poses())
    odo.profile(False)
    odo.step_host(batch(4))
    assert odo.profile_read()[1] == 0 and odo.profile_read_stages()[2] == 0
    odo.release(); ref.release()
    ctx.close()


@pytest.mark.parametrize("mode", ["detailed", "light", "workgroups", "controller"])
def test_phase_time_modes_leave_the_results_alone(oracle, mode):
    """cfear_odometry_phase_times: the timed instantiations of the features / registration kernels (per-phase clock stamps, the
    command-loop breakdown) and the workgroup stamps of the production kernels give the poses of an untimed run, stamps that
    increase along a kernel, and - in the controller mode - positive accumulators for every part of the command loop."""
    imgs, _ = synth.world_sequence(5, seed=23)
    ctx = capi.Context(mk_params(capi), 400, 3360)
    odo, ref = ctx.odometry(3), ctx.odometry(3)
    batch = lambda t: np.stack([imgs[t]] * 3)
    odo.step_host(batch(0)); ref.step_host(batch(0))
    odo.phase_times(None, light=mode == "light", workgroups_only=mode == "workgroups", controller=mode == "controller")
    for t in range(1, 5):
        odo.step_host(batch(t)); ref.step_host(batch(t))
        ticks = odo.phase_times(True)
        assert ticks.shape == (3, 32)
        for q in range(3):
            if mode == "workgroups":
                assert 0 < ticks[q, 0] < ticks[q, 1] <= ticks[q, 14] < ticks[q, 15]  # features start / end, registration start / end
            else:
                reg = ticks[q, 14:29]; reg = reg[reg > 0]
                assert len(reg) >= 4 and np.all(np.diff(reg) >= 0)
                if mode == "controller":
                    assert ticks[q, 3] >= 1 and np.all(ticks[q, [1, 4, 6]] > 0)  # commands; evaluation, state function (+ step), publish
                else:
                    feat = ticks[q, :14]; feat = feat[feat > 0]
                    assert len(feat) >= 8 and np.all(np.diff(feat) >= 0)
                if mode == "detailed":
                    assert ticks[q, 31] >= 1 and ticks[q, 29] > 0 and ticks[q, 30] > 0
    assert np.array_equal(odo.poses(), ref.poses())
    odo.phase_times(False)
    odo.step_host(batch(4)); ref.step_host(batch(4))
    assert np.array_equal(odo.poses(), ref.poses())
    odo.release(); ref.release()
    ctx.close()


def test_many_resident_sequences_are_independent_and_deterministic(oracle):
    """768 resident sequences (three registration workgroups on every compute unit) replaying 4 different sweeps streams:
    every replica of a stream ends bit-identical to the others, whatever workgroup slot and neighbours it had, and
    within tolerance of the oracle."""
    import torch
    T, U, B = 10, 4, 768
    streams = [synth.world_sequence(T, seed=40 + u, world_seed=1300 + u, t0=11 * u)[0] for u in range(U)]
    d_unique = torch.from_numpy(np.stack(streams)).cuda()  # [U, T, A, R]
    idx = (torch.randperm(B, generator=torch.Generator().manual_seed(7)) % U).cuda()
    pg, po = mk_params(capi), mk_params(oracle)
    # torch produces every step's input right before the step: both must use the same explicit stream (torch's default
    # stream has handle 0, which cfear_create reads as "create your own": unordered with torch's work)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ctx = capi.Context(pg, 400, 3360, stream=side.cuda_stream)
        odo = ctx.odometry(B)
        for t in range(T):
            d = d_unique[idx, t].contiguous()
            odo.step_device(d.data_ptr())
            side.synchronize()  # d stays alive until the step has read it
        got = odo.poses()
    kinds = idx.cpu().numpy()
    for u in range(U):
        rows = got[kinds == u]
        assert rows.shape[0] > 100
        assert np.all(rows == rows[0]), u  # bitwise
        fu = oracle.Fuser(po)
        for t in range(T):
            exp = fu.process_polar(streams[u][t])
        assert np.all(np.abs(rows[0][:2] - exp[:2]) < POS_TOL) and abs(rows[0][2] - exp[2]) < ROT_TOL
    odo.release()
    ctx.close()


def test_batched_odometry_covariances_match_the_oracle_fuser(oracle):
    """cov_current of the device fuser (cfear_odometry_covariances) = the oracle fuser's, sweep by sweep: the registration covariance
    30 * final_cost / dof * (J~^T J~)^-1 with its (1,5)/(5,1) quirk (n_scan_normal.cpp:392-433), for P2L and P2D."""
    imgs, _ = synth.world_sequence(7, seed=29, world_seed=91)
    for cost in (1, 2):
        kw = dict(range_res=np.float32(0.0595238), k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, submap_scan_size=4, cost=cost,
                  regularization=0.1, covar_scale=1.0)
        ctx = capi.Context(capi.default_params(**kw), 400, 3360)
        odo = ctx.odometry(2)
        fus = [oracle.Fuser(oracle.default_params(**kw)) for _ in range(2)]
        streams = [imgs, imgs[:, ::-1].copy()]
        for t in range(7):
            odo.step_host(np.stack([s[t] for s in streams]))
            cov = odo.covariances()
            for q in range(2):
                fus[q].process_polar(streams[q][t])
                if t > 0:
                    e = fus[q].last_cov()
                    assert np.allclose(cov[q], e, rtol=1e-6, atol=1e-14), (cost, t, q)
                    assert cov[q][1, 5] == 0 and cov[q][5, 1] == 0 and cov[q][0, 0] > 0
        odo.release()
        ctx.close()
