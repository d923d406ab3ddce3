"""The reference's large-submap presets through the device fuser (odometrykeyframefuser.cpp:470-494: a ring of up to
submap_scan_size keyframes; n_scan_normal.cpp:359-367: one scan pair per keyframe), against the oracle's fuser at EVERY sweep -
keyframe / outer / inner iteration / residual / cell counts and the pose (1e-4 m, 1e-5 rad):

  * params/baseline_p2d/oxford_cfear-3-s10 and -p2d-s10: 10 keyframes, k = 40, Cauchy(0.1), P2P resp. P2D (regularization 0.1);
  * params/submap_keyframes/submap_keyframe_cfear-3:15 sweeps submap_scan_size over 1..10 (k = 12, Huber, unweighted): 8 is the
    first value past the batched registration kernels compiled for <= 8 scans (CFEAR_STEP_SMALL_SCANS), 5 the first past the
    four-keyframes-at-once association;
  * launch/oxford_demo:62-71, CFEAR-3-s50: 50 keyframes, Cauchy, k = 40, P2P - the reference's most accurate published setting.

With submap_scan_size > 7 the batched step launches register_step_kernel<false, -1> of pipeline.hip (64-scan shared state, cost
read at run time); the replay route runs replay_chunk_kernel. Both routes are driven here, on the street canyon (several echoes
per azimuth, >= 500 oriented surface points per sweep at k = 12) long enough for the 50-slot ring to fill and turn over."""
import os

import numpy as np
import pytest

import drive_parity
from cfear_radarodometry_code_public_amd import capi, kitti, synth

pytestmark = pytest.mark.gpu

S10_P2P = dict(k_strongest=40, cost=0, loss=2, loss_limit=0.1, submap_scan_size=10, res=3.0, weight_intensity=1, weight_opt=4, regularization=0.1, covar_scale=1.0)
S10_P2D = dict(S10_P2P, cost=2)
S50 = dict(k_strongest=40, cost=0, loss=2, loss_limit=0.1, submap_scan_size=50, res=3.0, weight_intensity=1, weight_opt=4)
SWEEP = dict(k_strongest=12, loss=1, loss_limit=0.1, res=3.0, weight_intensity=0, weight_opt=0, regularization=1.0, covar_scale=1.0)


@pytest.mark.parametrize("name,kind,sweeps,params", [
    ("s10_p2p", "canyon", 300, S10_P2P),
    ("s10_p2d", "canyon", 300, S10_P2D),
    ("s8_p2l", "blocks", 400, dict(SWEEP, cost=1, submap_scan_size=8)),
    ("s5_p2d", "blocks", 300, dict(SWEEP, cost=2, submap_scan_size=5)),
    ("s50_cfear3", "canyon", 600, S50),
])
def test_large_submap_replay_matches_oracle_at_every_sweep(oracle, name, kind, sweeps, params):
    """replay route (cfear_odometry_replay_host, one persistent workgroup per sequence)"""
    T = int(os.environ.get("CFEAR_LARGE_SUBMAP_SWEEPS", str(sweeps)))
    out = drive_parity.run(oracle, T, kind, world_seed=4, seed=7, params=params)
    m = out["mismatches"]
    assert not m, "%s: %d sweeps disagree; first (sweep, what, device, oracle): %r" % (name, len(m), m[:3])
    d, c = out["drift_dev"], out["drift_cpu"]
    assert d["segments"] == c["segments"]
    if d["segments"]:
        assert abs(d["translation_percent"] - c["translation_percent"]) < 1e-6
    s = params["submap_scan_size"]
    if T >= 4 * s:  # the ring filled (and turned over)
        assert out["keyframes_max"] == s


_batched_parity = drive_parity.run_batched


@pytest.mark.parametrize("name,kind,sweeps,params,route", [
    ("s10_p2p", "canyon", 60, S10_P2P, "step"),
    ("s10_p2d", "canyon", 60, S10_P2D, "replay"),
    ("s8_p2l", "blocks", 80, dict(SWEEP, cost=1, submap_scan_size=8), "step"),
    ("s8_p2p", "blocks", 60, dict(SWEEP, cost=0, submap_scan_size=8), "replay"),
    ("s7_p2l", "blocks", 60, dict(SWEEP, cost=1, submap_scan_size=7), "step"),   # the last size of the production instantiations
    ("s50_cfear3", "canyon", 140, S50, "step"),
    ("s63_p2l", "blocks", 170, dict(SWEEP, cost=1, submap_scan_size=63), "step"),  # the most a sequence keeps (64 scans with the current one): sixteen groups of four keyframes
])
def test_large_submap_batched_route_matches_oracle(oracle, name, kind, sweeps, params, route):
    """batched route: features_step_kernel + register_step_kernel per sweep, three sequences side by side"""
    st = {}
    kmax = _batched_parity(oracle, params, kind, sweeps, route=route, stats=st)
    assert kmax <= params["submap_scan_size"]  # the ring never exceeds submap_scan_size
    if route == "step":  # which association path ran (cfear_reg_summary::assoc_path): the grouped one once there are more than four keyframes -
        assert 3 not in st["assoc_paths"] and (2 in st["assoc_paths"]) == (kmax > 4), st  # never the pair-by-pair general path
    if sweeps >= 3 * params["submap_scan_size"]:
        assert kmax == params["submap_scan_size"]


@pytest.mark.parametrize("name,kind,sweeps,params,large_kernel", [
    ("s10_p2p", "canyon", 60, S10_P2P, 1),   # the 256-thread shape compiled for 64 scans (what a batch of more sequences than compute units runs)
    ("s10_p2d", "blocks", 60, S10_P2D, 1),
    ("s50_cfear3", "canyon", 130, S50, 1),
    ("s8_p2l", "blocks", 60, dict(SWEEP, cost=1, submap_scan_size=8), 2),  # register_step_large.hip (forced; the default for these three sequences anyway)
    ("s10_p2d", "canyon", 60, S10_P2D, 2),
])
def test_large_submap_both_kernel_shapes_match_oracle(oracle, name, kind, sweeps, params, large_kernel):
    """cfear_tune LARGE_SUBMAP_KERNEL: with more than seven keyframes the batched step has two registration kernels (256 threads x three
    per unit compiled for 64 scans; 512 threads with a unit to itself) and picks by the number of sequences and the submap size - both
    against the oracle's fuser at every sweep, whatever the default would have picked for three sequences"""
    st = {}
    kmax = drive_parity.run_batched(oracle, params, kind, sweeps, route="step", stats=st, large_kernel=large_kernel)
    assert kmax <= params["submap_scan_size"] and 3 not in st["assoc_paths"], st


def test_repeat_shortcut_off_gives_identical_results():
    """An outer iteration that would repeat the previous one bit for bit is not recomputed (ctl_lm_done); with the shortcut
    switched off (cfear_tune REPEAT_SHORTCUT = 0) the iteration runs again - and every summary field, pose and covariance of a
    drive must be identical: the shortcut is an optimisation, not a change of the algorithm."""
    A, R, rr = 400, 3360, np.float32(0.0595238)
    T, B = 60, 2
    frames = np.empty((T, B, A, R), dtype=np.uint8)
    for q in range(B):
        for t0, chunk in synth.drive_chunks(T, "blocks", 31 + q, 41 + q, A, R, rr, ccw=False):
            frames[t0:t0 + len(chunk), q] = chunk
    results = {}
    for cost in (1, 2):
        for shortcut in (1, 0):
            kw = dict(drive_parity.BASE, range_res=rr, cost=cost)
            ctx = capi.Context(capi.default_params(**kw), A, R)
            ctx.tune(capi.TUNE_REPEAT_SHORTCUT, shortcut)
            odo = ctx.odometry(B)
            poses, summ = [], []
            for t in range(T):
                odo.step_host(frames[t])
                poses.append(odo.poses())
                for q in range(B):
                    S = odo.summary(q)[0]
                    summ.append((S.success, S.usable, S.outer_iterations, S.num_residuals, S.num_residual_blocks, S.final_cost, S.score,
                                 list(S.inner_iterations[:8]), list(S.termination[:8]), list(S.outer_cost[:8]), [list(p) for p in S.outer_pose[:8]]))
            cov = odo.covariances()
            results[(cost, shortcut)] = (np.array(poses), summ, cov)
            odo.release()
            ctx.close()
        a, b = results[(cost, 1)], results[(cost, 0)]
        assert np.array_equal(a[0], b[0])
        assert a[1] == b[1]
        assert np.array_equal(a[2], b[2])
        # the shortcut was really taken somewhere: registrations that end with repeated single-evaluation solves
        assert any(s[7][:s[2]][-2:] == [1, 1] for s in a[1] if 4 <= s[2] <= 8)


def test_max_cells_capacity_is_loud_and_otherwise_invisible():
    """cfear_tune MAX_CELLS sizes the scan blocks / residual-block scratch of a batched odometry object for fewer oriented surface
    points than filtered points (280 MB per sequence at s = 50, k = 40 otherwise). Enough capacity: bit-identical results. Too
    little: the reading calls fail with CFEAR_ERR_CAPACITY (-6) and a message naming the limit - never a silently truncated scan.
    An object that cannot fit the GPU is refused with the numbers in the message."""
    A, R, rr = 400, 3360, np.float32(0.0595238)
    T, B = 24, 2
    frames = np.empty((T, B, A, R), dtype=np.uint8)
    for q in range(B):
        for t0, chunk in synth.drive_chunks(T, "canyon", 51 + q, 61 + q, A, R, rr, ccw=False):
            frames[t0:t0 + len(chunk), q] = chunk
    kw = dict(drive_parity.BASE, range_res=rr, submap_scan_size=10, cost=0, loss=2)
    out = {}
    for mc in (0, 1024):
        ctx = capi.Context(capi.default_params(**kw), A, R)
        odo = ctx.odometry(B, max_cells=mc)
        rec = odo.replay_host(frames)
        out[mc] = (rec.copy(), odo.poses(), odo.covariances())
        assert rec["n_cells"].max() <= 1024 and rec["n_cells"][1:].min() > 300
        odo.release()
        ctx.close()
    for a, b in zip(out[0], out[1024]):
        assert np.array_equal(a, b)
    ctx = capi.Context(capi.default_params(**kw), A, R)
    odo = ctx.odometry(B, max_cells=200)
    for t in range(3):
        odo.step_host(frames[t])
    with pytest.raises(capi.CfearError, match=r"rc=-6.*more than 200 oriented surface points"):
        odo.poses()
    with pytest.raises(capi.CfearError, match="rc=-6"):
        odo.summary(0)
    with pytest.raises(capi.CfearError, match=r"rc=-6.*odometry_status"):  # the call for users of the asynchronous device-side replay
        odo.status()
    assert np.array_equal(odo.status(per_sequence=True), [1, 1])  # ... and which sequences (both drive through the canyon)
    odo.reset()  # a reset clears the condition
    odo.release()
    with pytest.raises(capi.CfearError, match=r"rc=-5.*sequences fit.*CFEAR_TUNE_MAX_CELLS"):
        ctx.odometry(200000)  # (sized for every filtered point again: max_cells = 200 above applied to that object only)
    ctx.close()


def test_registration_launch_order_does_not_change_results():
    """cfear_tune REGISTRATION_ORDER: the registration workgroups take the sequences longest first (a counting sort on the device
    over the previous sweep's work). Which workgroup slot a sequence gets must not show in any result."""
    A, R, rr = 400, 3360, np.float32(0.0595238)
    T, U, B = 14, 3, 301
    uniq = np.empty((T, U, A, R), dtype=np.uint8)
    for u, kind in enumerate(("blocks", "canyon", "field")):
        for t0, chunk in synth.drive_chunks(T, kind, 70 + u, 80 + u, A, R, rr, ccw=False):
            uniq[t0:t0 + len(chunk), u] = chunk
    kinds = np.random.default_rng(5).integers(0, U, B)
    out = {}
    for order in (0, 1):
        kw = dict(drive_parity.BASE, range_res=rr)
        ctx = capi.Context(capi.default_params(**kw), A, R)
        odo = ctx.odometry(B, reg_order=order)
        for t in range(T):
            odo.step_host(uniq[t][kinds])
        S = [odo.summary(q) for q in range(0, B, 7)]
        out[order] = (odo.poses(), odo.covariances(), [(s[0].outer_iterations, list(s[0].inner_iterations[:8]), s[0].num_residuals, s[0].final_cost, s[1], s[2]) for s in S])
        odo.release()
        ctx.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    for u in range(U):  # and the replicas of one stream agree among themselves
        assert np.all(out[1][0][kinds == u] == out[1][0][kinds == u][0])
