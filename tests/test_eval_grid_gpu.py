"""The reference's own evaluation grid (launch/oxford/eval/params/*: the hot-path settings its published tables sweep) through
the device fuser, against the oracle's fuser at EVERY sweep - keyframe / outer / inner iteration / residual / cell counts and
the pose (1e-4 m, 1e-5 rad) - by both routes: cfear_odometry_replay_host with one persistent workgroup per sequence
(drive_parity.run) and the batched features + registration launches per sweep (drive_parity.run_batched, three sequences).

  * params/baseline/oxford_cfear-1:13-25 - CFEAR-1, the reference's fastest published configuration: P2L, ONE keyframe
    (submap_scan_size 1: a ring of one slot that turns over at every new keyframe), res 3.5, unweighted, weight_option 4;
  * params/motion_compensation/motion_compensation_cfear-3:12-17 - disable_compensate = true (offline_odometry.cpp:168,272:
    compensate = !disable_compensate; odometrykeyframefuser.cpp:146-150 skipped) x {P2P, P2L} x submap 1 / 4 x k 12 / 40,
    weighted by intensity, weight_option 0;
  * params/resolution/oxford_cfear-3:13-16 - res 1 ... 5 x submap 1 / 3: at res 1 a dense scan has more than a thousand
    oriented surface points, which the registration of the fuser takes in several blocks of source cells
    (registration_dev.h build_problem_block) and a sized-down object has to hold (cfear_tune MAX_CELLS);
  * params/loss_function/loss_function_cfear-3:13-22 - loss_limit 0.01 ... 4 x {None, Cauchy, Tukey, SoftLOne, Huber}, P2P,
    four keyframes;
  * params/weight_residual/oxford_cfear-3:25 - weight_option 0 ... 5, P2P, k = 40;
  * params/submap_keyframes/submap_keyframe_cfear-3:13-15 - submap_scan_size 1 ... 10 x {P2P, P2L, P2D}, unweighted, the Mulran twin
    (mulran_keyframe_cfear-3:11,20) with radar_ccw = true and range_res 0.0595238 (tests/test_large_submap_gpu.py has 5, 7, 8, 10
    and 50 keyframes by the batched routes; here the sizes between, by the persistent replay route);
  * params/grid_search/oxford_cfear-3-p2l:13-20 - the corners of the P2L grid search: k 30 / 50, z_min 50 / 70, res 2.5 / 2.75,
    5 / 6 keyframes, loss_limit 0.2 / 0.3.
"""
import os

import numpy as np
import pytest

import drive_parity

pytestmark = pytest.mark.gpu

P2P, P2L, P2D = 0, 1, 2
NONE, HUBER, CAUCHY, SOFTLONE, COMBINED, TUKEY = 0, 1, 2, 3, 4, 5
LOSS_NAMES = {NONE: "None", HUBER: "Huber", CAUCHY: "Cauchy", SOFTLONE: "SoftLOne", TUKEY: "Tukey"}

CFEAR1 = dict(cost=P2L, submap_scan_size=1, res=3.5, k_strongest=12, z_min=60.0, loss=HUBER, loss_limit=0.1, covar_scale=1.0,
              regularization=1.0, weight_intensity=0, weight_opt=4, compensate=1)
# what the four sweeps below hold fixed (their lines 18-26)
CFEAR3_GRID = dict(k_strongest=12, z_min=60.0, res=3.0, loss=HUBER, loss_limit=0.1, covar_scale=1.0, regularization=1.0, weight_intensity=1,
                   weight_opt=0, compensate=1, submap_scan_size=4, cost=P2P)

SCALE = float(os.environ.get("CFEAR_EVAL_GRID_SCALE", "1"))  # tools/: longer drives with the same cases


def _replay(oracle, name, kind, sweeps, params, **kw):
    out = drive_parity.run(oracle, max(int(sweeps * SCALE), 8), kind, world_seed=6, seed=9, params=params, **kw)
    m = out["mismatches"]
    assert not m, "%s: %d sweeps disagree; first (sweep, what, device, oracle): %r" % (name, len(m), m[:3])
    d, c = out["drift_dev"], out["drift_cpu"]
    assert d["segments"] == c["segments"]
    if d["segments"]:
        assert abs(d["translation_percent"] - c["translation_percent"]) < 1e-6
    return out


# ---- CFEAR-1 -------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["blocks", "canyon"])
def test_cfear1_replay_matches_oracle_at_every_sweep(oracle, kind):
    out = _replay(oracle, "cfear-1 " + kind, kind, 450, CFEAR1)  # (10 000 sweeps of this preset: profiles/r05_drive10k_blocks_cfear1.json)
    assert out["keyframes_max"] == 1  # the ring of one


@pytest.mark.parametrize("route", ["step", "replay"])
def test_cfear1_batched_route_matches_oracle(oracle, route):
    st = {}
    drive_parity.run_batched(oracle, CFEAR1, "canyon" if route == "step" else "blocks", max(int(120 * SCALE), 8), route=route, stats=st)
    assert st["keyframes_max"] == 1 and st["residuals_max"] > 100


# ---- motion compensation off ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cost,s,k", [(c, s, k) for c in (P2P, P2L) for s in (1, 4) for k in (12, 40)])
def test_compensation_disabled_replay(oracle, cost, s, k):
    p = dict(CFEAR3_GRID, compensate=0, cost=cost, submap_scan_size=s, k_strongest=k)
    out = _replay(oracle, "compensate=0 cost %d s %d k %d" % (cost, s, k), "blocks" if k == 12 else "canyon", 150, p)
    assert out["keyframes_max"] == s


@pytest.mark.parametrize("cost,s,k,route", [(P2L, 1, 12, "step"), (P2P, 4, 40, "step"), (P2L, 4, 12, "replay"), (P2P, 1, 40, "replay"), (P2L, 2, 12, "step"),
                                            (P2P, 3, 12, "replay")])
def test_compensation_disabled_batched_route(oracle, cost, s, k, route):
    p = dict(CFEAR3_GRID, compensate=0, cost=cost, submap_scan_size=s, k_strongest=k, res=2.75 if s in (2, 3) else 3.0)
    drive_parity.run_batched(oracle, p, "blocks", max(int(70 * SCALE), 8), route=route)


def test_compensation_flag_changes_the_trajectory(oracle):
    """(the two settings are not accidentally the same code path: with the sensor moving, the uncompensated drive differs)"""
    a = drive_parity.run(oracle, 60, "blocks", world_seed=6, seed=9, params=dict(CFEAR3_GRID, compensate=1, cost=P2L))
    b = drive_parity.run(oracle, 60, "blocks", world_seed=6, seed=9, params=dict(CFEAR3_GRID, compensate=0, cost=P2L))
    assert not a["mismatches"] and not b["mismatches"]
    assert np.abs(a["poses_dev"] - b["poses_dev"]).max() > 1e-3


# ---- resolution -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("res,s,cost", [(r, s, c) for r in (1.0, 2.0, 5.0) for s in (1, 3) for c in (P2P, P2L)])
def test_resolution_sweep_replay(oracle, res, s, cost):
    p = dict(CFEAR3_GRID, res=res, submap_scan_size=s, cost=cost)
    _replay(oracle, "res %g s %d cost %d" % (res, s, cost), "canyon", 120, p)


@pytest.mark.parametrize("res,s,cost,route,max_cells", [
    (1.0, 3, P2L, "step", 2048), (1.0, 1, P2P, "replay", 2048), (2.0, 3, P2P, "step", None), (1.0, 3, P2P, "step", None)])
def test_resolution_dense_scans_take_several_blocks_of_source_cells(oracle, res, s, cost, route, max_cells):
    """'thicket' with k = 40: 1200-1600 oriented surface points per sweep - more than four blocks of 256 source cells, more
    residual blocks than the LDS match array holds - inside the fuser, with the object sized for 2048 cells or for every point"""
    p = dict(CFEAR3_GRID, res=res, submap_scan_size=s, cost=cost, k_strongest=40)
    st = {}
    drive_parity.run_batched(oracle, p, "thicket", max(int(40 * SCALE), 8), route=route, max_cells=max_cells, stats=st)
    assert st["cells_max"] > 1100, st
    assert st["residuals_max"] > 900, st
    if route == "step":
        assert st["assoc_paths"] == [2], st  # several blocks of source cells: the grouped path, not the pair ranges


def test_resolution_dense_scans_replay_persistent(oracle):
    p = dict(CFEAR3_GRID, res=1.0, submap_scan_size=3, cost=P2L, k_strongest=40)
    out = _replay(oracle, "res 1 thicket", "thicket", 120, p)
    assert out["cells"].max() > 1100


# ---- loss functions ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("loss,limit", [(l, v) for l in (NONE, CAUCHY, TUKEY, SOFTLONE, HUBER) for v in (0.01, 1.0, 4.0)])
def test_loss_function_sweep_replay(oracle, loss, limit):
    p = dict(CFEAR3_GRID, loss=loss, loss_limit=limit)
    _replay(oracle, "loss %s limit %g" % (LOSS_NAMES[loss], limit), "blocks", 120, p)


@pytest.mark.parametrize("loss,limit,route", [(TUKEY, 0.01, "step"), (CAUCHY, 4.0, "replay"), (SOFTLONE, 1.0, "step"), (NONE, 0.1, "replay"), (HUBER, 0.01, "step")])
def test_loss_function_sweep_batched_route(oracle, loss, limit, route):
    p = dict(CFEAR3_GRID, loss=loss, loss_limit=limit)
    drive_parity.run_batched(oracle, p, "canyon", max(int(60 * SCALE), 8), route=route)


# ---- residual weights -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("weight_opt", [0, 1, 2, 3, 4, 5])
def test_weight_option_sweep_replay(oracle, weight_opt):
    p = dict(CFEAR3_GRID, k_strongest=40, weight_opt=weight_opt)
    _replay(oracle, "weight_option %d" % weight_opt, "blocks", 120, p)


# ---- keyframes per submap ---------------------------------------------------------------------------------------------------------
SUBMAP_GRID = dict(CFEAR3_GRID, weight_intensity=0, weight_opt=0, min_keyframe_dist=1.5)


@pytest.mark.parametrize("s,cost", [(s, c) for s in (2, 3, 6, 9) for c in (P2P, P2L, P2D)])
def test_submap_keyframes_sweep_replay(oracle, s, cost):
    p = dict(SUBMAP_GRID, submap_scan_size=s, cost=cost)
    out = _replay(oracle, "submap %d cost %d" % (s, cost), "blocks", 60 + 25 * s, p)
    assert out["keyframes_max"] == s


@pytest.mark.parametrize("s,cost", [(4, P2D), (6, P2L)])
def test_submap_keyframes_mulran_twin(oracle, s, cost):
    """the Mulran presets turn the sensor the other way round (radar_ccw: the sign of the sweep's bearing in Compensate, utils.cpp:96-113)"""
    p = dict(SUBMAP_GRID, submap_scan_size=s, cost=cost, radar_ccw=1)
    _replay(oracle, "mulran submap %d cost %d" % (s, cost), "canyon", 60 + 25 * s, p)


# ---- the P2L grid search ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,zmin,res,s,limit,route", [(50, 70.0, 2.5, 6, 0.3, "replay"), (30, 50.0, 2.75, 5, 0.2, "step"), (50, 50.0, 3.0, 4, 0.1, "step")])
def test_grid_search_corners(oracle, k, zmin, res, s, limit, route):
    p = dict(CFEAR3_GRID, cost=P2L, k_strongest=k, z_min=zmin, res=res, submap_scan_size=s, loss_limit=limit)
    st = {}
    drive_parity.run_batched(oracle, p, "canyon", max(int((40 + 12 * s) * SCALE), 8), route=route, stats=st)
    assert st["keyframes_max"] == s, st
