"""The oracle against vectors produced BY THE REFERENCE ITSELF (tests/golden/ref_golden.npz, made on a ROS box by
oracle/ref_recipe/). The file cannot be produced in this repository's image; while it is absent these tests are SKIPPED (the
oracle stays "parity unpinned"), and the day it is committed they pin every [3P] assumption of DESIGN.md section 2."""
import os

import numpy as np
import pytest

REF = os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="tests/golden/ref_golden.npz absent: run oracle/ref_recipe on a machine with ROS/PCL/Ceres (parity unpinned until then)")
RR = np.float32(0.0595238)


@pytest.fixture(scope="module")
def ref():
    return np.load(REF)


def test_filter_selection_and_peaks_bit_exact(oracle, ref):
    gold = np.load(GOLD)
    for name in [n[5:] for n in gold.files if n.startswith("tile_")]:
        for k, z in ((12, 60), (5, 0), (40, 61)):
            slots = oracle.filter_polar(gold["tile_" + name], z, k)
            tag = "%s_k%d_z%d" % (name, k, z)
            assert np.array_equal(oracle.cloud(slots, np.float32(0.0438), -1.0), ref["tilecloud_" + tag]), tag
            assert np.array_equal(oracle.cloud(slots, np.float32(0.0438), -1.0, peaks=True), ref["tilepeaks_" + tag]), tag


def test_world_clouds_compensation_and_cells(oracle, ref):
    gold = np.load(GOLD)
    for t in range(8):
        assert np.array_equal(gold["world_cloud_%d" % t], ref["world_cloud_%d" % t]), t  # [3P]-free: filter + polar -> Cartesian
    comp = oracle.compensate(ref["world_cloud_3"], [1.0, 0.01, 0.02], 0)
    assert np.all(np.abs(comp - ref["world3_cloud_comp"]) <= np.spacing(np.abs(ref["world3_cloud_comp"])))  # libm atan2 / sincos
    cells = oracle.Scan(ref["world3_cloud_comp"], oracle.default_params(range_res=RR, res=3.0, weight_intensity=1)).cells()
    assert np.array_equal(cells["nsamples"], ref["world3_cells_nsamples"])  # VoxelGrid order, FLANN strict <, count >= 6
    for f in ("mean", "cov", "lambda_min", "lambda_max", "scale"):
        assert np.allclose(cells[f], ref["world3_cells_" + f], rtol=1e-9, atol=1e-9), f
    assert np.allclose(cells["normal"], ref["world3_cells_normal"], atol=1e-7)  # iterative vs closed-form eigenvectors


def test_nearest_cell_tie_order(oracle, ref):
    """GetClosestIdx (pointnormal.cpp:238-254) queried AT every cell mean: its own index where the float mean is unique; among the cells that
    share a float mean the reference returns whatever FLANN's descent visits first - the `[3P]` row DESIGN.md section 2 shows is worth up to a
    centimetre per registration. Both sides have the rule as a switch since round 5: this test says which setting the real binary matches -
    the restated kd-tree (CFO_PERT_NN_TIE_FLANN = library NN_TIE_RULE 2) must; if the production rule (lowest index) does too on this scan,
    good, and if neither does the restatement of flann::KDTreeSingleIndex in oracle/cfear_oracle.c (kd_*) is what has to be corrected."""
    if "world3_closest_self" not in ref:
        pytest.skip("ref_golden.npz predates the tie-order dump (oracle/ref_recipe/dump_ref_golden.cpp)")
    p = oracle.default_params(range_res=RR, res=3.0, weight_intensity=1)
    s = oracle.Scan(ref["world3_cloud_comp"], p)
    means = s.cells()["mean"]
    want = ref["world3_closest_self"]
    m32 = means.astype(np.float32)
    _, inv, cnt = np.unique(m32, axis=0, return_inverse=True, return_counts=True)
    dup = cnt[inv.ravel()] > 1
    got = {}
    for mode in ("lowest", "nn_tie_flann"):
        oracle.set_perturbation([] if mode == "lowest" else [mode])
        try:
            got[mode] = np.array([s.closest(x, y, 0.5) for x, y in means])
        finally:
            oracle.set_perturbation(0)
        assert np.array_equal(got[mode][~dup], want[~dup]) and np.array_equal(want[~dup], np.arange(len(means))[~dup]), mode
    agree = {mode: bool(np.array_equal(g[dup], want[dup])) for mode, g in got.items()}
    assert agree["nn_tie_flann"], ("the restated FLANN descent does not reproduce the binary's choices among cells with equal float means: binary %r, restatement %r, "
                                   "lowest index %r" % (want[dup].tolist(), got["nn_tie_flann"][dup].tolist(), got["lowest"][dup].tolist()))
    if not agree["lowest"]:
        print("the production tie rule (lowest index) differs from the binary on this scan: compare in the parity mode (NN_TIE_RULE 2)")


@pytest.mark.parametrize("tag,cost", [("p2l", 1), ("p2d", 2)])
def test_trajectory_and_iteration_counts(oracle, ref, tag, cost):
    kw = dict(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, submap_scan_size=4)
    f = oracle.Fuser(oracle.default_params(cost=cost, **kw))
    for t in range(8):
        pose = f.process_cloud(ref["world_cloud_%d" % t])
        S = f.last_summary()
        assert np.all(np.abs(pose[:2] - ref["traj_" + tag][t][:2]) < 1e-4) and abs(pose[2] - ref["traj_" + tag][t][2]) < 1e-5, t
        assert f.num_keyframes == ref["keyframes_" + tag][t]
        if t > 0:
            assert S.outer_iterations == ref["outer_" + tag][t], t                                      # outer loop exits (n_scan_normal.cpp:134-149)
            assert S.inner_iterations[S.outer_iterations - 1] == ref["last_inner_" + tag][t], t         # Ceres LM schedule
            assert S.num_residuals == ref["num_residuals_" + tag][t], t                                 # associations
            assert abs(S.final_cost - ref["final_cost_" + tag][t]) <= 1e-9 * abs(ref["final_cost_" + tag][t]), t


REG_CFGS = {"p2l_huber": dict(cost=1, loss=1, loss_limit=0.1), "p2l_cauchy": dict(cost=1, loss=2, loss_limit=0.2), "p2l_tukey": dict(cost=1, loss=5, loss_limit=0.5),
            "p2d_huber": dict(cost=2, loss=1, loss_limit=0.1), "p2p_huber": dict(cost=0, loss=1, loss_limit=0.1),
            "p2l_softlone": dict(cost=1, loss=3, loss_limit=0.1), "p2l_none": dict(cost=1, loss=0, loss_limit=0.1)}


@pytest.mark.parametrize("tag", sorted(REG_CFGS))
def test_direct_register_covariance_and_get_cost(oracle, ref, tag):
    """n_scan_normal_reg::Register / GetCovariance / GetCost called directly (n_scan_normal.cpp:82-213, 392-433) for every cost and
    the losses the fuser run does not use (registration.cpp:78-97)."""
    kw = dict(range_res=RR, res=3.0, weight_intensity=1, weight_opt=4, regularization=0.1, covar_scale=1.0)
    kw.update(REG_CFGS[tag])
    p = oracle.default_params(**kw)
    scans = [oracle.Scan(ref["world_cloud_%d" % t], p) for t in range(4)]
    gold = np.load(GOLD)
    poses = gold["world_gt"][:4].copy()
    poses[3] += [0.12, -0.07, 0.004]
    ok, P, cov, S = oracle.register(scans, poses, p)
    info = ref["reg_info_" + tag]  # ok, itr_, iterations.size() of the last solve, final_cost, num_residuals, score
    assert bool(ok) == bool(info[0]) and S.outer_iterations == int(info[1]) and S.inner_iterations[S.outer_iterations - 1] == int(info[2])
    assert S.num_residuals == int(info[4]) and abs(S.final_cost - info[3]) <= 1e-9 * abs(info[3])
    assert np.all(np.abs(P[:, :2] - ref["reg_poses_" + tag][:, :2]) < 1e-4) and np.all(np.abs(P[:, 2] - ref["reg_poses_" + tag][:, 2]) < 1e-5)
    assert np.allclose(cov, ref["reg_cov_" + tag], rtol=1e-6, atol=1e-12)  # 30 * final_cost / dof * (J~^T J~)^-1, (1,5)/(5,1) left 0
    got = oracle.get_cost(scans, P, p, itr=S.outer_iterations)
    ok_c, score = ref["getcost_score_" + tag]
    assert (got is not None) == bool(ok_c)
    if got is not None:
        res = ref["getcost_residuals_" + tag][:-1]  # (the recipe appends one 0 so that the record is never empty)
        assert abs(got[0] - score) <= 1e-9 * abs(score) and len(got[1]) == len(res) and np.allclose(got[1], res, rtol=1e-7, atol=1e-10)


def test_ceres_loss_forms(oracle, ref):
    """ceres::LossFunction::Evaluate of the box the vectors were made on against the oracle's restatement (Ceres 2.0 forms). The reference does not pin
    a Ceres version (CMakeLists.txt:38): TukeyLoss of Ceres <= 1.14 is half the 2.0 form - if this fails on the Tukey rows with a factor of two, the
    reference ran on an older Ceres and every Tukey cost / covariance of that run is half the oracle's (the minimum is the same point; the path to it
    is not quite - Jacobi scaling is not invariant under a scaled cost - so p2l_tukey's iteration counts may differ too)."""
    if "ceres_loss_probe" not in ref:
        pytest.skip("vectors made before the loss probe was added to the recipe")
    ss = [0.0, 0.005, 0.01, 0.04, 0.2, 0.25, 1.0]
    cfgs = [(1, 0.1), (2, 0.2), (3, 0.1), (5, 0.5)]  # CFO_LOSS_HUBER, _CAUCHY, _SOFTLONE, _TUKEY with the limits of the recipe
    probe = ref["ceres_loss_probe"]
    for f, (loss, lim) in enumerate(cfgs):
        for i, sv in enumerate(ss):
            got = oracle.loss_eval(loss, lim, sv)
            assert np.allclose(got, probe[f, i], rtol=1e-12, atol=1e-300), (tuple(ref["ceres_version"]), loss, sv, got, probe[f, i])


def test_ca_cfar_cloud(oracle, ref):
    """AzimuthCACFAR::getFilteredPointCloud (cfar.cpp:27-87) on sweep 0 of the fixture"""
    from cfear_radarodometry_code_public_amd import synth
    imgs, _ = synth.world_sequence(1, 400, 3360, RR, seed=21)
    got = oracle.cfar(imgs[0], RR, 60.0, 2.5, window_size=10, nb_guard_cells=20, false_alarm_rate=0.01, max_distance=400.0)
    assert got.shape == ref["cfar_cloud_0"].shape and np.array_equal(got, ref["cfar_cloud_0"])


def test_fuser_covariances(oracle, ref):
    """cov_current of every sweep: the registration covariance (GetCovariance) and, with estimate_cov_by_sampling, the
    cost-sampling one (approximateCovarianceBySampling, odometrykeyframefuser.cpp:261-380)"""
    kw = dict(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, submap_scan_size=4, cost=1)
    for key, sampling in (("fuser_reg_cov_p2l", False), ("fuser_sampled_cov_p2l", True)):
        f = oracle.Fuser(oracle.default_params(**kw))
        f.set_cov_sampling(sampling)
        for t in range(8):
            f.process_cloud(ref["world_cloud_%d" % t])
            if t > 0:
                assert np.allclose(f.last_cov(), ref[key][t], rtol=1e-5, atol=1e-12), (key, t)
