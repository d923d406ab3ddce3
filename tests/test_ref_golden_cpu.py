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
