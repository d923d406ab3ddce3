"""The HIP path against vectors produced by the reference itself (tests/golden/ref_golden.npz, oracle/ref_recipe/).
SKIPPED while the file is absent (it cannot be produced in this repository's image)."""
import os

import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import capi

REF = os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(REF), reason="tests/golden/ref_golden.npz absent: run oracle/ref_recipe on a machine with ROS/PCL/Ceres")]
RR = np.float32(0.0595238)


def test_filter_tiles_against_reference_clouds(oracle):
    ref, gold = np.load(REF), np.load(GOLD)
    for name in [n[5:] for n in gold.files if n.startswith("tile_")]:
        img = gold["tile_" + name]
        for k, z in ((12, 60), (5, 0), (40, 61)):
            ctx = capi.Context(capi.default_params(k_strongest=k, z_min=float(z)), img.shape[0], img.shape[1])
            slots = ctx.kstrongest_host(img)[0]
            ctx.close()
            tag = "%s_k%d_z%d" % (name, k, z)
            assert np.array_equal(oracle.cloud(slots, np.float32(0.0438), -1.0), ref["tilecloud_" + tag]), tag  # slots -> cloud: format conversion only
            assert np.array_equal(oracle.cloud(slots, np.float32(0.0438), -1.0, peaks=True), ref["tilepeaks_" + tag]), tag


def test_cells_and_trajectory_against_reference():
    ref = np.load(REF)
    ctx = capi.Context(capi.default_params(range_res=RR, res=3.0, weight_intensity=1), 400, 3360)
    cells = ctx.scan_create(ctx.cloud_upload(ref["world3_cloud_comp"])).cells()
    assert np.array_equal(cells["nsamples"], ref["world3_cells_nsamples"])
    for f in ("mean", "cov", "lambda_min", "lambda_max", "scale"):
        assert np.allclose(cells[f], ref["world3_cells_" + f], rtol=1e-9, atol=1e-9), f
    ctx.close()
    # the registration trajectory goes through tests/test_golden_gpu.py's caller loop with the reference's clouds: the
    # oracle-vs-reference test (test_ref_golden_cpu.py) and the HIP-vs-oracle tests together pin it; here the end pose
    from test_golden_gpu import replay_clouds  # noqa: E402 (same directory: pytest puts it on sys.path)
    for tag, cost in (("p2l", 1), ("p2d", 2)):
        traj = replay_clouds([ref["world_cloud_%d" % t] for t in range(8)], cost)
        assert np.all(np.abs(traj[:, :2] - ref["traj_" + tag][:, :2]) < 1e-4) and np.all(np.abs(traj[:, 2] - ref["traj_" + tag][:, 2]) < 1e-5)
