"""GPU parity against the committed golden fixtures (tests/golden/oracle_golden.npz, produced by
tests/golden/make_golden.py from the CPU oracle)."""
import os

import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import capi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz")
RR = np.float32(0.0595238)


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_filter_tiles_bit_exact(gold):
    names = [n[5:] for n in gold.files if n.startswith("tile_")]
    for n in names:
        img = gold["tile_" + n]
        for k, z in ((12, 60), (5, 0), (40, 61)):
            ctx = capi.Context(capi.default_params(k_strongest=k, z_min=float(z)), img.shape[0], img.shape[1])
            got = ctx.kstrongest_host(img)[0]
            ctx.close()
            assert np.array_equal(got, gold["slots_%s_k%d_z%d" % (n, k, z)]), (n, k, z)


def test_cloud_and_cells_golden(gold):
    ctx = capi.Context(capi.default_params(range_res=RR, res=3.0, weight_intensity=1), 400, 3360)
    c = ctx.cloud_upload(gold["world3_cloud"])
    ctx.compensate(c, [1.0, 0.01, 0.02], 0)
    got = c.download()
    exp = gold["world3_cloud_comp"]
    assert np.all(np.abs(got - exp) <= np.spacing(np.abs(exp)))
    cells = ctx.scan_create(ctx.cloud_upload(exp)).cells()
    assert np.array_equal(cells["nsamples"], gold["world3_cells_nsamples"])
    for f in ("mean", "cov", "normal", "lambda_min", "lambda_max", "scale"):
        assert np.allclose(cells[f], gold["world3_cells_" + f], rtol=1e-9, atol=1e-9), f
    ctx.close()


def replay_clouds(clouds, cost):
    """The reference's caller logic (odometrykeyframefuser.cpp:143-259) written out over the per-call API:
    -> (trajectory [T, 3], per-sweep [outer, inner...] iteration counts, per-sweep cell counts)."""
    kw = dict(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, submap_scan_size=4)

    def T(p):
        c, s = np.cos(p[2]), np.sin(p[2])
        return np.array([[c, -s, p[0]], [s, c, p[1]], [0, 0, 1.0]])

    def xyt(M):
        return np.array([M[0, 2], M[1, 2], np.arctan2(M[1, 0], M[1, 1])])

    ctx = capi.Context(capi.default_params(cost=cost, **kw), 400, 3360)
    T_prev, Tmot = np.eye(3), np.eye(3)
    ring, traj, iters, ncells = [], [], [], []
    for cl in clouds:
        c = ctx.cloud_upload(cl)
        ctx.compensate(c, xyt(Tmot), 0)
        cur = ctx.scan_create(c)
        Tguess = T_prev @ Tmot
        if not ring:
            ring.append((cur, np.eye(3)))
            pose = np.zeros(3)
            iters.append([0] * 9)
        else:
            poses = np.array([xyt(M) for _, M in ring] + [xyt(Tguess)])
            ok, P, cov, S = ctx.register([s for s, _ in ring] + [cur], poses)
            iters.append([S.outer_iterations] + list(S.inner_iterations[:8]))
            Tcur = T(P[-1])
            Tmot = np.linalg.inv(T_prev) @ Tcur
            Tkd = np.linalg.inv(ring[-1][1]) @ Tcur
            if np.hypot(Tkd[0, 2], Tkd[1, 2]) > 1.5 or abs(np.arctan2(Tkd[1, 0], Tkd[1, 1])) > np.deg2rad(5):
                ring.append((cur, Tcur))
                ring = ring[-4:]
            T_prev = Tcur
            pose = xyt(Tcur)
        traj.append(pose)
        ncells.append(cur.size)
    ctx.close()
    return np.array(traj), iters, ncells


def test_registration_golden_trajectory(gold):
    """Replays the golden clouds through the per-call API with the reference's caller logic."""
    for cost, tag in ((1, "p2l"), (2, "p2d")):
        traj, iters, ncells = replay_clouds([gold["world_cloud_%d" % t] for t in range(8)], cost)
        for t in range(8):
            if t > 0:
                assert iters[t] == list(gold["iters_" + tag][t]), (tag, t)
            assert ncells[t] == gold["ncells_" + tag][t]
            assert np.all(np.abs(traj[t][:2] - gold["traj_" + tag][t][:2]) < 1e-4), (tag, t)
            assert abs(traj[t][2] - gold["traj_" + tag][t][2]) < 1e-5
