"""The oracle's [3P] sensitivity modes (oracle/cfear_oracle.h CFO_PERT_*; tests/run_3p_sensitivity.py runs them over long drives and
writes profiles/r04_3p_sensitivity.json): here only that they do what they say - and that mode 0 is the oracle."""
import ctypes as C

import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import synth

RR = np.float32(0.0595238)


@pytest.fixture(autouse=True)
def _reset(oracle):
    yield
    oracle.set_perturbation(0)


def _cloud(oracle, t=3):
    img = synth.world_scan(synth.World(1234), t, 400, 3360, RR, seed=5)
    return oracle.cloud(oracle.filter_polar(img, 60, 12), RR, 2.5)


def test_summation_and_eigen_modes_move_cells_by_rounding_only(oracle):
    p = oracle.default_params(range_res=RR, res=3.0, weight_intensity=1)
    xyi = _cloud(oracle)
    base = oracle.Scan(xyi, p).cells()
    for mode in (["sum_reverse"], ["sum_pairwise"], ["wsum_eigen_redux"], ["eig_jacobi"], ["sum_pairwise", "eig_jacobi"]):
        oracle.set_perturbation(mode)
        c = oracle.Scan(xyi, p).cells()
        oracle.set_perturbation(0)
        assert len(c) == len(base) and np.array_equal(c["nsamples"], base["nsamples"])
        for f in ("mean", "cov", "normal", "lambda_min", "lambda_max", "scale", "sum_intensity"):
            assert np.allclose(c[f], base[f], rtol=1e-11, atol=1e-12), (mode, f)
    again = oracle.Scan(xyi, p).cells()
    assert again.tobytes() == base.tobytes()  # mode 0 is the oracle again, bit for bit


def test_stdsort_is_pcl19_voxel_order_and_only_rounds_centroids(oracle):
    """libstdc++ std::sort on the voxel index alone (PCL <= 1.9) is not stable: inside voxels of more than 16 points the point
    order differs from the stable one - the float centroid may move by an ulp or two, the set of points in a voxel never"""
    p = oracle.default_params(range_res=RR, res=3.0, weight_intensity=1)
    xyi = _cloud(oracle)
    s0 = oracle.Scan(xyi, p)
    base_samples = s0.samples().copy()
    moved = {}
    for mode in ("voxel_stdsort", "voxel_reverse", "voxel_random"):
        oracle.set_perturbation([mode], seed=3)
        s = oracle.Scan(xyi, p)
        oracle.set_perturbation(0)
        sm = s.samples()
        assert sm.shape == base_samples.shape
        ulp = np.spacing(np.abs(base_samples).astype(np.float32))
        assert np.all(np.abs(sm - base_samples) <= 8 * ulp)  # same points per voxel, another order of the float additions
        moved[mode] = int(np.sum(sm != base_samples))
    assert moved["voxel_reverse"] > 0 and moved["voxel_stdsort"] > 0  # (the synthetic scans have voxels of ~100 points)
    # the permutation itself: equal voxel keys come out of std::sort in another order than they went in
    oracle.set_perturbation(["voxel_stdsort"])  # loads oracle/libcfear_stdsort.so
    oracle.set_perturbation(0)
    import oracle.binding as ob
    n = 200
    vi = np.zeros(n, dtype=np.uint32); vi[::2] = 1
    pi = np.arange(n, dtype=np.uint32)
    ob._STDSORT.cfo_stdsort_perm(vi.ctypes.data_as(C.c_void_p), pi.ctypes.data_as(C.c_void_p), n)
    assert np.all(np.diff(vi.astype(np.int64)) >= 0) and sorted(pi.tolist()) == list(range(n))
    stable = np.concatenate([np.arange(1, n, 2), np.arange(0, n, 2)])
    assert not np.array_equal(pi, stable)


def test_nn_tie_mode_only_matters_for_equal_float_means(oracle):
    p = oracle.default_params(range_res=RR, res=3.0, weight_intensity=1)
    xyi = _cloud(oracle)
    s = oracle.Scan(xyi, p)
    cells = s.cells()
    m = cells["mean"].astype(np.float32)
    uniq, inv, cnt = np.unique(m, axis=0, return_inverse=True, return_counts=True)
    q = cells["mean"].copy()
    lo = np.array([s.closest(x, y, 2.0) for x, y in q])
    oracle.set_perturbation(["nn_tie_high"])
    hi = np.array([s.closest(x, y, 2.0) for x, y in q])
    oracle.set_perturbation(0)
    dup = cnt[inv.ravel()] > 1
    assert np.array_equal(lo[~dup], hi[~dup])           # a unique nearest cell: no tie, no difference
    assert np.all(hi[dup] >= lo[dup])                   # a tie: the other end of the group of cells with that float mean
    if dup.any():
        assert np.any(hi[dup] > lo[dup])
        assert np.array_equal(m[hi], m[lo])


def test_flann_restatement_returns_a_true_nearest_neighbour(oracle):
    """CFO_PERT_NN_TIE_FLANN: the restated flann::KDTreeSingleIndex descent (oracle/cfear_oracle.c kd_*) against brute force on clouds of
    cell means with exact duplicates, walls and clusters - the returned cell is always AT the minimum float distance, it IS the
    brute-force cell whenever the minimum is unique, and among equidistant cells it is some cell of the tie (which one is the tree's
    business: that is the point of the mode - and it is NOT always the lowest index)."""
    rng = np.random.default_rng(5)
    tie_not_lowest = 0
    for trial in range(16):
        n = int(rng.integers(1, 900)) if trial else 1
        pts = rng.uniform(-80, 80, size=(n, 2)).astype(np.float32)
        if trial % 3 == 0 and n > 20:  # exact duplicates (voxel centroids whose neighbourhoods hold the same weighted points)
            pts[rng.integers(0, n, size=n // 4)] = pts[rng.integers(0, n, size=n // 4)]
        if trial % 4 == 1:             # a wall: one coordinate constant
            pts[: n // 2, 1] = np.float32(12.5)
        if trial % 5 == 2:             # everything in one place
            pts[:] = pts[0]
        q = np.concatenate([pts[rng.integers(0, n, size=80)].astype(np.float64) + rng.normal(0, 0.7, size=(80, 2)),
                            pts[rng.integers(0, n, size=40)].astype(np.float64),                                      # queries AT a cell mean
                            0.5 * (pts[rng.integers(0, n, size=40)].astype(np.float64) + pts[rng.integers(0, n, size=40)]),  # midpoints
                            rng.uniform(-200, 200, size=(20, 2))]).astype(np.float32)                                   # outside the bounding box
        idx, dist = oracle.flann_nearest(pts, q)
        for i in range(len(q)):
            dx, dy = q[i, 0] - pts[:, 0], q[i, 1] - pts[:, 1]
            d2 = dx * dx
            d2 = d2 + dy * dy  # float32, the order of L2_Simple
            assert 0 <= idx[i] < n and d2[idx[i]] == d2.min() and dist[i] == d2.min(), (trial, i, idx[i], int(np.argmin(d2)))
            if np.sum(d2 == d2.min()) == 1:
                assert idx[i] == int(np.argmin(d2))
            else:
                tie_not_lowest += int(idx[i] != int(np.argmin(d2)))
    assert tie_not_lowest > 0


def test_flann_mode_through_the_scan_search(oracle):
    p = oracle.default_params(range_res=RR, res=3.0, weight_intensity=1)
    s = oracle.Scan(_cloud(oracle), p)
    cells = s.cells()
    m = cells["mean"].astype(np.float32)
    q = cells["mean"] + 0.3
    lo = np.array([s.closest(x, y, 2.0) for x, y in q])
    oracle.set_perturbation(["nn_tie_flann"])
    fl = np.array([s.closest(x, y, 2.0) for x, y in q])
    oracle.set_perturbation(0)
    assert np.array_equal(lo < 0, fl < 0)
    ok = lo >= 0
    assert np.array_equal(m[fl[ok]], m[lo[ok]])  # the same float mean: the same cell or an exact duplicate of it
