"""Feature builds of awkward clouds, device vs oracle: single voxels with thousands of members, duplicates, lines, clouds of
1 / 5 / 6 points, weights that are all zero, sparse clouds with one point per voxel, random mixtures. Same cells (or the same
failure) either way; the 1-NN grid answers the same queries."""
import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import capi

pytestmark = pytest.mark.gpu
RR = np.float32(0.0595238)


def mk(mod, **kw):
    base = dict(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, compensate=1, radar_ccw=0, cost=1, loss=1,
                loss_limit=0.1, submap_scan_size=4)
    base.update(kw)
    return mod.default_params(**base)


def clouds():
    rng = np.random.default_rng(2024)
    I = lambda n, lo=61, hi=255: rng.integers(lo, hi + 1, n).astype(np.float32)
    out = {}
    out["one_dense_voxel"] = np.column_stack([rng.uniform(30.1, 32.9, 4000), rng.uniform(-11.9, -9.1, 4000), I(4000)])
    out["four_dense_voxels"] = np.column_stack([rng.uniform(28.0, 33.9, 4800), rng.uniform(-13.0, -7.1, 4800), I(4800)])
    p = np.column_stack([rng.uniform(-40, 40, 300), rng.uniform(-40, 40, 300), I(300)])
    out["every_point_eight_times"] = np.repeat(p, 8, axis=0)[rng.permutation(2400)]
    t = rng.uniform(-150, 150, 3000)
    out["a_line"] = np.column_stack([t, 0.3 * t + 5.0, I(3000)])
    out["a_line_with_noise"] = np.column_stack([t, 0.3 * t + 5.0 + rng.normal(0, 0.05, 3000), I(3000)])
    blob = np.column_stack([rng.normal(50, 0.8, 40), rng.normal(20, 0.8, 40), I(40)])
    out["one_point"] = blob[:1]
    out["five_points"] = blob[:5]
    out["six_points"] = blob[:6]
    out["forty_points"] = blob
    w0 = np.column_stack([rng.normal(0, 30, 3000), rng.normal(0, 30, 3000), I(3000, 0, 60)])
    out["weights_all_zero"] = w0
    mix = w0.copy(); mix[::3, 2] = I(1000)
    out["a_third_of_the_weights_positive"] = mix
    gx, gy = np.meshgrid(np.arange(-30, 30), np.arange(-30, 30))
    out["one_point_per_voxel"] = np.column_stack([gx.ravel() * 3.0 + 1.5, gy.ravel() * 3.0 + 1.5, I(3600)])
    out["voxel_edges"] = np.column_stack([rng.integers(-40, 40, 4000) * 3.0 + rng.choice([0.0, 2.9999998, 1e-6], 4000), rng.integers(-40, 40, 4000) * 3.0, I(4000)])
    for s in range(6):
        n = int(rng.integers(200, 4864))
        nb = int(rng.integers(3, 60))
        cx, cy = rng.uniform(-180, 180, nb), rng.uniform(-180, 180, nb)
        sg = rng.uniform(0.2, 6.0, nb)
        b = rng.integers(0, nb, n)
        out["mixture_%d" % s] = np.column_stack([cx[b] + rng.normal(0, 1, n) * sg[b], cy[b] + rng.normal(0, 1, n) * sg[b], I(n, 0, 255)])
    return {k: v.astype(np.float32) for k, v in out.items()}


CLOUDS = clouds()


@pytest.mark.parametrize("name", sorted(CLOUDS))
@pytest.mark.parametrize("res,df,wi", [(3.0, 1.0, 1), (3.5, 2.0, 0)])
def test_awkward_clouds_give_the_oracles_cells(oracle, name, res, df, wi):
    xyi = CLOUDS[name]
    po, pg = mk(oracle, res=res, downsample_factor=df, weight_intensity=wi), mk(capi, res=res, downsample_factor=df, weight_intensity=wi)
    ctx = capi.Context(pg, 400, 3360)
    so = oracle.Scan(xyi, po)
    sg = ctx.scan_create(ctx.cloud_upload(xyi))
    co, cg = so.cells(), sg.cells()
    assert len(cg) == len(co), (len(cg), len(co))
    if len(co):
        for f in ("mean", "cov", "normal", "lambda_min", "lambda_max", "scale", "sum_intensity", "avg_intensity"):
            assert np.allclose(cg[f], co[f], rtol=1e-9, atol=1e-9), f
        assert np.array_equal(cg["nsamples"], co["nsamples"])
        rng = np.random.default_rng(7)
        q = co["mean"][rng.integers(0, len(co), 100)] + rng.normal(0, 1.5, (100, 2))
        assert np.array_equal(sg.closest(q, 2.0), np.array([so.closest(x, y, 2.0) for x, y in q]))
    ctx.close()


@pytest.mark.parametrize("df", [6.0, 16.0, 24.0])
def test_dense_blob_with_a_large_downsample_factor(oracle, df):
    """Hundreds of voxels that each see thousands of candidates: more than 2^16 chunks of 16 candidates. The chunk count
    and the active-sample count used to share one packed scan with 16 bits each; the chunk count carried into the other
    field and the chunk list ran past its array. Now the packed scan flags the overflow and the chunk size is doubled."""
    rng = np.random.default_rng(11)
    n = 4800
    xyi = np.column_stack([rng.uniform(10.0, 13.0, n), rng.uniform(-4.0, -1.0, n), rng.integers(61, 256, n)]).astype(np.float32)
    po, pg = mk(oracle, res=3.0, downsample_factor=df), mk(capi, res=3.0, downsample_factor=df)
    ctx = capi.Context(pg, 400, 3360)
    so = oracle.Scan(xyi, po)
    sg = ctx.scan_create(ctx.cloud_upload(xyi))
    co, cg = so.cells(), sg.cells()
    assert len(cg) == len(co) and len(co) > 30, (len(cg), len(co))
    for f in ("mean", "cov", "normal", "lambda_min", "lambda_max", "scale", "sum_intensity", "avg_intensity"):
        assert np.allclose(cg[f], co[f], rtol=1e-9, atol=1e-9), f
    assert np.array_equal(cg["nsamples"], co["nsamples"])
    ctx.close()
