"""The synthetic worlds (cfear_radarodometry_code_public_amd/synth.py): what the parity tests and the bench feed the path with."""
import numpy as np

from cfear_radarodometry_code_public_amd import synth


def _hits_world(img, pose, A, R, rr, z=120):
    """world coordinates of the strong returns of a sweep rendered without motion distortion"""
    az, rb = np.nonzero(img >= z)
    th = (az + 1) / A * 2 * np.pi + pose[2]
    r = (rb + 0.5) * float(rr)
    return np.stack([pose[0] + r * np.cos(th), pose[1] + r * np.sin(th)], axis=1)


def test_street_texture_is_fixed_to_the_surfaces():
    """render_scan(texture=...): the amplitude of an echo belongs to the patch of surface that was hit, so two sweeps from different places see their
    strong returns at the same world positions (the plain worlds draw a new amplitude per azimuth and sweep) - the property the reference's P2P preset
    needs to behave like odometry (profiles/r06_world_realism.json). And rendering is deterministic."""
    A, R, rr = 400, 3360, np.float32(0.0595238)
    wall = np.array([[[-40.0, 12.0], [40.0, 12.0]]])  # one facade along x, 12 m to the left
    kw = dict(distort=False, hits=1, p_extra=0.0, sigma=1.0, texture=(1.5, 0.2, 0.0))
    p0, p1 = (0.0, 0.0, 0.0), (3.7, 0.4, 0.05)
    a = synth.render_scan(wall, p0, (0, 0, 0), A, R, rr, np.random.Generator(np.random.PCG64(1)), **kw)
    b = synth.render_scan(wall, p1, (0, 0, 0), A, R, rr, np.random.Generator(np.random.PCG64(2)), **kw)
    a2 = synth.render_scan(wall, p0, (0, 0, 0), A, R, rr, np.random.Generator(np.random.PCG64(1)), **kw)
    assert np.array_equal(a, a2)
    ha, hb = _hits_world(a, p0, A, R, rr), _hits_world(b, p1, A, R, rr)
    ha, hb = ha[np.abs(ha[:, 1] - 12.0) < 0.5], hb[np.abs(hb[:, 1] - 12.0) < 0.5]
    assert len(ha) > 30 and len(hb) > 30
    # strong patches along the wall, as sets of 1.5 m patch indices: the two sweeps agree where both look (|x| < 15 m is seen well from both poses)
    pa = set(np.floor((ha[np.abs(ha[:, 0]) < 15, 0] + 40.0) / 1.5).astype(int))
    pb = set(np.floor((hb[np.abs(hb[:, 0]) < 15, 0] + 40.0) / 1.5).astype(int))
    assert len(pa) >= 2 and len(pa ^ pb) <= max(1, len(pa) // 4), (sorted(pa), sorted(pb))
    frac = len(pa) / 20.0  # 20 patches in |x| < 15 m, a fifth of them strong on average
    assert 0.05 <= frac <= 0.5
    # without the texture every azimuth that hits the wall returns strongly: no patches
    c = synth.render_scan(wall, p0, (0, 0, 0), A, R, rr, np.random.Generator(np.random.PCG64(1)), distort=False, hits=1, p_extra=0.0, sigma=1.0)
    hc = _hits_world(c, p0, A, R, rr)
    hc = hc[np.abs(hc[:, 1] - 12.0) < 0.5]
    assert len(set(np.floor((hc[np.abs(hc[:, 0]) < 15, 0] + 40.0) / 1.5).astype(int))) >= 18


def test_street_world_is_the_canyon_with_a_texture():
    w = synth.DriveWorld("street", 3)
    c = synth.DriveWorld("canyon", 3)
    assert w.render.get("texture") == (1.5, 0.2, 0.05) and "texture" not in c.render
    assert w.segs.shape[1:] == (2, 2) and len(w.segs) > 1000
