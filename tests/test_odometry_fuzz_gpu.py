"""Batched odometry on pathological sweeps, device vs the oracle's fuser, sweep by sweep: uniform noise (4800 scattered points,
thousands of occupied voxels, registrations that find little), saturated and nearly blank sweeps in between real ones, sweeps
whose strongest returns are all ties. Cell counts, keyframe counts, poses and iteration counts agree, and a failed
registration is the same failure.

Rounds 1-2 did not assert equal iteration counts once a sequence had gone through an ill-posed registration (fewer than 30
residuals, or a solve that ran into the iteration limit: a saturated sweep is a ring of points at maximum range, 11 residual
blocks, rotation unobservable, 21 iterations per solve): one 21-21-21-21-15 against -14 had been seen. Since round 3 every one
of the 56 ill-posed registrations of these sequences agrees with the oracle in outer and inner iteration counts, residual count,
keyframe count and pose (tests/run_fuzz_report.py lists them), so nothing is exempt any more."""
import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import capi, synth

pytestmark = pytest.mark.gpu
RR = np.float32(0.0595238)


def mk(mod, **kw):
    base = dict(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, compensate=1, radar_ccw=0, cost=1, loss=1,
                loss_limit=0.1, submap_scan_size=4)
    base.update(kw)
    return mod.default_params(**base)


def sequences():
    rng = np.random.default_rng(99)
    world, _ = synth.world_sequence(8, seed=17, world_seed=31)
    A, R = world.shape[1:]
    noise = rng.integers(0, 256, (8, A, R), dtype=np.uint8)
    faint = rng.integers(0, 70, (8, A, R), dtype=np.uint8)  # a few bins above z_min per azimuth
    blank = np.zeros((A, R), dtype=np.uint8)
    blank[17, 900:960] = 180  # one bearing with a few returns: a cloud of 12 points, no cells (an empty cloud is refused by the oracle)
    sat = np.full((A, R), 255, dtype=np.uint8)
    ties = np.where(rng.random((A, R)) < 0.02, 200, 10).astype(np.uint8)  # every return has the same intensity
    out = {
        "world": world,  # control: well-posed to the end
        "noise": noise,
        "faint_noise": faint,
        "blank_between": np.stack([world[0], blank, world[2], world[3], blank, blank, world[6], world[7]]),
        "saturated_between": np.stack([world[0], world[1], sat, world[3], world[4], sat, world[6], world[7]]),
        "ties": np.stack([ties, np.roll(ties, 3, axis=1), world[2], np.roll(ties, 7, axis=1), world[4], world[5], ties, world[7]]),
        "noise_then_world": np.stack([noise[0], noise[1], world[2], world[3], world[4], noise[5], world[6], world[7]]),
    }
    return out


@pytest.fixture(scope="module")
def seqs():
    return sequences()


@pytest.mark.parametrize("cost", [1, 2])
def test_pathological_sweeps_match_the_oracle_fuser(oracle, seqs, cost):
    SEQS = seqs
    names = sorted(SEQS)
    kw = dict(cost=cost, regularization=0.1, covar_scale=1.0)
    po, pg = mk(oracle, **kw), mk(capi, **kw)
    ctx = capi.Context(pg, 400, 3360)
    odo = ctx.odometry(len(names))
    fus = [oracle.Fuser(po) for _ in names]
    ill_posed = 0
    for t in range(8):
        odo.step_host(np.stack([SEQS[n][t] for n in names]))
        got = odo.poses()
        for q, n in enumerate(names):
            exp = fus[q].process_polar(SEQS[n][t])
            S, nc, nk = odo.summary(q)
            So = fus[q].last_summary()
            assert nc == len(fus[q].last_cells()), (n, t, nc, len(fus[q].last_cells()))
            if t > 0 and (So.num_residuals < 30 or max(So.inner_iterations[:8]) > 20):
                ill_posed += 1  # (asserted like every other registration)
            assert nk == fus[q].num_keyframes, (n, t, nk, fus[q].num_keyframes)
            assert (S.usable, S.outer_iterations, list(S.inner_iterations[:8]), S.num_residuals) == \
                   (So.usable, So.outer_iterations, list(So.inner_iterations[:8]), So.num_residuals), (n, t)
            assert np.all(np.abs(got[q][:2] - exp[:2]) < 1e-4) and abs(got[q][2] - exp[2]) < 1e-5, (n, t, got[q], exp)
    assert ill_posed >= 20  # the sequences really are pathological
    odo.release()
    ctx.close()


@pytest.mark.parametrize("k", [1, 2, 3, 64])
def test_batched_odometry_at_the_ends_of_the_k_range(oracle, k):
    """k_strongest = 1 used to map every slot of the batched cloud pass to bearing 0 (the reciprocal of the slot -> bearing
    division is 2^32 for k = 1 and did not fit); the per-call path divides and never showed it. Clouds (through the cell counts),
    keyframes and poses of the batched path against the oracle's fuser at both ends of the supported range."""
    imgs, _ = synth.world_sequence(6, seed=5, world_seed=77)
    kw = dict(k_strongest=k, res=3.5)
    po, pg = mk(oracle, **kw), mk(capi, **kw)
    ctx = capi.Context(pg, 400, 3360)
    odo = ctx.odometry(2)
    fus = [oracle.Fuser(po), oracle.Fuser(po)]
    streams = [imgs, imgs[::-1].copy()]
    for t in range(6):
        odo.step_host(np.stack([s[t] for s in streams]))
        got = odo.poses()
        for q in range(2):
            exp = fus[q].process_polar(streams[q][t])
            S, nc, nk = odo.summary(q)
            So = fus[q].last_summary()
            assert nc == len(fus[q].last_cells()), (k, t, q, nc, len(fus[q].last_cells()))
            if t > 0 and So.num_residuals >= 30 and max(So.inner_iterations[:8]) <= 20:
                assert (nk, S.outer_iterations, S.num_residuals) == (fus[q].num_keyframes, So.outer_iterations, So.num_residuals), (k, t, q)
                assert np.all(np.abs(got[q][:2] - exp[:2]) < 1e-4) and abs(got[q][2] - exp[2]) < 1e-5, (k, t, q, got[q], exp)
    if k > 1:
        assert len(fus[0].last_cells()) > 20
    odo.release()
    ctx.close()


def test_batched_odometry_with_more_bearings_than_the_cloud_pass_tabulates(oracle):
    """1000 azimuths: the batched cloud pass keeps per-bearing values (angle, compensation rotation, cos / sin) for up to 810
    bearings in LDS and evaluates Compensate as written (atan2, sincos per point) beyond that; 12 000 slots per sweep also take
    its looped branch and the general feature path. Device vs the oracle's fuser, sweep by sweep."""
    A, R = 1000, 1680
    rr = np.float32(2 * 0.0595238)
    imgs, _ = synth.world_sequence(6, A=A, R=R, range_res=rr, seed=3, world_seed=41)
    kw = dict(range_res=rr, res=3.5)
    po, pg = mk(oracle, **kw), mk(capi, **kw)
    ctx = capi.Context(pg, A, R)
    odo = ctx.odometry(2)
    fus = [oracle.Fuser(po), oracle.Fuser(po)]
    streams = [imgs, imgs[::-1].copy()]
    for t in range(6):
        odo.step_host(np.stack([s[t] for s in streams]))
        got = odo.poses()
        for q in range(2):
            exp = fus[q].process_polar(streams[q][t])
            S, nc, nk = odo.summary(q)
            So = fus[q].last_summary()
            assert nc == len(fus[q].last_cells()), (t, q, nc, len(fus[q].last_cells()))
            if t > 0:
                assert (nk, S.outer_iterations, S.num_residuals) == (fus[q].num_keyframes, So.outer_iterations, So.num_residuals), (t, q)
                assert list(S.inner_iterations[:8]) == list(So.inner_iterations[:8]), (t, q)
                assert np.all(np.abs(got[q][:2] - exp[:2]) < 1e-4) and abs(got[q][2] - exp[2]) < 1e-5, (t, q, got[q], exp)
    assert len(fus[0].last_cells()) > 50
    odo.release()
    ctx.close()
