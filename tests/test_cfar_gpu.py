"""HIP CA-CFAR (csrc/cfar.hip) against the oracle through the C ABI: bit-exact clouds."""
import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import capi, synth

pytestmark = pytest.mark.gpu
RR = np.float32(0.0595238)


def run(img, oracle, zmin=60.0, mind=2.5, window=10, guard=20, pfa=0.01, device=False):
    A, R = img.shape
    ctx = capi.Context(capi.default_params(range_res=RR, z_min=zmin, min_distance=mind), A, R)
    if device:
        import torch
        d = torch.from_numpy(img).cuda()
        cloud = ctx.filter_cfar(d, window, guard, pfa)
    else:
        cloud = ctx.filter_cfar(img, window, guard, pfa)
    got = cloud.download()
    exp = oracle.cfar(img, RR, zmin, mind, window, guard, pfa)
    assert got.shape == exp.shape, (got.shape, exp.shape)
    assert np.array_equal(got, exp)
    return len(exp)


@pytest.mark.parametrize("A,R", [(400, 3360), (400, 3768), (37, 1001), (8, 64), (3, 16000)])
def test_random_images_bit_exact(oracle, A, R):
    rng = np.random.default_rng(A + R)
    img = rng.integers(0, 256, size=(A, R), dtype=np.uint8)
    img[:, ::11] = np.minimum(img[:, ::11].astype(int) + 100, 255).astype(np.uint8)
    assert run(img, oracle) > 0 or R < 100


@pytest.mark.parametrize("window,guard,pfa,mind,zmin", [(40, 5, 0.01, 2.5, 60.0), (3, 0, 0.2, 0.0, 0.0), (60, 30, 0.001, 1.0, 10.0), (1, 1, 0.5, 2.5, 0.0)])
def test_parameter_sweep_bit_exact(oracle, window, guard, pfa, mind, zmin):
    rng = np.random.default_rng(window * 100 + guard)
    img = rng.integers(0, 256, size=(64, 777), dtype=np.uint8)
    run(img, oracle, zmin, mind, window, guard, pfa)


@pytest.mark.parametrize("R", [776, 3360, 6000])
@pytest.mark.parametrize("window,guard,pfa,mind,zmin", [(40, 5, 0.01, 2.5, 60.0), (3, 0, 0.2, 0.0, 0.0), (60, 30, 0.001, 1.0, 10.0), (1, 1, 0.5, 2.5, 0.0),
                                                       (500, 10, 0.0001, 2.5, 20.0), (40, 10, 0.01, 0.0, 20.0)])
def test_parameter_sweep_rows_of_whole_dwords(oracle, R, window, guard, pfa, mind, zmin):
    """rows whose length is a multiple of four take cfar_detect_fast_kernel (16 bins per thread up to 4096 bins, 32 up to 8192):
    windows longer than the row's ends, a range gate that starts at bin 0, windows of one bin, and - the second image - plateaus of
    equal bytes, where I^2 sits on the threshold to the last bit of the float pre-test (the reference's double arithmetic decides)"""
    rng = np.random.default_rng(window * 100 + guard + R)
    img = rng.integers(0, 256, size=(24, R), dtype=np.uint8)
    img[:, -40:] = np.maximum(img[:, -40:], 180)  # strong returns inside the clipped windows of the far end
    img[:, :60] = np.maximum(img[:, :60], 150)
    run(img, oracle, zmin, mind, window, guard, pfa)
    levels = np.array([0, 40, 40, 80, 120, 200], dtype=np.uint8)
    img = np.repeat(levels[rng.integers(0, len(levels), size=(24, R // 8))], 8, axis=1)
    run(np.ascontiguousarray(img), oracle, zmin, mind, window, guard, pfa)


@pytest.mark.parametrize("R", [3360, 3768, 1792])
def test_rows_of_clutter(oracle, R):
    """hundreds of detections per row: a wave of the owner-layout detector (csrc/cfar.hip, round 6) collects more candidates than the 64 its one-pass
    decision takes and more hits than a segment's record slot holds - the threads' hit masks go out and the emit kernel walks them - next to quiet rows
    that take the record route, in one image; 1792 bins = exactly the 64 x 28 bins two waves hold (the row then takes the next wider shape)"""
    rng = np.random.default_rng(R)
    img = rng.integers(0, 12, size=(12, R), dtype=np.uint8)
    img[::2, ::8] = 255          # a comb: every spike is a detection
    img[1::4, 100:103] = 200     # quiet rows with one return
    img[3, :] = 90
    img[3, ::5] = 140            # spikes over a plateau: some are hits, some not
    n = run(img, oracle, zmin=20.0, window=40, guard=10)
    assert n > 6 * (R // 8 - 20)
    run(img, oracle, zmin=20.0, window=500, guard=10, pfa=0.0001)


def test_random_configurations_bit_exact(oracle):
    """sixty random (row length, window, guard, false-alarm rate, static threshold, range gate) combinations on small images of three kinds - random bytes
    with a comb, plateaus, a noise floor with blobs: every shape of the owner-layout detector (28 / 20 / 16 / 32 bins per thread, rows that are and are
    not a whole number of segments), every choice of its integer scale and row offsets, windows longer than the row, and the round-5 kernels for rows
    of odd length - against the oracle (its prefix-sum twin: tests/test_cfar_cpu.py holds that against the literal detector)"""
    rng = np.random.default_rng(20260929)
    ran = 0
    for it in range(60):
        R = int(rng.choice([64, 200, 776, 1000, 1792, 2048, 3360, 3584, 3768, 4096, 5000, 8000, 1001, 3361]))
        A = int(rng.integers(3, 9))
        window = int(rng.choice([1, 3, 10, 40, 70, 100, 150, 300, 500, 900]))
        guard = int(rng.choice([0, 1, 5, 10, 20, 37]))
        pfa = float(rng.choice([0.3, 0.1, 0.01, 0.001, 0.0001]))
        zmin = float(rng.choice([0.0, 20.0, 60.0, 100.0]))
        mind = float(rng.choice([0.0, 2.5, 30.0]))
        kind = it % 3
        if kind == 0:
            img = rng.integers(0, 256, size=(A, R), dtype=np.uint8)
            img[:, ::7] = np.minimum(img[:, ::7].astype(int) + 120, 255).astype(np.uint8)
        elif kind == 1:
            levels = np.array([0, 40, 40, 80, 120, 200], dtype=np.uint8)
            img = np.ascontiguousarray(np.repeat(levels[rng.integers(0, len(levels), size=(A, R // 8 + 1))], 8, axis=1)[:, :R])
        else:
            img = np.clip(rng.normal(25.0, 8.0, size=(A, R)), 0, 255).astype(np.uint8)
            for a in range(A):
                for c in rng.integers(0, R, size=6):
                    lo, hi = max(0, c - 3), min(R, c + 4)
                    img[a, lo:hi] = np.maximum(img[a, lo:hi], rng.integers(90, 255))
        ctx = capi.Context(capi.default_params(range_res=RR, z_min=zmin, min_distance=mind), A, R)
        got = ctx.filter_cfar(img, window, guard, pfa).download()
        exp = oracle.cfar(img, RR, zmin, mind, window, guard, pfa, prefix=True)
        assert got.shape == exp.shape and np.array_equal(got, exp), (it, R, A, window, guard, pfa, zmin, mind, kind, got.shape, exp.shape)
        ran += len(exp) > 0
        ctx.close()
    assert ran > 30


def test_world_sweep_host_and_device_entry_points(oracle):
    img = synth.world_scan(synth.World(7), 3, seed=2)
    n = run(img, oracle)
    assert n > 1000
    assert run(img, oracle, device=True) == n


def test_empty_result_and_saturated_image(oracle):
    assert run(np.zeros((16, 512), dtype=np.uint8), oracle) == 0
    run(np.full((16, 512), 255, dtype=np.uint8), oracle)  # uniform: I^2 = mean -> detection iff scaling < 1


@pytest.mark.parametrize("A,R,n", [(400, 3360, 5), (37, 1001, 7)])
def test_batched_device_entry_matches_per_image_clouds(oracle, A, R, n):
    """cfear_filter_cfar_batch_device: n images back to back, per image a cloud slot of `capacity` points and the true
    detection count - every image's cloud bit-identical to the oracle's, a too small capacity truncates without overrun."""
    import torch
    rng = np.random.default_rng(A * 7 + R)
    imgs = rng.integers(0, 256, size=(n, A, R), dtype=np.uint8)
    imgs[:, :, ::11] = np.minimum(imgs[:, :, ::11].astype(int) + 100, 255).astype(np.uint8)
    imgs[2] = synth.world_scan(synth.World(5), 3, A, R) if A == 400 else imgs[2] // 3  # a sparse image in the middle of the batch
    exp = [oracle.cfar(imgs[i], RR, 60.0, 2.5, 10, 20, 0.01) for i in range(n)]
    cap = max(len(e) for e in exp) + 5
    ctx = capi.Context(capi.default_params(range_res=RR, z_min=60.0, min_distance=2.5), A, R)
    d = torch.from_numpy(imgs).cuda()
    for capacity in (cap, max(cap // 3, 1)):
        cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
        flat = torch.full((n * capacity * 3 + 3,), -7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.filter_cfar_batch(d, n, flat, capacity, cnt)
        ctx.synchronize()
        got_n = cnt.cpu().numpy()
        got = flat.cpu().numpy()
        assert np.all(got[n * capacity * 3:] == -7.0)  # nothing past the last slot
        for i in range(n):
            assert got_n[i] == len(exp[i])
            m = min(len(exp[i]), capacity)
            slot = got[i * capacity * 3:(i + 1) * capacity * 3].reshape(capacity, 3)
            assert np.array_equal(slot[:m], exp[i][:m]), i
            assert np.all(slot[m:] == -7.0)
    ctx.close()
