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


def test_world_sweep_host_and_device_entry_points(oracle):
    img = synth.world_scan(synth.World(7), 3, seed=2)
    n = run(img, oracle)
    assert n > 1000
    assert run(img, oracle, device=True) == n


def test_empty_result_and_saturated_image(oracle):
    assert run(np.zeros((16, 512), dtype=np.uint8), oracle) == 0
    run(np.full((16, 512), 255, dtype=np.uint8), oracle)  # uniform: I^2 = mean -> detection iff scaling < 1
