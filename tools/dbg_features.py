import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfear_radarodometry_code_public_amd import capi, synth
from oracle import binding as ob
RR = np.float32(0.0595238)
imgs, gt = synth.world_sequence(3, seed=3)
po = ob.default_params(range_res=RR, res=3.0, weight_intensity=1)
pg = capi.default_params(range_res=RR, res=3.0, weight_intensity=1)
ctx = capi.Context(pg, 400, 3360)
xyi0 = ob.compensate(ob.cloud(ob.filter_polar(imgs[2], 60, 12), RR, 2.5), [1.0, 0.01, 0.02], 0)
for nlone_rows in (0, 5, 20, 30, 40, 50):
    rng = np.random.default_rng(11)
    keep = xyi0[rng.permutation(len(xyi0))[:2100]]
    gx, gy = np.meshgrid(np.arange(-26, 26), np.arange(-25, -25 + nlone_rows))
    lone = np.stack([gx.ravel() * 3.0 + 1.1 + rng.uniform(-0.9, 0.9, gx.size), gy.ravel() * 3.0 + 1.3 + rng.uniform(-0.9, 0.9, gx.size),
                     rng.integers(61, 200, gx.size)], axis=1).astype(np.float32).reshape(-1, 3)
    xyi = np.concatenate([keep, lone])[rng.permutation(2100 + gx.size)]
    so = ob.Scan(xyi, po)
    sg = ctx.scan_create(ctx.cloud_upload(xyi))
    print("lone rows", nlone_rows, "points", len(xyi), "oracle samples", len(so.samples()), "cells", so.size, "gpu cells", sg.size, flush=True)
