"""An independent statement of the registration (association, P2L / P2D / P2P residuals, scaled Huber loss, the trust-region
Levenberg-Marquardt schedule and the outer re-association loop), written in numpy from SURVEY.md section 9 F-I - not from
oracle/cfear_oracle.c - and run on the cells of oracle scans. It must make the same decisions as the C oracle: the same number
of outer iterations, the same number of inner iterations in every solve, the same residual counts, and the same poses and
costs to rounding. This pins the oracle's *transcription* of the schedule (a slip in the C code shows up as a different
iteration count); what it cannot pin is the reading of Ceres / FLANN itself (the [3P] rows of DESIGN.md section 2)."""
import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import synth

RR = np.float32(0.0595238)
COS30 = 0.86602540378443864676


def aff(p):
    c, s = np.cos(p[2]), np.sin(p[2])
    return np.array([[c, -s], [s, c]]), np.array([p[0], p[1]])


def associate(cells_tar, cells_src, Ttar, Tsrc, radius):
    """q = (Ttar^-1 Tsrc) u_src in double -> float; 1-NN over the float target means with float arithmetic; accepted
    iff d2 < radius^2; gate (R_rel n_src) . n_tar > cos(pi / 6). -> list of (j_src, i_tar, sim)"""
    Rt, tt = Ttar
    Rs, ts = Tsrc
    Rrel, trel = Rt.T @ Rs, Rt.T @ (ts - tt)
    mt = cells_tar["mean"].astype(np.float32)
    out = []
    for j in range(len(cells_src)):
        q = (Rrel @ cells_src["mean"][j] + trel).astype(np.float32)
        dx, dy = q[0] - mt[:, 0], q[1] - mt[:, 1]
        d2 = dx * dx
        d2 = d2 + dy * dy  # float32, one rounding per operation
        i = int(np.argmin(d2))  # first minimum = lowest index among exact ties
        if float(d2[i]) < radius * radius:
            sim = max(float((Rrel @ cells_src["normal"][j]) @ cells_tar["normal"][i]), 0.0)
            if sim > COS30:
                out.append((j, i, sim))
    return out


def similarity(a, b):
    return 2.0 * min(a, b) / (a + b)


def build(scans, poses, itr, cost, weight_opt):
    """residual blocks in (keyframe, source cell) order."""
    n = len(scans)
    Tsrc = aff(poses[n - 1])
    radius = 4.0 if itr == 1 else 2.0
    blocks = []
    for i in range(n - 1):
        Ttar = aff(poses[i])
        for (j, ti, sim) in associate(scans[i], scans[n - 1], Ttar, Tsrc, radius):
            cs, ct = scans[n - 1][j], scans[i][ti]
            w = {0: 1.0, 1: similarity(cs["nsamples"], ct["nsamples"]), 2: sim, 3: similarity(cs["scale"], ct["scale"])}.get(weight_opt)
            if weight_opt == 4:
                w = similarity(float(cs["nsamples"]), float(ct["nsamples"])) + sim + similarity(cs["scale"], ct["scale"])
            if w is None:
                w = 1.0
            Rt, tt = Ttar
            m = Rt @ ct["mean"] + tt
            b = {"s": cs["mean"].copy(), "m": m, "w": w}
            if cost == 1:
                b["n"] = Rt @ ct["normal"]
            elif cost == 2:
                C = np.array([[ct["cov"][0], ct["cov"][1]], [ct["cov"][1], ct["cov"][2]]])
                S = (0.1 * np.eye(2) + Rt @ C @ Rt.T) * 1.0  # regularization 0.1, covar_scale 1 (the context defaults)
                b["L"] = np.linalg.cholesky(np.linalg.inv(S))
            blocks.append(b)
    return blocks


def evaluate(blocks, x, cost, a):
    """robustified cost 1/2 sum w rho_H(s), gradient and Gauss-Newton matrix (corrector sqrt(rho'))"""
    c, s = np.cos(x[2]), np.sin(x[2])
    R = np.array([[c, -s], [s, c]])
    dR = np.array([[-s, -c], [c, -s]])
    f, g, H = 0.0, np.zeros(3), np.zeros((3, 3))
    for b in blocks:
        p = R @ b["s"] + x[:2]
        dth = dR @ b["s"]
        if cost == 1:
            r = np.array([(p - b["m"]) @ b["n"]])
            J = np.array([[b["n"][0], b["n"][1], b["n"] @ dth]])
        elif cost == 2:
            r = b["L"] @ (p - b["m"])
            J = b["L"] @ np.column_stack([np.eye(2), dth])
        else:
            r = b["m"] - p
            J = -np.column_stack([np.eye(2), dth])
        sq = float(r @ r)
        if sq > a * a:
            rho, d1 = 2 * a * np.sqrt(sq) - a * a, a / np.sqrt(sq)
        else:
            rho, d1 = sq, 1.0
        f += 0.5 * b["w"] * rho
        sr = np.sqrt(d1 * b["w"])
        rt, Jt = sr * r, sr * J
        g += Jt.T @ rt
        H += Jt.T @ Jt
    return f, g, H


def solve(blocks, x0, cost, a, max_inner=20):
    """Ceres trust-region LM, defaults. -> (x, iterations (incl. the initial evaluation), final_cost, last rho, termination)"""
    x = np.array(x0, dtype=np.float64)
    f, g, H = evaluate(blocks, x, cost, a)
    iters, final_cost, last_rho = 1, f, 0.0
    if np.max(np.abs(g)) <= 1e-10:
        return x, iters, final_cost, last_rho, 0
    sc = 1.0 / (1.0 + np.sqrt(np.diag(H)))  # Jacobi scaling, once
    radius, dec, reuse, invalid, it = 1e4, 2.0, False, 0, 0
    dg = np.zeros(3)
    while True:
        if it >= max_inner:
            return x, iters, final_cost, last_rho, 1
        if radius < 1e-32:
            return x, iters, final_cost, last_rho, 0
        it += 1
        Hs, gs = H * np.outer(sc, sc), g * sc
        if not reuse:
            dg = np.clip(np.diag(Hs), 1e-6, 1e32)
        A = Hs + np.diag(dg / radius)
        reuse = True
        ok = True
        try:
            y = np.linalg.solve(A, -gs)
            np.linalg.cholesky(A)
        except np.linalg.LinAlgError:
            ok = False
        if ok:
            mcc = -(y @ gs + 0.5 * y @ Hs @ y)
            ok = mcc > 0
        if not ok:
            invalid += 1
            if invalid >= 5:
                return x, iters, final_cost, last_rho, 2
            radius /= dec
            dec *= 2
            iters += 1
            last_rho = 0.0
            continue
        invalid = 0
        xc = x + y * sc
        fc, gc, Hc = evaluate(blocks, xc, cost, a)
        if np.linalg.norm(x - xc) <= 1e-8 * (np.linalg.norm(x) + 1e-8):
            return x, iters, final_cost, last_rho, 0
        change = f - fc
        if abs(change) <= 1e-6 * f:
            return x, iters, final_cost, last_rho, 0
        rho = change / mcc
        iters += 1
        last_rho = rho
        if rho > 1e-3:
            x, f, g, H = xc, fc, gc, Hc
            t = 2.0 * rho - 1.0
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - t * t * t))
            dec, reuse = 2.0, False
            final_cost = min(final_cost, f)
            if it >= max_inner:
                return x, iters, final_cost, last_rho, 1
            if np.max(np.abs(g)) <= 1e-10:
                return x, iters, final_cost, last_rho, 0
        else:
            radius /= dec
            dec *= 2
            reuse = True
            final_cost = min(final_cost, fc)


def register(scans, poses, cost, weight_opt, a=0.1, max_outer=8, min_itr=3):
    """the outer re-association loop. -> (poses, outer iterations as documented, [inner iterations], [residuals], [costs])"""
    poses = np.array([[p[0], p[1], np.arctan2(np.sin(p[2]), np.cos(p[2]))] for p in poses], dtype=np.float64)
    n = len(scans)
    prev_par, prev_score = poses[n - 1].copy(), np.finfo(np.float64).max
    inner, nres, costs = [], [], []
    itr = 1
    while itr <= max_outer:
        blocks = build(scans, poses, itr, cost, weight_opt)
        res = len(blocks) * (1 if cost == 1 else 2)
        if res <= 1:
            return None
        x, iters, final_cost, last_rho, term = solve(blocks, poses[n - 1], cost, a)
        if term != 2:
            poses[n - 1] = x
        inner.append(iters); nres.append(res); costs.append(final_cost)
        brk = False
        if itr > min_itr:
            rel = (prev_score - final_cost) / prev_score
            if prev_score < final_cost:
                poses[n - 1] = prev_par
                brk = True
            elif rel < 1e-5:
                brk = True
            elif last_rho < 1e-5 or iters == 1:
                brk = True
        if brk:
            break
        prev_score, prev_par = final_cost, poses[n - 1].copy()
        itr += 1
    return poses, itr, inner, nres, costs


@pytest.fixture(scope="module")
def world_scans(oracle):
    imgs, gts = synth.world_sequence(4, seed=77, world_seed=4711)
    p = oracle.default_params(range_res=RR, res=3.0, weight_intensity=1)
    scans = []
    for t in range(4):
        xyi = oracle.cloud(oracle.filter_polar(imgs[t], 60, 12), RR, 2.5)
        scans.append(oracle.Scan(xyi, p))
    return scans, gts


@pytest.mark.parametrize("cost,weight_opt,nkf", [(1, 4, 3), (1, 0, 1), (2, 4, 2), (0, 2, 2), (1, 3, 3)])
def test_python_statement_makes_the_same_decisions_as_the_oracle(oracle, world_scans, cost, weight_opt, nkf):
    all_scans, gt = world_scans
    scans = all_scans[:nkf] + [all_scans[3]]
    poses = np.array([list(g) for g in gt[:nkf]] + [[gt[3][0] - 0.25, gt[3][1] + 0.15, gt[3][2] - 0.01]])
    p = oracle.default_params(range_res=RR, res=3.0, weight_intensity=1, cost=cost, loss=1, loss_limit=0.1, weight_opt=weight_opt)
    ret, P, cov, S = oracle.register(scans, poses, p)
    cells = [s.cells() for s in scans]
    got = register(cells, poses, cost, weight_opt)
    assert got is not None and S.usable == 1
    gp, itr, inner, nres, costs = got
    n_solves = len(inner)
    assert itr == S.outer_iterations, (itr, S.outer_iterations)
    assert inner == list(S.inner_iterations[:n_solves]), (inner, list(S.inner_iterations[:n_solves]))
    assert nres[-1] == S.num_residuals
    assert np.allclose(costs, list(S.outer_cost[:n_solves]), rtol=1e-9, atol=1e-12)
    assert np.allclose(gp[-1], P[-1], rtol=0, atol=1e-9), (gp[-1], P[-1])
