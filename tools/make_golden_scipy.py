#!/usr/bin/env python3
"""Golden vectors from an INDEPENDENT optimiser (scipy 1.15 least_squares), not from the reference (g2o is not
installable here; the reference ships no vectors).  Each case stores the inputs and the minimiser of the SAME
objective g2o's LM monitors:  sum_robust ln(1 + e^T Omega e) + sum_plain e^T Omega e   (SURVEY.md Appendix A.4).

    python tools/make_golden_scipy.py tests/golden/scipy_minima.npz

Cases: (a) 3-DoF snapshots, 8 anchors, NLOS outliers; (b) 6-DoF snapshots with an antenna lever arm and an
IMU-style rotation-only prior (localization.cpp:333-334, :499-525); (c) 5-pose windows in the reference's topology
(one range per pose + zero-range smoothness edges, localization.cpp:331-340).
"""
import sys

import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

ANCH8 = np.array([[3.0, -3.0, 0.0], [3.0, 3.0, 2.0], [-3.0, 3.0, 0.0], [-3.0, -3.0, 2.0],
                  [3.0, -3.0, 2.0], [3.0, 3.0, 0.0], [-3.0, 3.0, 2.0], [-3.0, -3.0, 0.0]])
ANCH4 = np.array([[3.0, -3.0, 0.58], [3.0, 3.0, 1.97], [-3.0, 3.0, 0.54], [-3.0, -3.0, 1.76]])


def rob(z):  # residual whose square is ln(1 + z^2)
    return np.sign(z) * np.sqrt(np.log1p(z * z))


def solve(fun, x0):
    r = least_squares(fun, x0, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=5000)
    return r.x, 2 * r.cost


def main(out):
    rng = np.random.default_rng(2026)
    res = {}
    # ---- (a) 3-DoF snapshots --------------------------------------------------------------------------------
    N = 48
    truth = np.stack([rng.uniform(-2.5, 2.5, N), rng.uniform(-2.5, 2.5, N), rng.uniform(0.3, 1.7, N)], 1)
    d = np.linalg.norm(truth[:, None, :] - ANCH8[None], axis=2) + rng.normal(0, 0.05, (N, 8))
    d += (rng.random((N, 8)) < 0.08) * rng.uniform(1, 3, (N, 8))
    d = d.astype(np.float32)
    s = np.where(rng.random((N, 8)) < 0.5, 0.055, 0.024).astype(np.float32)
    x0 = truth + rng.normal(0, 0.15, truth.shape)
    xs, cs = [], []
    for i in range(N):
        f = lambda p: rob((d[i].astype(float) - np.linalg.norm(p[None] - ANCH8, axis=1)) / s[i].astype(float))
        x, c = solve(f, x0[i]); xs.append(x); cs.append(c)
    res.update(a_anchors=ANCH8, a_dist=d, a_err=s, a_init=x0, a_min=np.array(xs), a_cost=np.array(cs))
    # ---- (b) 6-DoF snapshots: lever arm + rotation-only prior ---------------------------------------------------
    N = 24
    off = np.array([0.10, 0.0, -0.05])
    cov = 4.592449e-06
    truth_t = np.stack([rng.uniform(-2, 2, N), rng.uniform(-2, 2, N), rng.uniform(0.5, 1.5, N)], 1)
    truth_R = Rotation.from_rotvec(rng.normal(0, 0.4, (N, 3)))
    imu_R = truth_R * Rotation.from_rotvec(rng.normal(0, np.sqrt(cov) * 2, (N, 3)))
    ant = truth_t + truth_R.apply(off)
    d = (np.linalg.norm(ant[:, None, :] - ANCH8[None], axis=2) + rng.normal(0, 0.05, (N, 8))).astype(np.float32)
    s = np.full((N, 8), 0.055, np.float32)
    t0 = truth_t + rng.normal(0, 0.1, truth_t.shape)
    mins_t, mins_q, cs = [], [], []
    for i in range(N):
        Zinv = imu_R[i].inv()
        def f(x):
            R = Rotation.from_rotvec(x[3:])
            a = x[:3] + R.apply(off)
            r1 = rob((d[i].astype(float) - np.linalg.norm(a[None] - ANCH8, axis=1)) / s[i].astype(float))
            q = (Zinv * R).as_quat()          # x y z w
            q = q * (1.0 if q[3] >= 0 else -1.0)
            r2 = q[:3] / np.sqrt(cov)         # prior: toVectorMQT(Z^-1 X) rotation part, Omega = 1/cov, not robust
            return np.concatenate([r1, r2])
        x, c = solve(f, np.concatenate([t0[i], imu_R[i].as_rotvec()]))
        mins_t.append(x[:3]); mins_q.append(Rotation.from_rotvec(x[3:]).as_quat()); cs.append(c)
    res.update(b_anchors=ANCH8, b_offset=off, b_dist=d, b_err=s, b_init_t=t0, b_imu_q_xyzw=imu_R.as_quat(),
               b_cov=np.array(cov), b_min_t=np.array(mins_t), b_min_q_xyzw=np.array(mins_q), b_cost=np.array(cs))
    # ---- (c) 5-pose windows, reference topology ------------------------------------------------------------------
    N, T = 16, 5
    sig_v = 5.0 * (1 / 32.0) / 3.0
    mins, cs, inits, dd, aid = [], [], [], [], []
    for i in range(N):
        p = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.8, 1.4)])
        traj = [p]
        for _ in range(T - 1):
            traj.append(traj[-1] + rng.normal(0, 0.03, 3))
        traj = np.array(traj)
        ids = np.arange(T) % 4
        dist = (np.linalg.norm(traj - ANCH4[ids], axis=1) + rng.normal(0, 0.03, T)).astype(np.float32)
        # a fixed prior pose in front of the window keeps it well posed: pose -1 is a fixed vertex here
        prev = traj[0] + rng.normal(0, 0.02, 3)
        x0 = (traj + rng.normal(0, 0.05, traj.shape)).ravel()
        def f(x):
            P = x.reshape(T, 3)
            r = [rob((dist.astype(float) - np.linalg.norm(P - ANCH4[ids], axis=1)) / 0.055)]
            Q = np.vstack([prev[None], P])
            r.append(rob(-np.linalg.norm(Q[1:] - Q[:-1], axis=1) / sig_v))
            # two extra anchors seen from the newest pose make the window observable
            r.append(rob((np.linalg.norm(traj[-1] - ANCH4[[1, 2]], axis=1) - np.linalg.norm(P[-1] - ANCH4[[1, 2]], axis=1)) / 0.055))
            return np.concatenate(r)
        x, c = solve(f, x0)
        mins.append(x.reshape(T, 3)); cs.append(c); inits.append(x0.reshape(T, 3)); dd.append(dist); aid.append(ids)
        res.setdefault("c_prev", []).append(prev); res.setdefault("c_extra", []).append(np.linalg.norm(traj[-1] - ANCH4[[1, 2]], axis=1))
    res.update(c_anchors=ANCH4, c_dist=np.array(dd), c_anchor_idx=np.array(aid), c_init=np.array(inits),
               c_min=np.array(mins), c_cost=np.array(cs), c_sigma_v=np.array(sig_v),
               c_prev=np.array(res["c_prev"]), c_extra=np.array(res["c_extra"]))
    np.savez_compressed(out, **res)
    print({k: np.asarray(v).shape for k, v in res.items()})


if __name__ == "__main__":
    main(sys.argv[1])
