"""Seeded synthetic UWB streams for the BASELINE configs (generator is ours, not the reference's).

cfg2: M = 8 anchors on the corners of a 6 x 6 x 2 m box (superset of the example bag's +-3 m / 0.54-1.97 m
geometry), B tags on seeded random walks (step <= vmax*dt, vmax = 5 m/s from cfg/uwb_only.yaml:4, dt = 1/32 s as in
the bag), ranges = true distance + N(0, 0.05^2) cast to float32 (the wire type of uwb_driver/UwbRange.distance),
distance_err = 0.055 (one of the bag's two values), 1 % NLOS outliers of +U(1, 3) m.
"""
import numpy as np

ANCHORS_8 = np.array([[3.0, -3.0, 0.0], [3.0, 3.0, 2.0], [-3.0, 3.0, 0.0], [-3.0, -3.0, 2.0],
                      [3.0, -3.0, 2.0], [3.0, 3.0, 0.0], [-3.0, 3.0, 2.0], [-3.0, -3.0, 0.0]], dtype=np.float64)
# anchors decoded from the reference's bag/data_example.bag (responder_location of ids 100..103)
ANCHORS_BAG = np.array([[3.0, -3.0, 0.58], [3.0, 3.0, 1.97], [-3.0, 3.0, 0.54], [-3.0, -3.0, 1.76]], dtype=np.float64)

LO = np.array([-2.5, -2.5, 0.3])
HI = np.array([2.5, 2.5, 1.7])


def make_snapshot_stream(B, K, seed=0, anchors=ANCHORS_8, sigma=0.05, err=0.055, outlier_frac=0.01,
                         vmax=5.0, dt=1.0 / 32.0):
    """Returns dict(anchors[M,3], truth[K,3,B] f64, dist[K,M,B] f32, err[K,M,B] f32, init[3,B] f64)."""
    rng = np.random.default_rng(seed)
    M = anchors.shape[0]
    p = rng.uniform(LO[:, None], HI[:, None], size=(3, B))
    truth = np.empty((K, 3, B))
    for k in range(K):
        step = rng.normal(size=(3, B))
        step *= (rng.uniform(0, vmax * dt, size=(1, B)) / np.maximum(np.linalg.norm(step, axis=0, keepdims=True), 1e-12))
        p = np.clip(p + step, LO[:, None], HI[:, None])
        truth[k] = p
    d = np.sqrt(((truth[:, None, :, :] - anchors[None, :, :, None]) ** 2).sum(axis=2))  # [K, M, B]
    d = d + rng.normal(0, sigma, size=d.shape)
    nlos = rng.random(d.shape) < outlier_frac
    d = d + nlos * rng.uniform(1.0, 3.0, size=d.shape)
    dist = d.astype(np.float32)
    errs = np.full(d.shape, err, dtype=np.float32)
    init = np.repeat(anchors.mean(axis=0)[:, None], B, axis=1).astype(np.float64)  # centroid start
    return dict(anchors=np.array(anchors), truth=truth, dist=dist, err=errs, init=init, nlos=nlos)


def make_snapshot_stream_torch(B, K, seed, device, anchors=ANCHORS_8, sigma=0.05, err=0.055, outlier_frac=0.01,
                               vmax=5.0, dt=1.0 / 32.0):
    """Same distribution, generated on `device` directly in the kernel's tile layout.

    Returns dict(dist_tiles[K,M4,B,4] f32, err_tiles[K,M4,B,4] f32, init[3,B] f64 (numpy), truth_last[3,B] (tensor))."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    M = anchors.shape[0]
    M4 = (M + 3) // 4
    a = torch.tensor(anchors, dtype=torch.float64, device=device)          # [M,3]
    lo = torch.tensor(LO, dtype=torch.float64, device=device)[:, None]
    hi = torch.tensor(HI, dtype=torch.float64, device=device)[:, None]
    p = lo + (hi - lo) * torch.rand((3, B), generator=g, device=device, dtype=torch.float64)
    dist_tiles = torch.zeros((K, M4, B, 4), dtype=torch.float32, device=device)
    err_tiles = torch.zeros((K, M4, B, 4), dtype=torch.float32, device=device)
    for k in range(K):
        step = torch.randn((3, B), generator=g, device=device, dtype=torch.float64)
        step = step * (torch.rand((1, B), generator=g, device=device, dtype=torch.float64) * (vmax * dt)
                       / step.norm(dim=0, keepdim=True).clamp_min(1e-12))
        p = torch.minimum(torch.maximum(p + step, lo), hi)
        d = (p[None, :, :] - a[:, :, None]).norm(dim=1)                     # [M,B]
        d = d + sigma * torch.randn((M, B), generator=g, device=device, dtype=torch.float64)
        nlos = torch.rand((M, B), generator=g, device=device) < outlier_frac
        d = d + nlos * (1.0 + 2.0 * torch.rand((M, B), generator=g, device=device, dtype=torch.float64))
        full = torch.zeros((M4 * 4, B), dtype=torch.float32, device=device)
        full[:M] = d.to(torch.float32)
        e = torch.zeros((M4 * 4, B), dtype=torch.float32, device=device)
        e[:M] = err
        dist_tiles[k] = full.view(M4, 4, B).permute(0, 2, 1)
        err_tiles[k] = e.view(M4, 4, B).permute(0, 2, 1)
    init = np.repeat(np.asarray(anchors).mean(axis=0)[:, None], B, axis=1).astype(np.float64)
    return dict(dist_tiles=dist_tiles.contiguous(), err_tiles=err_tiles.contiguous(), init=init, truth_last=p)


def make_fusion_stream(B, K, seed=0, anchors=ANCHORS_8, offset=(0.1, 0.0, -0.05), sigma=0.05, err=0.055, outlier_frac=0.01,
                       vmax=3.0, dt=1.0 / 32.0, imu_cov=4.592449e-06):
    """BASELINE config 3 stream: as make_snapshot_stream plus a true attitude random walk, the antenna
    lever arm on the tag side, and IMU quaternions = truth x small-angle N(0, imu_cov) noise; cfg/uwb_imu.yaml values
    (vmax = 3 m/s).  Returns dict(anchors, offset, dist[K,M,B] f32, err f32, imu[K,B,8] f64, init[7,B], truth_t[K,3,B],
    truth_q[K,B,4] xyzw)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    off = np.asarray(offset, dtype=np.float64)
    p = rng.uniform(LO[:, None], HI[:, None], size=(3, B))
    rv = rng.normal(0, 0.3, size=(B, 3))
    truth_t = np.empty((K, 3, B)); truth_q = np.empty((K, B, 4)); imu = np.zeros((K, B, 8))
    dist = np.empty((K, anchors.shape[0], B), dtype=np.float32)
    for k in range(K):
        step = rng.normal(size=(3, B))
        step *= (rng.uniform(0, vmax * dt, size=(1, B)) / np.maximum(np.linalg.norm(step, axis=0, keepdims=True), 1e-12))
        p = np.clip(p + step, LO[:, None], HI[:, None])
        rv = rv + rng.normal(0, 0.02, size=(B, 3))
        Rk = Rotation.from_rotvec(rv)
        truth_t[k] = p; truth_q[k] = Rk.as_quat()
        ant = p.T + Rk.apply(off)                                           # [B,3]
        d = np.sqrt(((ant[None, :, :] - anchors[:, None, :]) ** 2).sum(axis=2))  # [M,B]
        d = d + rng.normal(0, sigma, size=d.shape)
        d = d + (rng.random(d.shape) < outlier_frac) * rng.uniform(1.0, 3.0, size=d.shape)
        dist[k] = d.astype(np.float32)
        qi = (Rk * Rotation.from_rotvec(rng.normal(0, np.sqrt(imu_cov), size=(B, 3)))).as_quat()
        imu[k, :, :4] = qi; imu[k, :, 4:7] = imu_cov
    errs = np.full(dist.shape, err, dtype=np.float32)
    init = np.zeros((7, B)); init[:3] = np.asarray(anchors).mean(axis=0)[:, None]; init[6] = 1.0
    return dict(anchors=np.array(anchors), offset=off, dist=dist, err=errs, imu=imu, init=init, truth_t=truth_t, truth_q=truth_q)
