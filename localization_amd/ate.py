"""Absolute trajectory error, the step AFTER the hot path: what the reference's manual evaluation
loop computes with script/associate.py:71-101 (greedy nearest-stamp association within max_difference = 20 ms) and
script/evaluate_ate.py:47-79,152-162 (Horn alignment, RMSE / mean / median of the translational error), plus the
TUM-format writer the node uses for its logs (localization.cpp:630-642).  Pure numpy; the Python-2 originals do not
run under Python 3.
"""
import numpy as np


def write_tum(path, poses8, header=None):
    """rows: stamp x y z qx qy qz qw  — '%.9f' on the stamp, default float formatting otherwise (localization.cpp:633-641).
    Appends, like save_file; with `header` (list of strings) the file is started afresh with '# ' comment lines, like
    set_file (localization.cpp:647-668)."""
    if header is not None:
        with open(path, "w") as f:
            for h in header:
                f.write("# " + h + "\n")
    with open(path, "a") as f:
        for p in np.asarray(poses8):
            f.write("%.9f" % p[0] + " " + " ".join("%g" % v for v in p[1:]) + "\n")


def read_tum(path):
    rows = []
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        rows.append([float(v) for v in line.replace(",", " ").split()])
    return np.array(rows)


def associate(stamps_a, stamps_b, offset=0.0, max_difference=0.02):
    """Greedy best-first matching of two stamp lists (associate.py:71-101): EVERY pair with |a - (b + offset)| <
    max_difference is a candidate, candidates are taken in ascending (difference, a, b) order, each stamp is used once.
    Returns index pairs sorted by a."""
    a = np.asarray(stamps_a, dtype=float); b = np.asarray(stamps_b, dtype=float)
    order_b = np.argsort(b + offset, kind="stable")
    bs = (b + offset)[order_b]
    lo = np.searchsorted(bs, a - max_difference, side="left")
    hi = np.searchsorted(bs, a + max_difference, side="right")
    cand = []
    for i, t in enumerate(a):
        for jj in range(lo[i], hi[i]):
            j = int(order_b[jj])
            diff = abs(t - (b[j] + offset))
            if diff < max_difference:
                cand.append((diff, t, b[j], i, j))
    cand.sort()
    used_a, used_b, out = set(), set(), []
    for _, _, _, i, j in cand:
        if i in used_a or j in used_b:
            continue
        used_a.add(i); used_b.add(j); out.append((i, j))
    out.sort()
    return out


def horn_align(model, data):
    """Rigid alignment of model onto data (3xN each), closed form (evaluate_ate.py:47-79). Returns R, t, per-point error."""
    model = np.asarray(model, dtype=float); data = np.asarray(data, dtype=float)
    mz = model - model.mean(axis=1, keepdims=True)
    dz = data - data.mean(axis=1, keepdims=True)
    W = mz @ dz.T
    U, _, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    R = U @ S @ Vh
    t = data.mean(axis=1, keepdims=True) - R @ model.mean(axis=1, keepdims=True)
    err = np.sqrt(((R @ model + t - data) ** 2).sum(axis=0))
    return R, t, err


def evaluate_ate(estimate8, truth8, offset=0.0, max_difference=0.02, align=True):
    """estimate / truth: rows stamp x y z ...  Returns dict(pairs, rmse, mean, median, std, min, max) in metres."""
    est = np.asarray(estimate8, dtype=float); tru = np.asarray(truth8, dtype=float)
    pairs = associate(tru[:, 0], est[:, 0], offset, max_difference)
    if len(pairs) < 2:
        raise ValueError("Couldn't find matching timestamp pairs between groundtruth and estimated trajectory")
    a = tru[[i for i, _ in pairs], 1:4].T
    b = est[[j for _, j in pairs], 1:4].T
    if align:
        _, _, err = horn_align(b, a)
    else:
        err = np.sqrt(((b - a) ** 2).sum(axis=0))
    return dict(pairs=len(pairs), rmse=float(np.sqrt((err ** 2).mean())), mean=float(err.mean()), median=float(np.median(err)),
                std=float(err.std()), min=float(err.min()), max=float(err.max()))
