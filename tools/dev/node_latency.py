"""per-message latency of the drop-in node on the example recording (fixture), split into its parts (dev tool)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import localization_amd as la
bag = np.load(os.path.join(ROOT, "tests", "golden", "bag_example.npz"))
ids = list(bag["anchor_ids"]) + [200]
pos = np.concatenate([bag["anchor_pos"], [[0.0, 0.0, 1.0]]])
for jac in ("numeric", "analytic"):
    node = la.LocalizationNode(ids, pos, trajectory_length=10, maximum_velocity=5.0, distance_outlier=1.0, maximum_iteration=10, minimum_optimize_error=2000.0,
                               publish_range=True, jacobian=jac)
    call, parts = [], []
    for i in range(600):
        t0 = time.perf_counter()
        o = node.add_range(200, int(bag["uwb_responder"][i]), float(bag["uwb_stamp"][i]), float(bag["uwb_distance"][i]), float(bag["uwb_distance_err"][i]), 1, "uwb")
        dt = time.perf_counter() - t0
        if o["solved"]:
            call.append(dt * 1e3); parts.append(node.last_timing())
    p = np.array(parts[20:]); c = np.array(call[20:])
    print(f"{jac}: add_range call median {np.median(c):.3f} ms (p99 {np.percentile(c, 99):.3f}); pack {np.median(p[:,0]):.3f}  window call {np.median(p[:,1]):.3f}  kernel {np.median(p[:,2]):.3f}")
    node.close()
