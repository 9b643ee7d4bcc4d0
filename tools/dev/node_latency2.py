"""where the per-message time of the Python harness goes (dev tool): raw ctypes call vs wrapper"""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import localization_amd as la
from localization_amd.node import NodeOutput
bag = np.load(os.path.join(ROOT, "tests", "golden", "bag_example.npz"))
ids = list(bag["anchor_ids"]) + [200]
pos = np.concatenate([bag["anchor_pos"], [[0.0, 0.0, 1.0]]])
node = la.LocalizationNode(ids, pos, trajectory_length=10, maximum_velocity=5.0, distance_outlier=1.0, maximum_iteration=10, minimum_optimize_error=2000.0, publish_range=True)
L, h = node.L, node.h
resp = [int(x) for x in bag["uwb_responder"][:800]]; st = [float(x) for x in bag["uwb_stamp"][:800]]
d = [float(x) for x in bag["uwb_distance"][:800]]; de = [float(x) for x in bag["uwb_distance_err"][:800]]
o = NodeOutput(); ref = C.byref(o); fr = b"uwb"
raw, parts = [], []
for i in range(800):
    t0 = time.perf_counter()
    rc = L.loc_node_add_range(h, 200, resp[i], st[i], d[i], de[i], 1, fr, ref)
    dt = time.perf_counter() - t0
    if o.solved:
        raw.append(dt * 1e3); parts.append(node.last_timing())
raw = np.array(raw[20:]); p = np.array(parts[20:])
print(f"raw ctypes call median {np.median(raw):.3f} ms; inside: pack {np.median(p[:,0]):.3f} window call {np.median(p[:,1]):.3f} kernel {np.median(p[:,2]):.3f}; unaccounted {np.median(raw - p[:,0] - p[:,1]):.3f}")
