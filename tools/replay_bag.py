#!/usr/bin/env python3
"""One-command bag run (SURVEY.md §8(f-1)+(f-3)): what launch/localization_bag_play.launch does in the reference —
play a recording into the localization node with a cfg/*.yaml profile, log the realtime / optimized poses in the
reference's text format (localization.cpp:629-645: `stamp x y z qx qy qz qw`, header comment lines), then score them
against the bag's ground-truth topic the way script/evaluate_ate.py does (association within 20 ms, Horn alignment).

    python tools/replay_bag.py BAG CFG.yaml [--uwb anchor.yaml] [--prefix out/run] [--truth-topic /vicon_xb/viconPoseTopic]

Runs on the GPU through the C ABI (loc_node_*): there is no CPU path.  When no --uwb file is given, anchors are taken
from the recording itself (`responder_location` of every UwbRange) and the moving tag is the requester id.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bag")
    ap.add_argument("cfg")
    ap.add_argument("--uwb", help="anchor.yaml-style /uwb block (nodesId, nodesPos, antennaOffset)")
    ap.add_argument("--prefix", default=None, help="log/filename_prefix (default: next to the bag)")
    ap.add_argument("--range-topic", default=None, help="override topic/range (cfg/uwb_only.yaml names /lpsrange, the example bag publishes /uwb_endorange_info)")
    ap.add_argument("--truth-topic", default=None)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--jacobian", default="numeric", choices=["numeric", "analytic"],
                    help="numeric = g2o's central differences, the reference's configuration (default); analytic = the opt-in fast mode")
    a = ap.parse_args()
    import localization_amd as la
    from localization_amd import ate, bag

    cfg = la.load_config(a.cfg, a.uwb)
    evs = list(bag.events(a.bag))
    ranges = [e for e in evs if e["kind"] == "range"]
    topics_in_bag = sorted({e["topic"] for e in evs})
    range_topic = a.range_topic or cfg.topics.get("range")
    if range_topic not in topics_in_bag:
        cands = sorted({e["topic"] for e in ranges})
        if len(cands) != 1:
            sys.exit(f"range topic {range_topic!r} is not in the bag (topics: {topics_in_bag})")
        print(f"# topic/range {range_topic!r} is not in the bag; using {cands[0]!r}", file=sys.stderr)
        range_topic = cands[0]
    imu_topic = cfg.topics.get("imu") if cfg.publish_imu or "imu" in cfg.topics else None
    if imu_topic not in topics_in_bag:
        imu_topic = None
    if not cfg.nodes_id:  # anchors from the recording: responder ids with their surveyed locations; the tag goes last
        anchors = {}
        for e in ranges:
            if e["topic"] == range_topic:
                anchors.setdefault(e["responder_id"], e["responder_location"])
        tag = sorted({e["requester_id"] for e in ranges if e["topic"] == range_topic})
        if len(tag) != 1:
            sys.exit(f"cannot infer the moving tag: requesters {tag}; pass --uwb")
        cfg.nodes_id = sorted(anchors) + tag
        cfg.nodes_pos = [v for i in sorted(anchors) for v in anchors[i]] + [0.0, 0.0, 1.0]
    node = la.LocalizationNode.from_config(cfg, device=a.device, jacobian=a.jacobian)
    import time
    realtime, optimized, n_solved, lat, parts = [], [], 0, [], []
    call_ms = []   # duration of the node call (this harness's ctypes wrapper included) of every message that triggered a solve

    def timed(fn):
        def wrapper(*args, **kw):
            t0 = time.perf_counter()
            o = fn(*args, **kw)
            if o["solved"]:
                call_ms.append((time.perf_counter() - t0) * 1e3)
            return o
        return wrapper
    for name in ("add_range", "add_imu", "add_pose", "add_twist", "add_lidar", "add_rl_range"):
        setattr(node, name, timed(getattr(node, name)))
    it = bag.replay(a.bag, node, range_topic, imu_topic)
    while True:   # the generator runs the node between yields: time from one solve's output to the next = feed + solve
        t0 = time.perf_counter()
        o = next(it, None)
        if o is None:
            break
        lat.append(time.perf_counter() - t0)
        parts.append(node.last_timing())
        n_solved += 1
        if o["published"]:
            realtime.append(o["realtime"]); optimized.append(o["optimized"])
    # Localization::~Localization (localization.cpp:708-717): at shutdown the second half of the window, path[T/2 .. T-1], is
    # appended to the optimized log (path[T/2] therefore appears twice, as in the reference's file)
    # — through the C ABI's loc_node_flush_tail, the call the shim's destructor makes too
    tail = node.flush_tail()
    n_flushed = len(tail)
    optimized.extend(row.copy() for row in tail)
    prefix = a.prefix or os.path.splitext(a.bag)[0]
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    header = [f"iteration_max:{cfg.maximum_iteration}", f"trajectory_length:{cfg.trajectory_length}", f"maximum_velocity:{cfg.maximum_velocity}"]
    files = {}
    for name, rows in (("realtime", realtime), ("optimized", optimized)):
        files[name] = f"{prefix}_{name}.txt"
        ate.write_tum(files[name], np.array(rows).reshape(-1, 8), header=header)
    truth = [e for e in evs if e["kind"] == "truth" and (a.truth_topic is None or e["topic"] == a.truth_topic)]
    rep = {"bag": a.bag, "cfg": a.cfg, "range_topic": range_topic, "imu_topic": imu_topic, "nodes_id": cfg.nodes_id,
           "solves": n_solved, "published": len(realtime), "optimized_rows_flushed_at_exit": n_flushed, "jacobian": a.jacobian, "files": files}
    if len(lat) > 1:   # the first one includes decoding the bag; the reference prints the same figure per solve (localization.cpp:191)
        l = np.array(lat[1:]) * 1e3
        rep["ms_per_solve_incl_feed"] = {"median": float(np.median(l)), "p99": float(np.percentile(l, 99)), "max": float(l.max()),
                                         "budget_ms_between_ranges": 31.0}
        if len(call_ms) > 1:
            c = np.array(call_ms[1:])
            rep["ms_per_solve_node_call"] = {"median": float(np.median(c)), "p99": float(np.percentile(c, 99)), "max": float(c.max()),
                                             "note": "the loc_node_add_* call of a message that triggers a solve, through this harness's ctypes wrapper: what a caller of the library waits for"}
        pt = np.array(parts[1:])
        rep["ms_per_solve_parts_median"] = {"pack_host": float(np.median(pt[:, 0])), "window_solve_call": float(np.median(pt[:, 1])),
                                            "of_which_launch_to_completion": float(np.median(pt[:, 2])),
                                            "note": "inside the library (loc_node_last_timing); the rest of ms_per_solve_incl_feed is this Python harness (decoding the next messages of the bag, ctypes)"}
    if truth and realtime:
        t8 = np.array([[e["stamp"], *e["pose"]] for e in truth])
        for name, rows in (("realtime", realtime), ("optimized", optimized)):
            r = ate.evaluate_ate(np.array(rows), t8)
            rep[f"ate_{name}"] = {k: (float(v) if np.isscalar(v) else v) for k, v in r.items() if k in ("rmse", "mean", "median", "std", "min", "max", "pairs")}
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
