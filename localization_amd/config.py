"""Parameter surface of the reference node (cfg/*.yaml, read at reference localization.cpp:58-159).

Same keys, same defaults.  ``load_config`` reads a reference-style yaml (robot/*, optimizer/*, topic/*,
publish_flag/*, frame/*) and, optionally, an ``anchor.yaml``-style /uwb block (nodesId, nodesPos, antennaOffset;
reference README.md:51-59).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import yaml


@dataclass
class LocalizationConfig:
    # robot/*   (localization.cpp:72-79)
    trajectory_length: Optional[int] = None   # no default in the reference (getParam, :72)
    maximum_velocity: float = 1.0
    distance_outlier: float = 1.0
    # optimizer/*   (:58-69)
    maximum_iteration: int = 20
    minimum_optimize_error: float = 1000.0
    verbose: bool = False
    # topic/*  (localization_node.cpp:52-88)
    topics: Dict[str, str] = field(default_factory=dict)
    # publish_flag/*  (:140-159) — all false by default
    publish_tf: bool = False
    publish_range: bool = False
    publish_pose: bool = False
    publish_twist: bool = False
    publish_lidar: bool = False
    publish_imu: bool = False
    publish_relative_range: bool = False
    # frame/*  (:134-138)
    frame_target: str = "estimation"
    frame_source: str = "local_origin"
    # /uwb/*  (:83-123)
    nodes_id: List[int] = field(default_factory=list)
    nodes_pos: List[float] = field(default_factory=list)
    antenna_offset: Optional[List[float]] = None

    @property
    def has_relative_range(self):
        return "relative_range" in self.topics  # n.hasParam("topic/relative_range"), :94


def load_config(path, uwb_path=None, **overrides):
    with open(path) as f:
        y = yaml.safe_load(f) or {}
    c = LocalizationConfig()
    robot = y.get("robot", {}) or {}
    opt = y.get("optimizer", {}) or {}
    pf = y.get("publish_flag", {}) or {}
    fr = y.get("frame", {}) or {}
    if "trajectory_length" in robot: c.trajectory_length = int(robot["trajectory_length"])
    if "maximum_velocity" in robot: c.maximum_velocity = float(robot["maximum_velocity"])
    if "distance_outlier" in robot: c.distance_outlier = float(robot["distance_outlier"])
    if "maximum_iteration" in opt: c.maximum_iteration = int(opt["maximum_iteration"])
    if "minimum_optimize_error" in opt: c.minimum_optimize_error = float(opt["minimum_optimize_error"])
    if "verbose" in opt: c.verbose = bool(opt["verbose"])
    c.topics = dict(y.get("topic", {}) or {})
    for k in ("tf", "range", "pose", "twist", "lidar", "imu", "relative_range"):
        if k in pf: setattr(c, "publish_" + k, bool(pf[k]))
    if "target" in fr: c.frame_target = str(fr["target"])
    if "source" in fr: c.frame_source = str(fr["source"])
    if uwb_path:
        with open(uwb_path) as f:
            u = yaml.safe_load(f) or {}
        u = u.get("uwb", u)
        c.nodes_id = [int(i) for i in u.get("nodesId", [])]
        c.nodes_pos = [float(v) for v in u.get("nodesPos", [])]
        if "antennaOffset" in u: c.antenna_offset = [float(v) for v in u["antennaOffset"]]
    for k, v in overrides.items():
        if not hasattr(c, k):
            raise KeyError(k)
        setattr(c, k, v)
    return c
