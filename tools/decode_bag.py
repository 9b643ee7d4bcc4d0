#!/usr/bin/env python3
"""Decode the reference's example recording (rosbag v2.0) WITHOUT ROS into a small numeric fixture:
tests/golden/bag_example.npz.  Run in the build container only (the reference tree does not travel):

    python tools/decode_bag.py /root/reference/bag/data_example.bag tests/golden/bag_example.npz

The bag is a DATA file of the reference (inputs + Vicon ground truth); nothing but numbers is extracted.
The reader itself is localization_amd/bag.py (message layouts come from the definitions embedded in the bag).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localization_amd import bag


def main(path, out):
    conns, msgs = bag.read_bag(path)
    print("connections:", {c.topic: c.msg_type for c in conns.values()})
    uwb, imu, vic = [], [], []
    frames = {}
    # file order (not record-time order): the fixture keeps both stamps so tests can sort either way
    for cid, t_rec, m in msgs:
        c = conns[cid]
        if c.msg_type == "uwb_driver/UwbRange":
            e = bag.decode_uwb_range(m)
            uwb.append((e["stamp"], t_rec, e["requester_id"], e["responder_id"], e["distance"], e["distance_err"], e["antenna"]) + tuple(e["responder_location"]))
            frames["uwb"] = e["frame_id"]
        elif c.msg_type == "sensor_msgs/Imu":
            e = bag.decode_imu(m)
            cov = e["orientation_covariance"]
            imu.append((e["stamp"], t_rec) + tuple(e["q_xyzw"]) + (cov[0], cov[4], cov[8]))
            frames["imu"] = e["frame_id"]
        elif "vicon" in c.msg_type.lower():
            e = bag.decode_header_pose(m)
            vic.append((e["stamp"], t_rec) + tuple(e["pose"]))
    uwb = np.array(uwb); imu = np.array(imu); vic = np.array(vic)
    anchors = {int(r[3]): tuple(r[7:10]) for r in uwb}
    ids = sorted(anchors)
    np.savez_compressed(
        out,
        uwb_stamp=uwb[:, 0], uwb_rectime=uwb[:, 1], uwb_requester=uwb[:, 2].astype(np.int32),
        uwb_responder=uwb[:, 3].astype(np.int32), uwb_distance=uwb[:, 4].astype(np.float32),
        uwb_distance_err=uwb[:, 5].astype(np.float32), uwb_antenna=uwb[:, 6].astype(np.int32),
        anchor_ids=np.array(ids, dtype=np.int32), anchor_pos=np.array([anchors[i] for i in ids]),
        imu_stamp=imu[:, 0], imu_rectime=imu[:, 1], imu_q_xyzw=imu[:, 2:6], imu_orientation_cov_diag=imu[:, 6:9],
        vicon_stamp=vic[:, 0], vicon_rectime=vic[:, 1], vicon_pos=vic[:, 2:5], vicon_q_xyzw=vic[:, 5:9],
        frame_uwb=np.array(frames.get("uwb", "")), frame_imu=np.array(frames.get("imu", "")),
    )
    print("uwb", uwb.shape, "imu", imu.shape, "vicon", vic.shape, frames, anchors)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
