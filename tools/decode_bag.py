#!/usr/bin/env python3
"""Decode the reference's example recording (rosbag v2.0, uncompressed chunks) WITHOUT ROS into a small numeric
fixture: tests/golden/bag_example.npz.  Run in the build container only (the reference tree does not travel):

    python tools/decode_bag.py /root/reference/bag/data_example.bag tests/golden/bag_example.npz

The bag is a DATA file of the reference (inputs + Vicon ground truth); nothing but numbers is extracted.
Message layouts come from the message definitions embedded in the bag's connection records.
"""
import struct
import sys

import numpy as np


def read_records(buf, pos, end):
    while pos < end:
        (hlen,) = struct.unpack_from("<I", buf, pos); pos += 4
        hdr = {}
        hend = pos + hlen
        while pos < hend:
            (flen,) = struct.unpack_from("<I", buf, pos); pos += 4
            field = buf[pos:pos + flen]; pos += flen
            k, v = field.split(b"=", 1)
            hdr[k.decode()] = v
        (dlen,) = struct.unpack_from("<I", buf, pos); pos += 4
        data = buf[pos:pos + dlen]; pos += dlen
        yield hdr, data


def parse_header(msg, o):
    seq, secs, nsecs = struct.unpack_from("<III", msg, o); o += 12
    (n,) = struct.unpack_from("<I", msg, o); o += 4
    frame = msg[o:o + n].decode(); o += n
    return seq, secs + nsecs * 1e-9, frame, o


def main(path, out):
    buf = open(path, "rb").read()
    assert buf.startswith(b"#ROSBAG V2.0\n")
    conns = {}
    msgs = []
    for hdr, data in read_records(buf, 13, len(buf)):
        op = hdr["op"][0]
        if op == 0x05:  # chunk
            assert hdr["compression"] == b"none", hdr["compression"]
            for h2, d2 in read_records(data, 0, len(data)):
                op2 = h2["op"][0]
                if op2 == 0x07:
                    cid = struct.unpack("<I", h2["conn"])[0]
                    ch = dict(read_conn_header(d2))
                    conns[cid] = (h2["topic"].decode(), ch)
                elif op2 == 0x02:
                    cid = struct.unpack("<I", h2["conn"])[0]
                    secs, nsecs = struct.unpack("<II", h2["time"])
                    msgs.append((cid, secs + nsecs * 1e-9, d2))
        elif op == 0x07:
            cid = struct.unpack("<I", hdr["conn"])[0]
            conns.setdefault(cid, (hdr["topic"].decode(), dict(read_conn_header(data))))
    topics = {cid: t for cid, (t, _) in conns.items()}
    types = {cid: ch.get("type", b"").decode() for cid, (_, ch) in conns.items()}
    print("connections:", {topics[c]: types[c] for c in conns})
    for cid, (t, ch) in conns.items():
        if "vicon" in t:
            print(ch["message_definition"].decode())

    uwb, imu, vic = [], [], []
    frames = {}
    for cid, t_rec, m in msgs:
        ty = types[cid]
        if ty == "uwb_driver/UwbRange":
            seq, stamp, frame, o = parse_header(m, 0)
            rq, rqi, rs, rsi = struct.unpack_from("<BBBB", m, o); o += 4
            o += 8  # 4 x u16
            dist, derr, ddot, ddoterr = struct.unpack_from("<ffff", m, o); o += 16
            (ant,) = struct.unpack_from("<B", m, o); o += 1
            o += 2 + 4
            lx, ly, lz = struct.unpack_from("<ddd", m, o); o += 24
            assert o == len(m), (o, len(m))
            uwb.append((stamp, t_rec, rq, rs, dist, derr, ant, lx, ly, lz))
            frames["uwb"] = frame
        elif ty == "sensor_msgs/Imu":
            seq, stamp, frame, o = parse_header(m, 0)
            q = struct.unpack_from("<dddd", m, o); o += 32
            cov = struct.unpack_from("<9d", m, o); o += 72
            imu.append((stamp, t_rec) + q + (cov[0], cov[4], cov[8]))
            frames["imu"] = frame
        elif "vicon" in ty.lower():
            seq, stamp, frame, o = parse_header(m, 0)
            # geometry_msgs/Pose pose (+ whatever follows): position xyz, orientation xyzw
            vals = struct.unpack_from("<7d", m, o)
            vic.append((stamp, t_rec) + vals)
            frames["vicon"] = frame
    uwb = np.array(uwb); imu = np.array(imu); vic = np.array(vic)
    print("uwb", uwb.shape, "imu", imu.shape, "vicon", vic.shape, frames)
    anchors = {}
    for r in uwb:
        anchors[int(r[3])] = tuple(r[7:10])
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
    print("anchors", anchors)
    print("requester ids", set(uwb[:, 2].astype(int)), "antenna", set(uwb[:, 6].astype(int)),
          "err", set(np.round(uwb[:, 5], 4)))


def read_conn_header(data):
    pos = 0
    while pos < len(data):
        (flen,) = struct.unpack_from("<I", data, pos); pos += 4
        field = data[pos:pos + flen]; pos += flen
        k, v = field.split(b"=", 1)
        yield k.decode(), v


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
