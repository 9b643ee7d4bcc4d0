"""Test helper: write a rosbag v2.0 file (one chunk, optional bz2 / lz4 compression) from the decoded-bag fixture, with
the message layouts the reader decodes (uwb_driver/UwbRange md5 1b3efd63…, sensor_msgs/Imu, a Header + Pose truth
message).  Lets GPU-box tests run file-level tools without the reference's recording."""
import bz2
import struct

import numpy as np


def _rec(hdr, data):
    h = b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k + b"=" + v for k, v in hdr.items())
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def _stamp(t):
    secs = int(np.floor(t)); nsecs = int(round((t - secs) * 1e9))
    if nsecs >= 1000000000: secs += 1; nsecs -= 1000000000
    return secs, nsecs


def _header(seq, t, frame):
    f = frame.encode()
    return struct.pack("<III", seq, *_stamp(t)) + struct.pack("<I", len(f)) + f


def uwb_range_msg(seq, t, frame, requester, responder, distance, distance_err, antenna, responder_location):
    return (_header(seq, t, frame) + struct.pack("<BBBB", requester, 0, responder, 0) + struct.pack("<HHHH", 0, 0, 0, 0) +
            struct.pack("<ffff", distance, distance_err, 0.0, 0.0) + struct.pack("<B", antenna) + struct.pack("<H", 0) +
            struct.pack("<I", 0) + struct.pack("<ddd", *responder_location))


def imu_msg(seq, t, frame, q_xyzw, cov_diag):
    cov = [cov_diag[0], 0, 0, 0, cov_diag[1], 0, 0, 0, cov_diag[2]]
    return _header(seq, t, frame) + struct.pack("<4d", *q_xyzw) + struct.pack("<9d", *cov) + bytes(8 * 24)


def truth_msg(seq, t, frame, pos, q_xyzw):
    return _header(seq, t, frame) + struct.pack("<7d", *pos, *q_xyzw)


def write_bag(path, fixture, n_ranges=None, compression="none", range_topic="/uwb_endorange_info", imu_topic="/imu/data",
              truth_topic="/vicon_xb/viconPoseTopic"):
    z = fixture
    n = len(z["uwb_stamp"]) if n_ranges is None else n_ranges
    t_end = z["uwb_rectime"][n - 1]
    apos = {int(i): p for i, p in zip(z["anchor_ids"], z["anchor_pos"])}
    conns = [(0, range_topic, b"uwb_driver/UwbRange", b"1b3efd633e416bfcfbaaf891dd23ac23", b"Header header\nuint8 requester_id\n"),
             (1, imu_topic, b"sensor_msgs/Imu", b"6a62c6daae103f4ff57a132d6f95cec2", b"Header header\ngeometry_msgs/Quaternion orientation\n"),
             (2, truth_topic, b"geometry_msgs/PoseStamped", b"d3812c3cbc69362b77dc0b19b345f8f5", b"Header header\ngeometry_msgs/Pose pose\n")]
    inner = b""
    for cid, topic, typ, md5, definition in conns:
        data = b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k + b"=" + v for k, v in
                        {b"topic": topic.encode(), b"type": typ, b"md5sum": md5, b"message_definition": definition}.items())
        inner += _rec({b"op": b"\x07", b"conn": struct.pack("<I", cid), b"topic": topic.encode()}, data)
    msgs = []
    for i in range(n):
        msgs.append((z["uwb_rectime"][i], 0, uwb_range_msg(i, z["uwb_stamp"][i], str(z["frame_uwb"]), int(z["uwb_requester"][i]),
                     int(z["uwb_responder"][i]), float(z["uwb_distance"][i]), float(z["uwb_distance_err"][i]), int(z["uwb_antenna"][i]),
                     apos[int(z["uwb_responder"][i])])))
    for i in np.nonzero(z["imu_rectime"] <= t_end)[0]:
        msgs.append((z["imu_rectime"][i], 1, imu_msg(int(i), z["imu_stamp"][i], str(z["frame_imu"]), z["imu_q_xyzw"][i], z["imu_orientation_cov_diag"][i])))
    for i in np.nonzero(z["vicon_rectime"] <= t_end)[0]:
        msgs.append((z["vicon_rectime"][i], 2, truth_msg(int(i), z["vicon_stamp"][i], "world", z["vicon_pos"][i], z["vicon_q_xyzw"][i])))
    msgs.sort(key=lambda m: m[0])
    for t, cid, data in msgs:
        inner += _rec({b"op": b"\x02", b"conn": struct.pack("<I", cid), b"time": struct.pack("<II", *_stamp(t))}, data)
    if compression == "bz2":
        payload = bz2.compress(inner)
    elif compression == "lz4":
        import pyarrow as pa
        payload = pa.compress(inner, codec="lz4", asbytes=True)
    else:
        payload = inner
    chunk = _rec({b"op": b"\x05", b"compression": compression.encode(), b"size": struct.pack("<I", len(inner))}, payload)
    with open(path, "wb") as f:
        f.write(b"#ROSBAG V2.0\n" + _rec({b"op": b"\x03", b"index_pos": struct.pack("<Q", 0), b"conn_count": struct.pack("<I", len(conns)),
                                          b"chunk_count": struct.pack("<I", 1)}, bytes(16)) + chunk)
    return len(msgs)
