"""rosbag v2.0 reader without ROS: the step BEFORE the hot path — turns a recording into the
time-ordered range / IMU / pose event stream the node front-end consumes (reference wiring:
launch/localization_bag_play.launch:12, localization_node.cpp:52-88).

Supports uncompressed, bz2 and lz4 chunks (lz4 through the plain-Python frame decoder in lz4frame.py).  Only the message
types the path uses are decoded: uwb_driver/UwbRange, sensor_msgs/Imu, geometry_msgs/PoseWithCovarianceStamped,
geometry_msgs/TwistWithCovarianceStamped and any message that starts with Header + geometry_msgs/Pose (Vicon truth).
"""
import bz2
import struct
from dataclasses import dataclass
from typing import Dict, Iterator, List, Tuple

from . import lz4frame


@dataclass
class Connection:
    conn_id: int
    topic: str
    msg_type: str
    md5sum: str
    definition: str


def _records(buf, pos, end):
    while pos < end:
        (hlen,) = struct.unpack_from("<I", buf, pos); pos += 4
        hdr = {}
        hend = pos + hlen
        while pos < hend:
            (flen,) = struct.unpack_from("<I", buf, pos); pos += 4
            k, v = bytes(buf[pos:pos + flen]).split(b"=", 1); pos += flen
            hdr[k.decode()] = v
        (dlen,) = struct.unpack_from("<I", buf, pos); pos += 4
        yield hdr, buf[pos:pos + dlen]
        pos += dlen


def _fields(data):
    pos, out = 0, {}
    while pos < len(data):
        (flen,) = struct.unpack_from("<I", data, pos); pos += 4
        k, v = bytes(data[pos:pos + flen]).split(b"=", 1); pos += flen
        out[k.decode()] = v
    return out


def read_bag(path) -> Tuple[Dict[int, Connection], List[Tuple[int, float, bytes]]]:
    """Returns (connections, messages) with messages = [(conn_id, record_time, serialized bytes)] in file order."""
    buf = memoryview(open(path, "rb").read())
    if bytes(buf[:13]) != b"#ROSBAG V2.0\n":
        raise ValueError("not a rosbag v2.0 file")
    conns: Dict[int, Connection] = {}
    msgs: List[Tuple[int, float, bytes]] = []

    def add_conn(h, d):
        cid = struct.unpack("<I", h["conn"])[0]
        f = _fields(d)
        conns.setdefault(cid, Connection(cid, h["topic"].decode(), f.get("type", b"").decode(), f.get("md5sum", b"").decode(),
                                         f.get("message_definition", b"").decode(errors="replace")))

    for hdr, data in _records(buf, 13, len(buf)):
        op = hdr["op"][0]
        if op == 0x05:
            comp = hdr.get("compression", b"none")
            if comp == b"none":
                chunk = data
            elif comp == b"bz2":
                chunk = memoryview(bz2.decompress(bytes(data)))
            elif comp == b"lz4":
                size = struct.unpack("<I", hdr["size"])[0] if "size" in hdr else None
                chunk = memoryview(lz4frame.decompress(data, size))
            else:
                raise NotImplementedError(f"chunk compression {comp!r}")
            for h2, d2 in _records(chunk, 0, len(chunk)):
                op2 = h2["op"][0]
                if op2 == 0x07:
                    add_conn(h2, d2)
                elif op2 == 0x02:
                    secs, nsecs = struct.unpack("<II", h2["time"])
                    msgs.append((struct.unpack("<I", h2["conn"])[0], secs + nsecs * 1e-9, bytes(d2)))
        elif op == 0x07:
            add_conn(hdr, data)
    return conns, msgs


def _header(m, o=0):
    seq, secs, nsecs = struct.unpack_from("<III", m, o); o += 12
    (n,) = struct.unpack_from("<I", m, o); o += 4
    frame = m[o:o + n].decode(errors="replace"); o += n
    return seq, secs + nsecs * 1e-9, frame, o


def decode_uwb_range(m):
    """uwb_driver/UwbRange (md5 1b3efd633e416bfcfbaaf891dd23ac23), packed little-endian."""
    seq, stamp, frame, o = _header(m)
    rq, rqi, rs, rsi = struct.unpack_from("<BBBB", m, o); o += 4
    o += 8
    dist, derr, ddot, ddoterr = struct.unpack_from("<ffff", m, o); o += 16
    (ant,) = struct.unpack_from("<B", m, o); o += 1
    o += 6
    loc = struct.unpack_from("<ddd", m, o)
    return dict(kind="range", stamp=stamp, frame_id=frame, requester_id=rq, responder_id=rs, distance=dist,
                distance_err=derr, antenna=ant, responder_location=loc)


def decode_imu(m):
    seq, stamp, frame, o = _header(m)
    q = struct.unpack_from("<dddd", m, o); o += 32
    cov = struct.unpack_from("<9d", m, o)
    return dict(kind="imu", stamp=stamp, frame_id=frame, q_xyzw=q, orientation_covariance=cov)


def decode_pose_cov(m):
    seq, stamp, frame, o = _header(m)
    pose = struct.unpack_from("<7d", m, o); o += 56
    cov = struct.unpack_from("<36d", m, o)
    return dict(kind="pose", stamp=stamp, frame_id=frame, pose=pose, covariance=cov)


def decode_twist_cov(m):
    seq, stamp, frame, o = _header(m)
    tw = struct.unpack_from("<6d", m, o); o += 48
    cov = struct.unpack_from("<36d", m, o)
    return dict(kind="twist", stamp=stamp, frame_id=frame, twist=tw, covariance=cov)


def decode_header_pose(m):
    seq, stamp, frame, o = _header(m)
    return dict(kind="truth", stamp=stamp, frame_id=frame, pose=struct.unpack_from("<7d", m, o))


DECODERS = {
    "uwb_driver/UwbRange": decode_uwb_range,
    "sensor_msgs/Imu": decode_imu,
    "geometry_msgs/PoseWithCovarianceStamped": decode_pose_cov,
    "geometry_msgs/TwistWithCovarianceStamped": decode_twist_cov,
}


def events(path, topics=None) -> Iterator[dict]:
    """Time-ordered (by record time, like `rosbag play`) decoded events of the message types the path uses."""
    conns, msgs = read_bag(path)
    out = []
    for cid, t, data in msgs:
        c = conns[cid]
        if topics is not None and c.topic not in topics:
            continue
        dec = DECODERS.get(c.msg_type)
        if dec is None and "geometry_msgs/Pose pose" in c.definition and c.definition.lstrip().startswith("Header"):
            dec = decode_header_pose
        if dec is None:
            continue
        ev = dec(data)
        ev["topic"] = c.topic; ev["record_time"] = t
        out.append(ev)
    out.sort(key=lambda e: e["record_time"])
    return iter(out)


def replay(path, node, range_topic, imu_topic=None, self_filter=None):
    """Feed a bag into a LocalizationNode the way the reference's subscribers would (localization_node.cpp:52-88).
    Yields the node's output dict after every message that triggered a solve."""
    topics = {range_topic} | ({imu_topic} if imu_topic else set())
    for ev in events(path, topics):
        if ev["kind"] == "range":
            o = node.add_range(ev["requester_id"], ev["responder_id"], ev["stamp"], ev["distance"], ev["distance_err"],
                               ev["antenna"], ev["frame_id"])
        elif ev["kind"] == "imu":
            o = node.add_imu(ev["stamp"], ev["q_xyzw"], ev["orientation_covariance"], ev["frame_id"])
        else:
            continue
        if o["solved"]:
            yield o
