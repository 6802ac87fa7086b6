"""ingest.py — replaying the reference's recorded logs into the pipeline.

The flight logs the reference ships (`flight_experiments/hgtData.yaml`, `optical_flow_experiments/BeispielDaten…/imuData.yaml`)
are yaml dumps of ROS message OBJECTS: `!!python/object/new:sensor_msgs.msg._Imu.Imu` nodes whose `state:` list holds the
message's slots in declaration order.  The reference reads them with `yaml.load` and the ROS classes installed
(evaluate_exp.py:50-58, of_library.py:327-351), which executes constructors named by the file.  `load_ros_yaml` parses the
same files with a SafeLoader whose only extra constructors build plain records — nothing named in the file is imported or
called — and `associate` hands the nearest-timestamp association of evaluate_exp.py:68-95 to the device
(`ofk_associate_sensors`).
"""
import types

import numpy as np

# slots of the message types that occur in the logs, in genpy declaration order (= order of the `state:` list)
SLOTS = {
    "Header": ("seq", "stamp", "frame_id"),
    "Time": ("secs", "nsecs"),
    "Duration": ("secs", "nsecs"),
    "Quaternion": ("x", "y", "z", "w"),
    "Vector3": ("x", "y", "z"),
    "Point": ("x", "y", "z"),
    "Imu": ("header", "orientation", "orientation_covariance", "angular_velocity", "angular_velocity_covariance",
            "linear_acceleration", "linear_acceleration_covariance"),
    "Range": ("header", "radiation_type", "field_of_view", "min_range", "max_range", "range"),
    "CompressedImage": ("header", "format", "data"),
}


class RosMsg(types.SimpleNamespace):
    """Plain record standing in for a ROS message object: attribute access like the reference's code uses it."""


def _loader():
    import yaml
    base = getattr(yaml, "CSafeLoader", yaml.SafeLoader)

    class Loader(base):
        pass

    def construct_new(loader, suffix, node):
        m = loader.construct_mapping(node, deep=True) if isinstance(node, yaml.MappingNode) else {}
        state = m.get("state", m.get("args", []))
        name = suffix.rsplit(".", 1)[-1]
        if isinstance(state, dict):                                 # `!!python/object:` form: attribute mapping
            return RosMsg(_type=suffix, **state)
        slots = SLOTS.get(name)
        if slots is None or len(slots) != len(state):
            return RosMsg(_type=suffix, state=list(state))
        return RosMsg(_type=suffix, **dict(zip(slots, state)))

    def construct_obj(loader, suffix, node):
        m = loader.construct_mapping(node, deep=True) if isinstance(node, yaml.MappingNode) else {}
        return RosMsg(_type=suffix, **{str(k): v for k, v in m.items()})

    Loader.add_multi_constructor("tag:yaml.org,2002:python/object/new:", construct_new)
    Loader.add_multi_constructor("tag:yaml.org,2002:python/object:", construct_obj)
    Loader.add_constructor("tag:yaml.org,2002:python/tuple", lambda l, n: tuple(l.construct_sequence(n, deep=True)))
    return yaml, Loader


def _wrap(v):
    """Untagged mappings (a log re-dumped as plain yaml) become records too."""
    if isinstance(v, dict):
        return RosMsg(**{str(k): _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def load_ros_yaml(path_or_text, is_text=False):
    """List of RosMsg records from a python-tagged ROS yaml log (safe: builds records only)."""
    yaml, Loader = _loader()
    if is_text:
        data = yaml.load(path_or_text, Loader=Loader)
    else:
        with open(path_or_text, "r") as fh:
            data = yaml.load(fh, Loader=Loader)
    return [_wrap(e) if isinstance(e, dict) else e for e in (data or [])]


def stamp_seconds(msgs, secs0):
    """evaluate_exp.py:70,75,78 — float(secs - secs0) + float(nsecs)/10**9 per message."""
    return np.array([float(m.header.stamp.secs - secs0) + float(m.header.stamp.nsecs) / 10 ** 9 for m in msgs], np.float64)


def imu_arrays(msgs, secs0):
    """(t [n], quaternion xyzw [n,4], angular velocity [n,3]) of a list of Imu records."""
    t = stamp_seconds(msgs, secs0)
    q = np.array([[m.orientation.x, m.orientation.y, m.orientation.z, m.orientation.w] for m in msgs], np.float64).reshape(-1, 4)
    w = np.array([[m.angular_velocity.x, m.angular_velocity.y, m.angular_velocity.z] for m in msgs], np.float64).reshape(-1, 3)
    return t, q, w


def range_arrays(msgs, secs0):
    """(t [n], range [n]) of a list of Range records."""
    return stamp_seconds(msgs, secs0), np.array([m.range for m in msgs], np.float64)


def associate(ctx, t_img, imu_msgs, hgt_msgs, secs0, sensors=None):
    """Sensor rows for frames taken at `t_img` (seconds relative to secs0) from recorded IMU and range logs:
    evaluate_exp.py:77-95 for a whole batch of frames in one device call.  Returns (sensors, imu_index, hgt_index)."""
    it, iq, iw = imu_arrays(imu_msgs, secs0)
    ht, hr = range_arrays(hgt_msgs, secs0)
    return ctx.associate_sensors(t_img, it, iq, iw, ht, hr, sensors)
