"""Host-side quaternion helpers (numpy part of fluidlab/utils/geom.py; the in-kernel versions live in csrc/fe_math.h)."""
import numpy as np


def xyzw_to_wxyz(q):
    return np.array([q[3], q[0], q[1], q[2]])


def xyzw_from_wxyz(q):
    return np.array([q[1], q[2], q[3], q[0]])


def euler_to_quat_wxyz(euler_deg):
    """Effector init_euler -> wxyz quaternion, as effector.py:45:
    Rotation.from_euler('zyx', euler[::-1], degrees=True).as_quat() reordered to wxyz."""
    from scipy.spatial.transform import Rotation
    return xyzw_to_wxyz(Rotation.from_euler('zyx', np.asarray(euler_deg, dtype=np.float64)[::-1], degrees=True).as_quat())


def transform_by_quat_np(v, quat):
    qvec = quat[1:]
    uv = np.cross(qvec, v)
    uuv = np.cross(qvec, uv)
    return v + 2 * (quat[0] * uv + uuv)
