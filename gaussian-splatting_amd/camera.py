"""Camera poses through libgsx's host entry points (reference Camera.h, Node.cpp,
OrbitCamera.h semantics; see include/gsx.h)."""
import ctypes as C
import sys

import numpy as np


def _lib():
    pkg = sys.modules["gaussian_splatting_amd"]
    if not hasattr(_lib, "lib"):
        _lib.lib = pkg.load_library()
    return _lib.lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def orbit_pose(distance, rotations):
    """rotations: sequence of (axis3, angle) applied in order with Node::rotate."""
    axes = np.ascontiguousarray([r[0] for r in rotations], np.float32).reshape(-1, 3)
    angles = np.ascontiguousarray([r[1] for r in rotations], np.float32)
    out = np.empty(16, np.float32)
    rc = _lib().gsx_host_orbit_pose(C.c_float(distance), _p(axes) if len(angles) else None, _p(angles) if len(angles) else None,
                                    C.c_uint32(len(angles)), _p(out))
    assert rc == 0
    return out


def orbit_events(events4, distance=5.0, orbit_rate=1.0, scroll_rate=1.0):
    """World matrix of the orbit camera after window events (OrbitCamera.h:32-74 as main.cpp:139-155 forwards them):
    rows {0, xpos, ypos, dt} cursor moved, {1, button, action, -} mouse button, {2, yoffset, -, dt} scrolled."""
    ev = np.ascontiguousarray(events4, np.float64).reshape(-1, 4)
    out = np.empty(16, np.float32)
    rc = _lib().gsx_host_orbit_events(C.c_float(distance), C.c_float(orbit_rate), C.c_float(scroll_rate), _p(ev) if len(ev) else None,
                                      C.c_uint32(len(ev)), _p(out))
    assert rc == 0
    return out


def orbit_world(distance, azimuth, elevation):
    """Camera at `distance`, orbit node rotated about +Y by azimuth then about its local X by elevation."""
    rots = []
    if azimuth != 0.0:
        rots.append(((0.0, 1.0, 0.0), azimuth))
    if elevation != 0.0:
        rots.append(((float(np.cos(azimuth)), 0.0, float(-np.sin(azimuth))), elevation))
    return orbit_pose(distance, rots)


def camera_matrices(world, fov=45.0, aspect=1.0, near=0.1, far=100.0, want_viewproj=False):
    world = np.ascontiguousarray(world, np.float32)
    view, proj, vp = (np.empty(16, np.float32) for _ in range(3))
    rc = _lib().gsx_host_camera(_p(world), C.c_float(fov), C.c_float(aspect), C.c_float(near), C.c_float(far), _p(view),
                                _p(proj), _p(vp))
    assert rc == 0
    pos = world[12:15].copy()
    return (view, proj, pos, vp) if want_viewproj else (view, proj, pos)
