#!/usr/bin/env python3
"""Generates tests/golden/viewer_orbit.npz by running the REFERENCE's OrbitCamera (OrbitCamera.h, compiled unmodified into
oracle/_ref/libgsxref.so by `make -C oracle`) on synthetic window-event sequences.  Build container only (needs /root/reference).
The fixture is data: the event rows and the camera world matrix after every prefix of each sequence."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402

MOVE, BUTTON, SCROLL = 0.0, 1.0, 2.0


def sequences():
    """name -> (events4, distance, orbit_rate, scroll_rate).  Every sequence starts with a cursor event while no button is down."""
    dt = 1.0 / 60.0
    out = {}
    drag = [[MOVE, 500, 400, dt], [BUTTON, 0, 1, 0]] + [[MOVE, 500 + 7 * k, 400 + 3 * k, dt] for k in range(1, 40)] + [[BUTTON, 0, 0, 0]]
    out["drag"] = (drag + [[MOVE, 100, 100, dt], [MOVE, 900, 700, dt]], 5.0, 1.0, 1.0)            # (moves after the release do nothing)
    # straight up until the pitch would pass the pole: the reference takes those steps back (OrbitCamera.h:63-72)
    out["over_the_pole"] = ([[MOVE, 300, 600, dt], [BUTTON, 0, 1, 0]] + [[MOVE, 300, 600 - 25 * k, 0.05] for k in range(1, 60)], 5.0, 2.0, 1.0)
    out["under_the_pole"] = ([[MOVE, 300, 100, dt], [BUTTON, 0, 1, 0]] + [[MOVE, 300 + k, 100 + 25 * k, 0.05] for k in range(1, 60)], 5.0, 2.0, 1.0)
    # jumps far larger than maxOrbitRate allows per event, both signs
    out["clamped"] = ([[MOVE, 0, 0, dt], [BUTTON, 0, 1, 0], [MOVE, 4000, -3000, 0.5], [MOVE, -4000, 3000, 0.5], [MOVE, 10, 10, 0.25]], 3.0, 0.3, 5.0)
    out["right_button_is_ignored"] = ([[MOVE, 10, 10, dt], [BUTTON, 1, 1, 0], [MOVE, 400, 300, dt], [BUTTON, 0, 1, 0], [MOVE, 420, 330, dt],
                                       [BUTTON, 1, 0, 0], [MOVE, 440, 300, dt]], 5.0, 1.0, 1.0)
    out["scroll"] = ([[MOVE, 1, 1, dt]] + [[SCROLL, 1.0, 0, dt]] * 30 + [[SCROLL, -2.5, 0, 0.1]] * 7 + [[SCROLL, 3.0, 0, 0.5]] * 9, 5.0, 1.0, 20.0)   # ends at minDistance
    rng = np.random.default_rng(20240611)
    for i in range(4):
        ev = [[MOVE, float(rng.integers(0, 1000)), float(rng.integers(0, 800)), dt]]
        x, y = ev[0][1], ev[0][2]
        for _ in range(300):
            r = rng.random()
            step_dt = float(rng.choice([1 / 240, 1 / 60, 1 / 30, 0.2]))
            if r < 0.70:
                x += float(rng.normal(0, 30)); y += float(rng.normal(0, 30))
                ev.append([MOVE, x, y, step_dt])
            elif r < 0.85:
                ev.append([BUTTON, float(rng.integers(0, 2)), float(rng.integers(0, 2)), 0])
            else:
                ev.append([SCROLL, float(rng.normal(0, 2)), 0, step_dt])
        out[f"random{i}"] = (ev, float(rng.uniform(0.5, 8)), float(rng.uniform(0.01, 2.0)), float(rng.uniform(0.1, 20.0)))
    return out


def main():
    R = orc.Reference()
    data = {}
    for name, (ev, dist, orate, srate) in sequences().items():
        ev = np.asarray(ev, np.float64).reshape(-1, 4)
        data[f"{name}_events"] = ev
        data[f"{name}_params"] = np.array([dist, orate, srate], np.float32)
        data[f"{name}_world"] = np.stack([R.orbit_events(ev[:k], dist, orate, srate) for k in range(len(ev) + 1)])
    data["names"] = np.array(sorted(n[:-7] for n in data if n.endswith("_events")))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "viewer_orbit.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
