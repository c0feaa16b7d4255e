"""Segment / polyline intersection with the reference's names (tools/intersection.py:4-30 upstream).
Host-only f64 geometry on a handful of points per frame: the count observable, not a kernel."""
import sys

_EPS = sys.float_info.epsilon


def _cross(ax, ay, bx, by):
    return ax * by - ay * bx


def intersection(p, pr, q, qs):
    rx, ry = float(pr[0]) - float(p[0]), float(pr[1]) - float(p[1])
    sx, sy = float(qs[0]) - float(q[0]), float(qs[1]) - float(q[1])
    mx, my = float(q[0]) - float(p[0]), float(q[1]) - float(p[1])
    rxs = _cross(rx, ry, sx, sy)
    qpxr = _cross(mx, my, rx, ry)
    if abs(rxs) < _EPS:
        if abs(qpxr) >= _EPS:
            return False                       # parallel, apart
        rr = rx * rx + ry * ry                 # collinear: overlap of the projections on r
        if rr == 0.0:
            ex, ey = float('nan'), float('nan')
        else:
            ex, ey = rx / rr, ry / rr
        t0 = mx * ex + my * ey
        t1 = t0 + sx * ex + sy * ey
        if t0 > t1:
            t0, t1 = t1, t0
        return not (t1 < 0 or t0 > 1)
    t = _cross(mx, my, sx, sy) / rxs
    u = qpxr / rxs
    return 0.0 <= t <= 1.0 and 0.0 <= u <= 1.0


def any_intersection(p1, q1, pts):
    for a, b in zip(pts, pts[1:]):
        if intersection(p1, q1, a, b):
            return True
    return False
