"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the count-line logic.

Follows tools/intersection.py:4-30 and the caller-side bookkeeping in
deepdish.py:1035-1114 (process_results) and deepdish.py:1303-1312
(check_deleted_track), including the "delcounts is overwritten per deleted
track" quirk at deepdish.py:1040-1044.  Pinned by the six in-tree asserts of
tools/intersection.py:35-57 and by tests/golden/scene_*.npz.
"""
import sys
import numpy as np

_EPS = sys.float_info.epsilon


def _cross2(a, b):
    return a[0] * b[1] - a[1] * b[0]


def intersection(p, pr, q, qs):
    """tools/intersection.py:4-24 -- segment p->pr against segment q->qs."""
    p, pr, q, qs = (np.asarray(v, dtype=float) for v in (p, pr, q, qs))
    r, s = pr - p, qs - q
    rxs = _cross2(r, s)
    qmp = q - p
    qpxr = _cross2(qmp, r)
    if abs(rxs) < _EPS:
        if abs(qpxr) < _EPS:
            rdrr = r / np.dot(r, r)
            t0 = np.dot(qmp, rdrr)
            t1 = t0 + np.dot(s, rdrr)
            if t0 > t1:
                t0, t1 = t1, t0
            return bool(not (t1 < 0 or t0 > 1))
        return False
    t = _cross2(qmp, s) / rxs
    u = qpxr / rxs
    return bool(0.0 <= t <= 1.0 and 0.0 <= u <= 1.0)


def any_intersection(p1, q1, pts):
    """tools/intersection.py:26-30."""
    return any(intersection(p1, q1, a, b) for a, b in zip(pts, pts[1:]))


class CountLine:
    """The counters of deepdish.py:521-528 driven as process_results drives them."""

    def __init__(self, line, wanted_labels=('person',)):
        self.line = np.asarray(line, dtype=float).reshape(2, 2)     # deepdish.py:739-744
        self.db = {}
        self.poscount = {l: 0 for l in wanted_labels}
        self.negcount = {l: 0 for l in wanted_labels}
        self.intcount = {l: 0 for l in wanted_labels}
        self.delcount = {l: 0 for l in wanted_labels}

    def _check_deleted(self, track):
        out = {}
        i = track.track_id
        if i in self.db and len(self.db[i]) > 1:
            if any_intersection(self.line[0], self.line[1], np.array(self.db[i])):
                l = track.get_label()
                out[l] = out.get(l, 0) + 1
            self.db[i] = []
        return out

    def step(self, tracker):
        """One call per frame, after tracker.update()."""
        delcounts = {}
        for trk in tracker.deleted_tracks:
            if trk.is_deleted():
                delcounts = self._check_deleted(trk)       # overwritten, not merged
        events = []
        for trk in tracker.tracks:
            lbl = trk.get_label()
            if not trk.is_confirmed() or trk.time_since_update > 1:
                continue
            pts = self.db.setdefault(trk.track_id, [])
            bb = trk.to_tlbr()
            pts.append(np.array([(bb[0] + bb[2]) / 2.0, bb[3]]))
            if len(pts) > 1:
                p1, q1 = self.line
                p2, q2 = pts[-1], pts[-2]
                cp = _cross2(q1 - p1, q2 - p2)
                if intersection(p1, q1, p2, q2):
                    events.append((lbl, cp))
        for lbl, cp in events:
            if cp >= 0:
                self.poscount[lbl] += 1
            else:
                self.negcount[lbl] += 1
            self.intcount[lbl] += 1
        for lbl, d in delcounts.items():
            self.delcount[lbl] += d
        return events

    def vector(self, labels=None):
        """int64 [n_labels, 4] = (pos, neg, int, del) -- what the RCCL reduce carries."""
        labels = list(self.poscount) if labels is None else labels
        return np.array([[self.poscount[l], self.negcount[l], self.intcount[l], self.delcount[l]]
                         for l in labels], dtype=np.int64)
