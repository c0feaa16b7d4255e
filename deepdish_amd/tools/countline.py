"""Count-line book-keeping of the reference's Pipeline.process_results / check_deleted_track
(deepdish.py:1035-1114, 1303-1312 upstream), including the quirk that `delcounts` is overwritten
(not merged) for every deleted track of a frame (deepdish.py:1040-1044)."""
import numpy as np

from .intersection import intersection, any_intersection


class CountLine:
    def __init__(self, line, wanted_labels=('person',)):
        self.line = np.asarray(line, dtype=float).reshape(2, 2)
        self.labels = list(wanted_labels)
        self.db = {}
        self.poscount = {l: 0 for l in self.labels}
        self.negcount = {l: 0 for l in self.labels}
        self.intcount = {l: 0 for l in self.labels}
        self.delcount = {l: 0 for l in self.labels}

    def check_deleted_track(self, track):
        out = {}
        tid = track.track_id
        if tid in self.db and len(self.db[tid]) > 1:
            if any_intersection(self.line[0], self.line[1], self.db[tid]):
                lbl = track.get_label()
                out[lbl] = out.get(lbl, 0) + 1
            self.db[tid] = []
        return out

    def step(self, tracker):
        delcounts = {}
        for trk in tracker.deleted_tracks:
            if trk.is_deleted():
                delcounts = self.check_deleted_track(trk)
        (px, py), (qx, qy) = self.line
        events = []
        for trk in tracker.tracks:
            if not trk.is_confirmed() or trk.time_since_update > 1:
                continue
            bb = trk.to_tlbr()
            pts = self.db.setdefault(trk.track_id, [])
            pts.append(((bb[0] + bb[2]) / 2.0, bb[3]))
            if len(pts) > 1:
                new, old = pts[-1], pts[-2]
                cp = (qx - px) * (old[1] - new[1]) - (qy - py) * (old[0] - new[0])   # deepdish.py:1075
                if intersection(self.line[0], self.line[1], new, old):
                    events.append((trk.get_label(), cp, trk.track_id))
        for lbl, cp, _ in events:
            if lbl not in self.poscount:
                continue
            if cp >= 0:
                self.poscount[lbl] += 1
            else:
                self.negcount[lbl] += 1
            self.intcount[lbl] += 1
        for lbl, d in delcounts.items():
            if lbl in self.delcount:
                self.delcount[lbl] += d
        return events

    def vector(self):
        """int64 [n_labels, 4] = (pos, neg, int, del): the payload of the multi-GPU count reduce."""
        return np.array([[self.poscount[l], self.negcount[l], self.intcount[l], self.delcount[l]]
                         for l in self.labels], dtype=np.int64)
