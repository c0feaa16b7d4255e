"""Tracker with the reference's surface (deep_sort/tracker.py:10-138 upstream) over the C++/HIP
tracker of csrc/tracker.hip: `predict()`, `update(detections)`, `.tracks`, `.deleted_tracks`,
`.kf`, `._next_id`, `.metric`."""
import ctypes
import numpy as np

from .._lib import lib, check, P
from ..runtime import default_context, ptr
from . import kalman_filter
from .track import Track


class Tracker:
    def __init__(self, metric, max_iou_distance=0.7, max_age=30, n_init=3, context=None,
                 track_capacity=1024, gallery_capacity=256):
        self.metric = metric
        self.max_iou_distance = max_iou_distance
        self.max_age = max_age
        self.n_init = n_init
        self.ctx = context or default_context()
        self.kf = kalman_filter.KalmanFilter(self.ctx)
        self._tracks = []
        self.deleted_tracks = []
        self._by_id = {}
        budget = metric.budget if getattr(metric, 'budget', None) else 0
        h = P()
        check(lib().dd_tracker_create(self.ctx.handle, float(metric.matching_threshold), float(max_iou_distance),
                                      int(max_age), int(n_init), int(budget), int(track_capacity),
                                      int(gallery_capacity), ctypes.byref(h)), 'dd_tracker_create')
        self._h = h

    def __del__(self):
        try:
            if self._h:
                lib().dd_tracker_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def tracks(self):
        return self._tracks

    @tracks.setter
    def tracks(self, new_tracks):
        """deepdish.py:1047 assigns the list framerecords.process_tracking returns: live tracks that are no longer
        in it leave the device tracker too."""
        new_tracks = list(new_tracks)
        unknown = [getattr(t, 'track_id', None) for t in new_tracks if self._by_id.get(getattr(t, 'track_id', None)) is not t]
        if unknown:
            raise ValueError('tracker.tracks may only be reduced or reordered; unknown tracks %r' % unknown)
        keep = {id(t) for t in new_tracks}
        self._drop([t for t in self._tracks if id(t) not in keep], relist=False)
        self._tracks = new_tracks

    def _drop(self, tracks, relist=True):
        if not tracks:
            return
        ids = np.array([t.track_id for t in tracks], dtype=np.int64)
        check(lib().dd_tracker_remove(self._h, ptr(ids), len(ids)), 'dd_tracker_remove')
        for t in tracks:
            self._by_id.pop(t.track_id, None)
        if relist:
            gone = {id(t) for t in tracks}
            self._tracks = [t for t in self._tracks if id(t) not in gone]

    def _track_update(self, trk, detection):
        tlwh = np.ascontiguousarray(detection.tlwh, dtype=np.float64).reshape(4)
        feat = np.ascontiguousarray(detection.feature, dtype=np.float32).reshape(128)
        check(lib().dd_tracker_track_update(self._h, int(trk.track_id), ptr(tlwh), ptr(feat), 0), 'dd_tracker_track_update')
        ints, means = self._read(0)
        row = {int(r[0]): (r, m) for r, m in zip(ints, means)}[trk.track_id]
        trk._note_update(detection)
        trk._mirror(mean=row[1].copy(), _covariance=None, state=int(row[0][1]), time_since_update=int(row[0][2]),
                    hits=int(row[0][3]))

    def _track_predict(self, trk):
        check(lib().dd_tracker_track_predict(self._h, int(trk.track_id)), 'dd_tracker_track_predict')
        ints, means = self._read(0)
        row = {int(r[0]): (r, m) for r, m in zip(ints, means)}[trk.track_id]
        trk._mirror(mean=row[1].copy(), _covariance=None, time_since_update=int(row[0][2]), age=int(row[0][4]))

    def _track_set(self, trk):
        if trk.state in (1, 2) and self._by_id.get(trk.track_id) is trk:
            check(lib().dd_tracker_track_set(self._h, int(trk.track_id), int(trk.state), int(trk.time_since_update)),
                  'dd_tracker_track_set')

    @property
    def _next_id(self):
        v = ctypes.c_int64()
        check(lib().dd_tracker_next_id(self._h, ctypes.byref(v)), 'dd_tracker_next_id')
        return v.value

    def predict(self):
        check(lib().dd_tracker_predict(self._h), 'dd_tracker_predict')
        for t in self._tracks:                      # host mirror of track.py:124-125
            t._mirror(age=t.age + 1, time_since_update=t.time_since_update + 1, _covariance=None)

    def update(self, detections):
        n = len(detections)
        tlwh = np.ascontiguousarray([d.tlwh for d in detections], dtype=np.float64).reshape(n, 4)
        feats = np.ascontiguousarray([d.feature for d in detections], dtype=np.float32).reshape(n, 128)
        check(lib().dd_tracker_update(self._h, ptr(tlwh), ptr(feats), 0, n), 'dd_tracker_update')
        self._refresh(detections)

    def update_arrays(self, tlwh, feats_device, detections=None):
        """Hot-path variant: tlwh f64 [n,4] on the host, features f32 [n,128] already in HBM."""
        tlwh = np.ascontiguousarray(tlwh, dtype=np.float64).reshape(-1, 4)
        n = len(tlwh)
        check(lib().dd_tracker_update(self._h, ptr(tlwh), ptr(feats_device) if n else None, 1, n),
              'dd_tracker_update')
        self._refresh(detections)

    def last_cost(self):
        """(appearance cost [T, n], IoU cost [T, n]) the last update() associated with (parity tests): T = tracks before
        that update, n = its detections; gated appearance entries are 1e5, rows of unconfirmed tracks unspecified."""
        r, c = ctypes.c_int(), ctypes.c_int()
        check(lib().dd_tracker_last_cost(self._h, None, None, 0, ctypes.byref(r), ctypes.byref(c)), 'dd_tracker_last_cost')
        app = np.zeros((r.value, c.value)); iou = np.zeros((r.value, c.value))
        if app.size:
            check(lib().dd_tracker_last_cost(self._h, ptr(app), ptr(iou), app.size, ctypes.byref(r), ctypes.byref(c)),
                  'dd_tracker_last_cost')
        return app, iou

    def _read(self, which):
        n = ctypes.c_int()
        check(lib().dd_tracker_count(self._h, which, ctypes.byref(n)), 'dd_tracker_count')
        ints = np.zeros((n.value, 6), dtype=np.int64)
        means = np.zeros((n.value, 8), dtype=np.float64)
        if n.value:
            check(lib().dd_tracker_read(self._h, which, ptr(ints), ptr(means), None), 'dd_tracker_read')
        return ints, means

    def _refresh(self, detections):
        live_i, live_m = self._read(0)
        dead_i, dead_m = self._read(1)
        by_id = {}
        out = [[], []]
        for which, (ints, means) in enumerate(((live_i, live_m), (dead_i, dead_m))):
            for (tid, state, tsu, hits, age, last), m in zip(ints.tolist(), means):
                trk = self._by_id.get(tid)
                det = detections[last] if (detections is not None and last >= 0) else None
                if trk is None:
                    trk = Track(m.copy(), None, tid, self.n_init, self.max_age, det or _NO_DET, _owner=self)
                elif det is not None:
                    trk._note_update(det)
                trk._mirror(mean=m.copy(), _covariance=None, state=state, time_since_update=tsu, hits=hits, age=age)
                if which == 0:
                    by_id[tid] = trk
                out[which].append(trk)
        for trk in out[0]:                         # tracker.py:84-91: a confirmed track's feature cache moves to the metric
            if trk.state == 2:                     # (here: the device gallery) at the end of every update
                trk.__dict__['features'] = []
        self._by_id = by_id
        self._tracks, self.deleted_tracks = out

    def _fill_covariances(self):
        n = len(self._tracks)
        if n == 0:
            return
        covs = np.zeros((n, 64), dtype=np.float64)
        check(lib().dd_tracker_read(self._h, 0, None, None, ptr(covs)), 'dd_tracker_read')
        for t, c in zip(self._tracks, covs):
            t._covariance = c.reshape(8, 8).copy()


class _NoDetection:
    feature, label, confidence = None, None, 0.0


_NO_DET = _NoDetection()
