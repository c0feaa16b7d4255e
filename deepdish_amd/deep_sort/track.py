"""Host-side view of one track (deep_sort/track.py:5-209 upstream).

The Kalman state lives in HBM inside the C++ tracker; `mean` is mirrored to the host after every
`Tracker.update`, `covariance` is fetched on demand.  Label voting (track.py:154-188) is host-only.
"""
import numpy as np


class TrackState:
    Tentative = 1
    Confirmed = 2
    Deleted = 3


class Track:
    def __init__(self, mean, covariance, track_id, n_init, max_age, detection, _owner=None):
        self.mean = mean
        self._covariance = covariance
        self._owner = _owner
        self.track_id = track_id
        self.hits = 1
        self.age = 1
        self.time_since_update = 0
        self.state = TrackState.Tentative
        self.features = [detection.feature]
        self.labels = [detection.label]
        self.dist = {detection.label: [detection.confidence]}
        self.detections = [detection]
        self._n_init = n_init
        self._max_age = max_age
        self._synced = _owner is not None          # from here on, assignments to state / time_since_update reach the device

    @property
    def covariance(self):
        if self._covariance is None and self._owner is not None:
            self._owner._fill_covariances()
        return self._covariance

    @covariance.setter
    def covariance(self, value):
        self._covariance = value

    def to_tlwh(self):
        out = self.mean[:4].copy()
        out[2] *= out[3]
        out[:2] -= out[2:] / 2
        return out

    def to_tlbr(self):
        out = self.to_tlwh()
        out[2:] = out[:2] + out[2:]
        return out

    def _note_update(self, detection):
        """Host-side part of track.py:139-152 (the Kalman part already ran on the device)."""
        self.features.append(detection.feature)
        self.labels.append(detection.label)
        self.dist.setdefault(detection.label, []).append(detection.confidence)
        self.detections.append(detection)

    def predict(self, kf):
        """track.py:113-125 for this one track (Tracker.predict() does all of them in one launch)."""
        if self._owner is None:
            raise RuntimeError('this track is not attached to a device tracker')
        self._owner._track_predict(self)

    def update(self, kf, detection):
        """track.py:127-152 for this one track, outside Tracker.update (framerecords.py:158 calls it to extend a track
        from an annotation): Kalman update + gallery append on the device (dd_tracker_track_update), label vote here."""
        if self._owner is None:
            raise RuntimeError('this track is not attached to a device tracker')
        self._owner._track_update(self, detection)

    def mark_missed(self):
        """track.py:190-196.  A track that becomes Deleted leaves the device tracker at once (upstream it lingers in
        tracker.tracks until the end of the next update, where nothing can match it any more)."""
        if self.state == TrackState.Tentative or self.time_since_update > self._max_age:
            self.state = TrackState.Deleted
            if self._owner is not None:
                self._owner._drop([self])

    def _mirror(self, **fields):
        """The tracker copying device-side book-keeping into this view: no write-back."""
        self.__dict__.update(fields)

    def __setattr__(self, name, value):
        # the host may assign state / time_since_update (framerecords.py:160-161): keep the device tracker's copy in step
        object.__setattr__(self, name, value)
        if name in ('state', 'time_since_update') and getattr(self, '_synced', False) and self._owner is not None:
            self._owner._track_set(self)

    def get_label(self, return_confidence=False):
        if not self.labels:
            return (None, 0) if return_confidence else None
        names = list(self.dist)
        counts = np.array([len(self.dist[n]) for n in names])
        avgs = np.array([np.average(self.dist[n]) for n in names])
        post = (avgs + counts) / (counts.sum() + avgs.sum())      # Dirichlet-multinomial expectation
        ranked = sorted(zip(post, names), reverse=True)
        best = ranked[0][1]
        if len(ranked) > 1 and ranked[0][1] == 'motorbike' and ranked[1][1] == 'bicycle':
            best = 'motorbike' if ranked[0][0] > ranked[1][0] * 4 else 'bicycle'
        return (best, np.average(self.dist[best])) if return_confidence else best

    def is_tentative(self):
        return self.state == TrackState.Tentative

    def is_confirmed(self):
        return self.state == TrackState.Confirmed

    def is_deleted(self):
        return self.state == TrackState.Deleted
