"""Detection record -- the layout contract of deep_sort/detection.py:29-50 (upstream)."""
import numpy as np


class Detection(object):
    """tlwh f64[4], label, confidence float, feature f32[128]."""

    def __init__(self, tlwh, label, confidence, feature):
        self.tlwh = np.asarray(tlwh, dtype=np.float64)
        self.label = label
        self.confidence = float(confidence)
        self.feature = np.asarray(feature, dtype=np.float32)

    def to_tlbr(self):
        out = self.tlwh.copy()
        out[2:] += out[:2]
        return out

    def to_xyah(self):
        out = self.tlwh.copy()
        out[:2] += out[2:] / 2
        out[2] /= out[3]
        return out
