"""Background subtraction (csrc/mog2.hip): cv2.createBackgroundSubtractorMOG2() / backSub.apply(frame) and the
motion test on detector boxes, deepdish.py:889,920-924,957 upstream (SURVEY.md section 8 f, n2).

`createBackgroundSubtractorMOG2` has cv2's signature and returns an object with cv2's `apply`; `n_streams` > 1
turns it into S independent subtractors advanced by one launch (frames [S, H, W, 3]).  There is no CPU path."""
import ctypes
import numpy as np
import torch

from ._lib import lib, check, P
from .runtime import default_context, ptr


class BackgroundSubtractorMOG2:
    def __init__(self, history=500, varThreshold=16, detectShadows=True, n_streams=1, context=None):
        self.ctx = context or default_context()
        self._args = (int(history), float(varThreshold), bool(detectShadows))
        self.S = int(n_streams)
        self._h = None
        self._shape = None
        self.mask = None                     # device u8 [S, H, W] of the last apply

    # cv2 getters the host prints / a maintainer may query
    def getHistory(self): return self._args[0]
    def getVarThreshold(self): return self._args[1]
    def getDetectShadows(self): return self._args[2]
    def getNMixtures(self): return 5
    def getBackgroundRatio(self): return 0.9
    def getVarThresholdGen(self): return 9.0
    def getVarInit(self): return 15.0
    def getVarMin(self): return 4.0
    def getVarMax(self): return 75.0
    def getComplexityReductionThreshold(self): return 0.05
    def getShadowValue(self): return 127
    def getShadowThreshold(self): return 0.5

    def _ensure(self, h, w):
        if self._shape == (h, w):
            return
        self.close()                         # cv2 re-initialises the model when the frame size changes
        hd = P()
        check(lib().dd_mog2_create(self.ctx.handle, self.S, h, w, self._args[0], self._args[1], int(self._args[2]),
                                   ctypes.byref(hd)), 'dd_mog2_create')
        self._h, self._shape = hd, (h, w)
        self.mask = self.ctx.empty((self.S, h, w), torch.uint8)

    def close(self):
        if self._h:
            lib().dd_mog2_destroy(self._h)
            self._h = None
            self._shape = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def apply_device(self, frames_dev, learningRate=-1, masked_out=None):
        """frames_dev: device u8 [S, H, W, 3] (torch tensor or anything with .shape/.data_ptr()).  Queues the update
        on the context's stream and returns the device mask [S, H, W] (valid in stream order)."""
        s, h, w, c = frames_dev.shape
        if (s, c) != (self.S, 3):
            raise ValueError('frames must be [%d, H, W, 3] u8, got %r' % (self.S, tuple(frames_dev.shape)))
        self._ensure(h, w)
        check(lib().dd_mog2_apply(self._h, ptr(frames_dev), float(learningRate), ptr(self.mask),
                                  ptr(masked_out) if masked_out is not None else None, None), 'dd_mog2_apply')
        return self.mask

    def apply(self, image, fgmask=None, learningRate=-1):
        """cv2 signature: image u8 [H, W, 3] (or [S, H, W, 3] for a multi-stream subtractor) -> fgmask u8 ndarray."""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        single = a.ndim == 3
        if single:
            a = a[None]
        out = self.ctx.to_host(self.apply_device(self.ctx.to_device(a), learningRate))
        out = out[0] if single else out
        if fgmask is not None:
            fgmask[...] = out
            return fgmask
        return out

    def state(self, stream=0):
        """Test aid -> (weight [5, P], variance [5, P], mean [5, 3, P], nmodes [P]) of one stream, zero past nmodes."""
        h, w = self._shape
        n = h * w
        wt, var, mu = np.zeros((5, n), np.float32), np.zeros((5, n), np.float32), np.zeros((5, 3, n), np.float32)
        nm = np.zeros(n, np.uint8)
        check(lib().dd_mog2_state(self._h, int(stream), wt.ctypes.data_as(P), var.ctypes.data_as(P), mu.ctypes.data_as(P),
                                  nm.ctypes.data_as(P)), 'dd_mog2_state')
        return wt, var, mu, nm

    def box_counts(self, boxes_xywh, box_stream=None):
        """np.count_nonzero(fgMask[y:y+h, x:x+w]) of the last mask for int boxes already clipped to the frame."""
        b = np.ascontiguousarray(boxes_xywh, dtype=np.int32).reshape(-1, 4)
        z = np.zeros(len(b), np.int32) if box_stream is None else np.ascontiguousarray(box_stream, dtype=np.int32)
        out = np.zeros(len(b), np.int32)
        if len(b):
            h, w = self._shape
            check(lib().dd_mask_box_count(self.ctx.handle, ptr(self.mask), self.S, h, w, b.ctypes.data_as(P), z.ctypes.data_as(P),
                                          len(b), out.ctypes.data_as(P), None), 'dd_mask_box_count')
        return out


def createBackgroundSubtractorMOG2(history=500, varThreshold=16, detectShadows=True, n_streams=1, context=None):
    return BackgroundSubtractorMOG2(history, varThreshold, detectShadows, n_streams, context)


def motion_filter(counts, boxes_xywh, ratio):
    """deepdish.py:957: box k survives iff count_k >= ratio * w * h."""
    return [bool(c >= ratio * w * h) for c, (_, _, w, h) in zip(counts, boxes_xywh)]
