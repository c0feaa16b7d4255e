"""Frame ingest ring (csrc/ingest.hip): the GPU side of Pipeline.capture (deepdish.py:837-878 upstream).

A decoder thread writes raw BGR frames of S streams into a pinned slot (`host(slot)` is a numpy view of
it), `submit(slot)` queues upload + cv2.flip(frame, 0) + cv2.resize(frame, input_size) on a private copy
stream, and `acquire(slot)` hands the device frames to the hot path once its stream has been told to wait
for them -- the host never blocks on a copy."""
import ctypes
import numpy as np

from ._lib import lib, check, P
from .runtime import default_context


class FrameIngest:
    def __init__(self, n_streams, src_size, dst_size=None, slots=2, flip=False, context=None):
        """src_size / dst_size: (width, height) like the reference's `input_size`; dst defaults to src."""
        self.ctx = context or default_context()
        self.S, self.slots = int(n_streams), int(slots)
        self.sw, self.sh = src_size
        self.dw, self.dh = dst_size or src_size
        h = P()
        check(lib().dd_ingest_create(self.ctx.handle, self.slots, self.S, self.sh, self.sw, self.dh, self.dw, int(bool(flip)),
                                     ctypes.byref(h)), 'dd_ingest_create')
        self._h = h
        self._host = []
        for i in range(self.slots):
            p, n = ctypes.c_void_p(), ctypes.c_int64()
            check(lib().dd_ingest_host_slot(self._h, i, ctypes.byref(p), ctypes.byref(n)), 'dd_ingest_host_slot')
            buf = (ctypes.c_uint8 * n.value).from_address(p.value)
            self._host.append(np.frombuffer(buf, dtype=np.uint8).reshape(self.S, self.sh, self.sw, 3))

    def __del__(self):
        try:
            if self._h:
                self._host = []
                lib().dd_ingest_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def host(self, slot):
        """Pinned numpy view [S, src_h, src_w, 3] of a slot: fill it, then submit(slot).  Blocks until the slot's
        previous upload has left the host buffer."""
        check(lib().dd_ingest_wait_uploaded(self._h, slot), 'dd_ingest_wait_uploaded')
        return self._host[slot]

    def submit(self, slot):
        check(lib().dd_ingest_submit(self._h, slot), 'dd_ingest_submit')

    def acquire(self, slot, stream=None):
        """-> device address of u8 [S, dst_h, dst_w, 3]; the consumer stream waits for the slot's upload."""
        p = ctypes.c_void_p()
        check(lib().dd_ingest_acquire(self._h, slot, stream, ctypes.byref(p)), 'dd_ingest_acquire')
        return p.value

    def release(self, slot, stream=None):
        check(lib().dd_ingest_release(self._h, slot, stream), 'dd_ingest_release')

    def frames(self, slot, stream=None):
        """acquire() wrapped as an object with .shape / .data_ptr() (what MultiStreamPipeline.step takes)."""
        return _DevView(self.acquire(slot, stream), (self.S, self.dh, self.dw, 3))


class _DevView:
    """Minimal stand-in for a device tensor: what runtime.ptr() and MultiStreamPipeline.step need."""

    def __init__(self, addr, shape):
        self._addr, self.shape = addr, shape

    def data_ptr(self):
        return self._addr
