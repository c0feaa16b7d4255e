"""Device plumbing: one dd_ctx per (thread, device); torch supplies device memory and streams."""
import ctypes
import threading
import numpy as np
import torch

from ._lib import lib, check, P

_tls = threading.local()


class Context:
    """Owns a dd_ctx (a HIP stream + scratch) on one device."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError('deepdish_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU path')
        self.device = torch.cuda.current_device() if device is None else int(device)
        h = P()
        check(lib().dd_ctx_create(self.device, ctypes.byref(h)), 'dd_ctx_create')
        self.handle = h
        s = P()
        check(lib().dd_ctx_stream(h, ctypes.byref(s)), 'dd_ctx_stream')
        self.stream_ptr = s.value
        self.torch_stream = torch.cuda.ExternalStream(s.value, device=self.device)

    def sync(self):
        check(lib().dd_ctx_sync(self.handle), 'dd_ctx_sync')

    def close(self):
        if self.handle:
            lib().dd_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- host <-> device helpers used by the reference-shaped wrappers (not the hot path)
    def to_device(self, arr, dtype=None):
        a = np.ascontiguousarray(arr, dtype=dtype)
        t = torch.from_numpy(a).to(f'cuda:{self.device}')
        torch.cuda.current_stream(self.device).synchronize()
        return t

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=f'cuda:{self.device}')

    def to_host(self, t):
        self.sync()
        return t.cpu().numpy()


def default_context():
    ctx = getattr(_tls, 'ctx', None)
    if ctx is None or ctx.handle is None:
        ctx = Context()
        _tls.ctx = ctx
    return ctx


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array / None as c_void_p."""
    if t is None:
        return P(None)
    if isinstance(t, np.ndarray):
        return P(t.ctypes.data)
    return P(t.data_ptr())
