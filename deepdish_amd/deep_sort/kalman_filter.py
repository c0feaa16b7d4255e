"""KalmanFilter with the reference's method surface (deep_sort/kalman_filter.py:23-229), computed by
the f64 HIP kernels of csrc/kalman.hip.  Accepts one state or a batch (leading dimension)."""
import numpy as np
import torch

from .._lib import lib, check
from ..runtime import default_context, ptr

chi2inv95 = {1: 3.8415, 2: 5.9915, 3: 7.8147, 4: 9.4877, 5: 11.070,
             6: 12.592, 7: 14.067, 8: 15.507, 9: 16.919}      # kalman_filter.py:11-20


class KalmanFilter(object):
    def __init__(self, context=None):
        self.ctx = context or default_context()

    def _up(self, mean, covariance):
        m = np.asarray(mean, dtype=np.float64)
        single = m.ndim == 1
        m = m.reshape(-1, 8)
        c = np.asarray(covariance, dtype=np.float64).reshape(-1, 8, 8)
        return single, self.ctx.to_device(m), self.ctx.to_device(c)

    def initiate(self, measurement):
        z = np.asarray(measurement, dtype=np.float64)
        single = z.ndim == 1
        z = z.reshape(-1, 4)
        n = len(z)
        dz = self.ctx.to_device(z)
        m, c = self.ctx.empty((n, 8), torch.float64), self.ctx.empty((n, 8, 8), torch.float64)
        check(lib().dd_kf_initiate(self.ctx.handle, ptr(m), ptr(c), None, ptr(dz), n, None), 'dd_kf_initiate')
        m, c = self.ctx.to_host(m), self.ctx.to_host(c)
        return (m[0], c[0]) if single else (m, c)

    def predict(self, mean, covariance):
        single, m, c = self._up(mean, covariance)
        check(lib().dd_kf_predict(self.ctx.handle, ptr(m), ptr(c), None, len(m), None), 'dd_kf_predict')
        m, c = self.ctx.to_host(m), self.ctx.to_host(c)
        return (m[0], c[0]) if single else (m, c)

    def project(self, mean, covariance):
        single, m, c = self._up(mean, covariance)
        n = len(m)
        pm, pc = self.ctx.empty((n, 4), torch.float64), self.ctx.empty((n, 4, 4), torch.float64)
        check(lib().dd_kf_project(self.ctx.handle, ptr(m), ptr(c), None, n, ptr(pm), ptr(pc), None), 'dd_kf_project')
        pm, pc = self.ctx.to_host(pm), self.ctx.to_host(pc)
        return (pm[0], pc[0]) if single else (pm, pc)

    def update(self, mean, covariance, measurement):
        single, m, c = self._up(mean, covariance)
        z = self.ctx.to_device(np.asarray(measurement, dtype=np.float64).reshape(-1, 4))
        check(lib().dd_kf_update(self.ctx.handle, ptr(m), ptr(c), None, ptr(z), len(m), None), 'dd_kf_update')
        m, c = self.ctx.to_host(m), self.ctx.to_host(c)
        return (m[0], c[0]) if single else (m, c)

    def gating_distance(self, mean, covariance, measurements, only_position=False):
        single, m, c = self._up(mean, covariance)
        z = self.ctx.to_device(np.asarray(measurements, dtype=np.float64).reshape(-1, 4))
        n, nd = len(m), len(z)
        out = self.ctx.empty((n, nd), torch.float64)
        check(lib().dd_kf_gate(self.ctx.handle, ptr(m), ptr(c), None, n, ptr(z), nd, int(bool(only_position)),
                               ptr(out), None), 'dd_kf_gate')
        out = self.ctx.to_host(out)
        return out[0] if single else out
