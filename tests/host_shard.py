"""numpy stand-in for the HIP engine of one row shard (TEST INFRASTRUCTURE).

Implements the same phase_a / all-reduce / phase_b protocol as
nmf_amd.dist.DeviceShard with the oracle's arithmetic on the local rows, so the
sharding logic of nmf_amd/dist.py can be exercised with gloo on a CPU-only box."""
import numpy as np
import torch

from oracle import nmf_ref as R


class HostShard:
    def __init__(self, v_local, k, w0_local, h0):
        self.v = np.asarray(v_local, dtype=np.float64)
        self.w = w0_local.copy()
        self.h = h0.copy()
        self.k = k
        n = self.v.shape[1]
        self.x32 = torch.zeros(k * n + k * k + k, dtype=torch.float64)   # f64 here: compare tightly
        self.x64 = torch.zeros(4, dtype=torch.float64)
        self.obj = []
        self.flag, self.stop_i = 0, -1
        self.w_new = None

    def buffers(self):
        return self.x32, self.x64

    def _local_objective(self, kind):
        wh = self.w @ self.h
        if kind == 0:
            return 0.5 * np.sum((self.v - wh) ** 2)
        with np.errstate(all="ignore"):
            t = self.v * np.log(self.v / wh)
        t = np.where(np.isnan(t) | (t == np.inf), 0, t)
        return np.sum(t - self.v + wh)

    def phase_a(self, kind, lambda_w, j):
        if self.flag:
            return
        k, n = self.k, self.v.shape[1]
        self.x64.zero_()
        self.x64[0] = self._local_objective(kind)
        wh = self.w @ self.h
        self.w_new = R.mur_w_step("eu" if kind == 0 else "kl", self.v, self.w, self.h, wh, lambda_w)
        x = self.x32.numpy()
        x[:] = 0
        if kind == 0:
            x[:k * n] = (self.w_new.T @ self.v).ravel()
            x[k * n:k * n + k * k] = (self.w_new.T @ self.w_new).ravel()
        else:
            x[:k * n] = (self.w_new.T @ (self.v / (self.w_new @ self.h + R.EPS))).ravel()
            x[k * n:k * n + k] = self.w_new.sum(axis=0)

    def _record(self, min_iter, tol1, tol2, j):
        obj = float(self.x64[0])
        rule = 0
        if j >= 1 and (j - 1) > min_iter:
            rule = R.stop_rule(obj, self.obj[j - 1], tol1, tol2)
        self.obj.append(obj)
        if rule:
            self.flag, self.stop_i = rule, j - 1
        return rule

    def phase_b(self, kind, lambda_h, min_iter, tol1, tol2, j):
        if self.flag or self._record(min_iter, tol1, tol2, j):
            return
        k, n = self.k, self.v.shape[1]
        x = self.x32.numpy()
        b = x[:k * n].reshape(k, n)
        self.w = self.w_new
        if kind == 0:
            g = x[k * n:k * n + k * k].reshape(k, k)
            self.h = self.h * b / (g @ self.h + lambda_h * self.h + R.EPS)
        else:
            d = x[k * n:k * n + k].reshape(k, 1) + 0 * self.h
            c = self.h * b
            self.h = 2 * c / (d + np.sqrt(d ** 2 + 4 * lambda_h * c))

    def finish_a(self, kind, j):
        if not self.flag:
            self.x64.zero_()
            self.x64[0] = self._local_objective(kind)

    def finish_b(self, min_iter, tol1, tol2, j):
        if not self.flag:
            self._record(min_iter, tol1, tol2, j)

    def state(self):
        return self.flag, self.stop_i, len(self.obj)

    def objectives(self, first, count):
        return np.asarray(self.obj[first:first + count])

    def get_factors(self):
        return self.w, self.h
