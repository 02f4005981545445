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
        self.x64 = torch.zeros(8 + 4 * 64, dtype=torch.float64)     # [8 + 4 r + c]: norm table of the fused W sub-problem
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

    # ---- AO-ADMM, Euclidean loss (nmf/ao_admm.py:46-68, 259-292) in the sharded protocol ----
    def _ao_init(self):
        if not hasattr(self, "dual_w"):
            self.dual_w = np.zeros_like(self.w)
            self.dual_h = np.zeros_like(self.h)
            self.inner = {}

    def ao_h_products(self, j):
        self._ao_init()
        if self.flag:
            return
        k, n = self.k, self.v.shape[1]
        self.x64.zero_()
        self.x64[0] = self._local_objective(0)
        x = self.x32.numpy()
        x[:] = 0
        x[:k * n] = (self.w.T @ self.v).ravel()
        x[k * n:k * n + k * k] = (self.w.T @ self.w).ravel()

    @staticmethod
    def _prox(kind, aux, dual, rho, lam):
        d = aux - dual - (lam / rho if kind == 1 else 0.0)
        return np.where(d < 0, 0, d)

    def ao_h_solve(self, prox_h, lam_h, admm_iter, min_iter, tol1, tol2, j):
        import scipy.linalg as sla
        if self.flag or self._record(min_iter, tol1, tol2, j):
            return
        k, n = self.k, self.v.shape[1]
        x = self.x32.numpy()
        b = x[:k * n].reshape(k, n).copy()
        g = x[k * n:k * n + k * k].reshape(k, k).copy()
        rho = np.trace(g) / k
        chol = sla.cholesky(g + rho * np.eye(k), lower=True)
        ran = 0
        for r in range(admm_iter):
            aux = sla.cho_solve((chol, True), b + rho * (self.h + self.dual_h))
            prev = self.h
            self.h = self._prox(prox_h, aux, self.dual_h, rho, lam_h)
            self.dual_h = self.dual_h + self.h - aux
            ran = r + 1
            if R.inner_stop(self.h, prev, aux, self.dual_h):
                break
        self.inner[(j, 0)] = ran

    def ao_w_products(self, min_iter, tol1, tol2, j):
        import scipy.linalg as sla
        if self.flag:
            return
        g = self.h @ self.h.T
        self._rho_w = np.trace(g) / self.k
        self._chol_w = sla.cholesky(g + self._rho_w * np.eye(self.k), lower=True)
        self._b_w = self.h @ self.v.T                       # k x m_local
        self._w_stop, self._w_ran = False, 0

    def ao_w_round(self, prox_w, lam_w, rnd):
        import scipy.linalg as sla
        if self.flag:
            return
        if rnd > 0 and not self._w_stop:                    # all-reduced sums of the previous round
            n0, n1, n2, n3 = (float(t) for t in self.x64[1:5])
            with np.errstate(divide="ignore", invalid="ignore"):
                r = np.sqrt(n0) / np.sqrt(n1)
                s = np.sqrt(n2) / np.sqrt(n3)
            self._w_stop = bool(r < 1e-2 and s < 1e-2)
        if self._w_stop:
            return
        wt, dt = self.w.T, self.dual_w.T
        aux = sla.cho_solve((self._chol_w, True), self._b_w + self._rho_w * (wt + dt))
        new = self._prox(prox_w, aux, dt, self._rho_w, lam_w)
        dual = dt + new - aux
        self.x64[1] = np.sum((new - aux) ** 2)
        self.x64[2] = np.sum(new ** 2)
        self.x64[3] = np.sum((new - wt) ** 2)
        self.x64[4] = np.sum(dual ** 2)
        self.w, self.dual_w = new.T.copy(), dual.T.copy()
        self._w_ran = rnd + 1

    def ao_w_close(self, admm_iter, j):
        if not self.flag:
            self.inner[(j, 1)] = self._w_ran

    # the W sub-problem with ONE exchange (nmfx_aoadmm_phase_w_fused / _repair): all rounds speculatively, the norm
    # sums of every round into x64[8 + 4 r + c]; after the all-reduce the stopping round is derived and the rows are
    # recomputed from the saved start for exactly that many rounds
    def _w_rounds(self, prox_w, lam_w, rounds, table=None):
        import scipy.linalg as sla
        wt, dt = self._w_start
        for rnd in range(rounds):
            aux = sla.cho_solve((self._chol_w, True), self._b_w + self._rho_w * (wt + dt))
            new = self._prox(prox_w, aux, dt, self._rho_w, lam_w)
            dual = dt + new - aux
            if table is not None:
                table[rnd] = [np.sum((new - aux) ** 2), np.sum(new ** 2), np.sum((new - wt) ** 2), np.sum(dual ** 2)]
            wt, dt = new, dual
        return wt, dt

    def ao_w_fused(self, prox_w, lam_w, admm_iter):
        if self.flag:
            return
        self._w_start = (self.w.T.copy(), self.dual_w.T.copy())
        table = np.zeros((admm_iter, 4))
        wt, dt = self._w_rounds(prox_w, lam_w, admm_iter, table)
        self.w, self.dual_w = wt.T.copy(), dt.T.copy()
        self.x64[8:8 + 4 * admm_iter] = torch.from_numpy(table.ravel())

    def ao_w_repair(self, prox_w, lam_w, admm_iter, j):
        if self.flag:
            return
        table = self.x64[8:8 + 4 * admm_iter].numpy().reshape(admm_iter, 4)
        rounds = admm_iter
        for rnd in range(admm_iter):
            with np.errstate(divide="ignore", invalid="ignore"):
                r = np.sqrt(table[rnd, 0]) / np.sqrt(table[rnd, 1])
                s = np.sqrt(table[rnd, 2]) / np.sqrt(table[rnd, 3])
            if r < 1e-2 and s < 1e-2:
                rounds = rnd + 1
                break
        if rounds < admm_iter:
            wt, dt = self._w_rounds(prox_w, lam_w, rounds)
            self.w, self.dual_w = wt.T.copy(), dt.T.copy()
        self.inner[(j, 1)] = rounds

    # ---- AO-ADMM, KL loss (nmf/ao_admm.py:71-101, 277-283) in the sharded protocol: one exchange per inner round ----
    def _ao_kl_init(self):
        self._ao_init()
        self.obj_kind = 1
        if not hasattr(self, "v_aux"):
            self.v_aux = np.zeros_like(self.v)
            self.dual_v = np.zeros_like(self.v)

    def ao_kl_h_products(self, j, rnd):
        self._ao_kl_init()
        if self.flag or (rnd > 0 and self._h_stop):
            return
        k, n = self.k, self.v.shape[1]
        x = self.x32.numpy()
        x[:] = 0
        x[:k * n] = (self.w.T @ (self.v_aux + self.dual_v)).ravel()
        x[k * n:k * n + k * k] = (self.w.T @ self.w).ravel()
        if rnd == 0:
            self.x64.zero_()
            self.x64[0] = self._local_objective(1)
            self._h_stop, self._h_ran = False, 0

    def ao_kl_h_round(self, prox_h, lam_h, rnd, min_iter, tol1, tol2, j):
        import scipy.linalg as sla
        if self.flag:
            return
        k, n = self.k, self.v.shape[1]
        x = self.x32.numpy()
        if rnd == 0:
            if self._record(min_iter, tol1, tol2, j):
                return
            g = x[k * n:k * n + k * k].reshape(k, k).copy()
            self._rho_h = np.trace(g) / k
            self._chol_h = sla.cholesky(g + self._rho_h * np.eye(k), lower=True)
        if self._h_stop:
            return
        b = x[:k * n].reshape(k, n)
        aux = sla.cho_solve((self._chol_h, True), b + self._rho_h * (self.h + self.dual_h))
        prev = self.h
        self.h = self._prox(prox_h, aux, self.dual_h, self._rho_h, lam_h)
        v_bar = self.w @ aux - self.dual_v
        self.v_aux = 0.5 * ((v_bar - 1) + np.sqrt((v_bar - 1) ** 2 + 4 * self.v))
        self.dual_h = self.dual_h + self.h - aux
        self.dual_v = self.dual_v + self.v_aux - self.w @ aux
        self._h_ran = rnd + 1
        self._h_stop = bool(R.inner_stop(self.h, prev, aux, self.dual_h))

    def ao_kl_h_close(self, admm_iter, min_iter, tol1, tol2, j):
        import scipy.linalg as sla
        if self.flag:
            return
        self.inner[(j, 0)] = self._h_ran
        g = self.h @ self.h.T
        self._rho_w = np.trace(g) / self.k
        self._chol_w = sla.cholesky(g + self._rho_w * np.eye(self.k), lower=True)
        self._w_stop, self._w_ran = False, 0

    def ao_kl_w_round(self, prox_w, lam_w, rnd):
        import scipy.linalg as sla
        if self.flag:
            return
        if rnd > 0 and not self._w_stop:                    # all-reduced sums of the previous round
            n0, n1, n2, n3 = (float(t) for t in self.x64[1:5])
            with np.errstate(divide="ignore", invalid="ignore"):
                r = np.sqrt(n0) / np.sqrt(n1)
                s = np.sqrt(n2) / np.sqrt(n3)
            self._w_stop = bool(r < 1e-2 and s < 1e-2)
        if self._w_stop:
            return
        wt, dt = self.w.T, self.dual_w.T
        st = (self.v_aux + self.dual_v).T                   # n x m_local
        aux = sla.cho_solve((self._chol_w, True), self.h @ st + self._rho_w * (wt + dt))
        new = self._prox(prox_w, aux, dt, self._rho_w, lam_w)
        v_bar = self.h.T @ aux - self.dual_v.T
        v_aux_t = 0.5 * ((v_bar - 1) + np.sqrt((v_bar - 1) ** 2 + 4 * self.v.T))
        dual = dt + new - aux
        self.dual_v = (self.dual_v.T + v_aux_t - self.h.T @ aux).T.copy()
        self.v_aux = v_aux_t.T.copy()
        self.x64[1] = np.sum((new - aux) ** 2)
        self.x64[2] = np.sum(new ** 2)
        self.x64[3] = np.sum((new - wt) ** 2)
        self.x64[4] = np.sum(dual ** 2)
        self.w, self.dual_w = new.T.copy(), dual.T.copy()
        self._w_ran = rnd + 1

    def ao_kl_w_close(self, admm_iter, j):
        if not self.flag:
            self.inner[(j, 1)] = self._w_ran

    # ---- ADMM (nmf/admm.py:292-334) in the sharded protocol ----
    def set_l2n_operator(self, which, p):
        if not hasattr(self, "l2n"):
            self.l2n = {}
        self.l2n[which] = np.asarray(p, dtype=np.float64)

    def admm_products(self, kind, rho, prox_w, prox_h, j):
        if not hasattr(self, "w_aux"):
            self.w_aux, self.h_aux = self.w.copy(), self.h.copy()
            self.dual_w, self.dual_h = np.zeros_like(self.w), np.zeros_like(self.h)
            self.v_aux, self.dual_v = np.zeros_like(self.v), np.zeros_like(self.v)
        self.obj_kind = kind
        if self.flag:
            return
        k, n = self.k, self.v.shape[1]
        self.x64[:8] = 0
        self.x64[0] = self._local_objective(kind)
        x = self.x32.numpy()
        x[:] = 0
        data = self.v if kind == 0 else self.v_aux + self.dual_v
        x[:k * n] = (self.w_aux.T @ data).ravel()
        x[k * n:k * n + k * k] = (self.w_aux.T @ self.w_aux).ravel()

    def _admm_prox(self, code, aux, dual, rho, lam, which):
        kind = {0: "nn", 1: "l1n", 2: "l2n", 3: "l1inf", 4: "l1inf_transpose"}[code]
        if kind == "l2n":
            out = self.l2n[which] @ (aux - dual)
            return np.where(out < 0, 0, out)
        return R.prox(kind, aux, dual, rho=rho, lam=lam)

    def admm_update(self, kind, rho, prox_w, lam_w, prox_h, lam_h, min_iter, tol1, tol2, j):
        if self.flag or self._record(min_iter, tol1, tol2, j):
            return
        k, n = self.k, self.v.shape[1]
        x = self.x32.numpy()
        b = x[:k * n].reshape(k, n)
        g = x[k * n:k * n + k * k].reshape(k, k)
        self.h_aux = np.linalg.solve(g + rho * np.eye(k), b + rho * (self.h + self.dual_h))
        data = self.v if kind == 0 else self.v_aux + self.dual_v
        a = self.h_aux @ self.h_aux.T + rho * np.eye(k)
        self.w_aux = np.linalg.solve(a, self.h_aux @ data.T + rho * (self.w.T + self.dual_w.T)).T
        self.h = self._admm_prox(prox_h, self.h_aux, self.dual_h, rho, lam_h, 1)
        self.w = self._admm_prox(prox_w, self.w_aux.T, self.dual_w.T, rho, lam_w, 0).T
        if kind == 1:
            v_bar = self.w_aux @ self.h_aux - self.dual_v
            self.v_aux = 0.5 * ((v_bar - 1) + np.sqrt((v_bar - 1) ** 2 + 4 * self.v))
            self.dual_v = self.dual_v + self.v_aux - self.w_aux @ self.h_aux
        self.dual_h = self.dual_h + self.h - self.h_aux
        self.dual_w = self.dual_w + self.w - self.w_aux

    def objective_partial(self):
        if not self.flag:
            self.x64[0] = self._local_objective(getattr(self, "obj_kind", 0))

    # ---- ANLS (nmf/anls.py:18-47, 112-126) in the sharded protocol ----
    def anls_set_distance(self, kind):
        self.obj_kind = kind

    def anls_objective(self, j):
        if not self.flag:
            self.x64.zero_()
            self.x64[0] = self._local_objective(getattr(self, "obj_kind", 0))

    def anls_w(self, lam_w, min_iter, tol1, tol2, j):
        if self.flag or self._record(min_iter, tol1, tol2, j):
            return
        k, n = self.k, self.v.shape[1]
        self.w = R.anls_w_step(self.v, self.h, lam_w)
        x = self.x32.numpy()
        x[:] = 0
        x[:k * n] = (self.w.T @ self.v).ravel()
        x[k * n:k * n + k * k] = (self.w.T @ self.w).ravel()

    def anls_h(self, lam_h, j):
        if self.flag:
            return
        # the stacked problem of anls.py:34-47 from its normal equations (what the device solves):
        # min ||[W; sqrt(2 lam) I] h - [v; 0]||  <=>  G = W^T W + 2 lam I, r = W^T v
        import scipy.optimize as so
        k, n = self.k, self.v.shape[1]
        x = self.x32.numpy()
        b = x[:k * n].reshape(k, n)
        g = x[k * n:k * n + k * k].reshape(k, k) + 2 * lam_h * np.eye(k)
        chol = np.linalg.cholesky(g)                        # g = chol chol^T: NNLS on (chol^T, chol^-1 r)
        rhs = np.linalg.solve(chol, b)
        self.h = np.stack([so.nnls(chol.T, rhs[:, c])[0] for c in range(n)], axis=1)

    def state(self):
        return self.flag, self.stop_i, len(self.obj)

    def objectives(self, first, count):
        return np.asarray(self.obj[first:first + count])

    def get_factors(self):
        return self.w, self.h


class ChunkedHostShard(HostShard):
    """HostShard with the exchange buffer of the device's split-bf16 path -- [column][factor] -- and phase A in pieces
    (phase_a_head / phase_a_cols, Euclidean loss), so that dist.run_iterations' chunked loop runs on the CPU."""
    unit = 8                                           # column granularity of the stand-in

    def chunk_ranges(self, kind, chunks):
        n, k = self.v.shape[1], self.k
        if kind != 0 or chunks < 2 or n < 2 * self.unit:
            return None
        step = max(self.unit, -(-n // chunks) // self.unit * self.unit)
        edges = list(range(0, n, step)) + [n]
        if len(edges) > 2 and edges[-1] - edges[-2] < self.unit:
            del edges[-2]
        total = self.x32.numel()
        return [(c0, c1, c0 * k, c1 * k if c1 < n else total) for c0, c1 in zip(edges[:-1], edges[1:])]

    def phase_a_head(self, kind, lambda_w, j):
        assert kind == 0
        self.cols_seen = []
        if self.flag:
            return
        self.x64.zero_()
        self.x64[0] = self._local_objective(kind)
        self.w_new = R.mur_w_step("eu", self.v, self.w, self.h, self.w @ self.h, lambda_w)
        self.x32.zero_()

    def phase_a_cols(self, kind, c0, c1):
        self.cols_seen.append((c0, c1))
        if self.flag:
            return
        k, n = self.k, self.v.shape[1]
        x = self.x32.numpy()
        x[c0 * k:c1 * k] = (self.v[:, c0:c1].T @ self.w_new).ravel()          # [column][factor]
        if c1 == n:
            x[k * n:k * n + k * k] = (self.w_new.T @ self.w_new).ravel()

    def phase_a(self, kind, lambda_w, j):
        if kind != 0:
            return super().phase_a(kind, lambda_w, j)
        self.phase_a_head(kind, lambda_w, j)
        self.phase_a_cols(kind, 0, self.v.shape[1])

    def phase_b(self, kind, lambda_h, min_iter, tol1, tol2, j):
        if kind != 0:
            return super().phase_b(kind, lambda_h, min_iter, tol1, tol2, j)
        if self.flag or self._record(min_iter, tol1, tol2, j):
            return
        k, n = self.k, self.v.shape[1]
        x = self.x32.numpy()
        b = x[:k * n].reshape(n, k).T
        g = x[k * n:k * n + k * k].reshape(k, k)
        self.w = self.w_new
        self.h = self.h * b / (g @ self.h + lambda_h * self.h + R.EPS)


class SlicedHostShard(ChunkedHostShard):
    """ChunkedHostShard with the phase B of the reduce-scatter / all-gather exchange (nmfx_mur_slice_info / _phase_b_slice /
    _phase_b_rest): this rank updates its n / world columns of H from its reduce-scattered range of the [column][factor] buffer,
    leaves them in that range, and takes the other ranks' columns from theirs after the all-gather."""

    def slice_info(self, kind, world):
        n, k = self.v.shape[1], self.k
        if kind != 0 or n % world:
            return 0, 0
        self.slices_seen = getattr(self, "slices_seen", [])
        return n // world, (n // world) * k

    def phase_b_slice(self, kind, lambda_h, min_iter, tol1, tol2, j, c0, c1):
        assert kind == 0
        self.slices_seen.append((c0, c1))
        self._stopped_now = bool(self.flag or self._record(min_iter, tol1, tol2, j))
        if self._stopped_now:
            return
        k, n = self.k, self.v.shape[1]
        x = self.x32.numpy()
        b = x[c0 * k:c1 * k].reshape(c1 - c0, k).T
        g = x[k * n:k * n + k * k].reshape(k, k)
        self.w = self.w_new
        h = self.h[:, c0:c1]
        new = h * b / (g @ h + lambda_h * h + R.EPS)
        self.h = self.h.copy()
        self.h[:, c0:c1] = new
        x[c0 * k:c1 * k] = new.T.ravel()                 # what the all-gather sends

    def phase_b_rest(self, kind, c0, c1):
        if self._stopped_now:
            return
        k, n = self.k, self.v.shape[1]
        full = self.x32.numpy()[:k * n].reshape(n, k).T
        keep = self.h[:, c0:c1].copy()
        self.h = full.copy()
        self.h[:, c0:c1] = keep
