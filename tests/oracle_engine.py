"""A CPU stand-in for `pmf_hip.Context` built on the oracle -- TEST ONLY.

It exposes the half-sweep methods `pmf_hip.dist` sequences (fused, accumulate,
finalize) so the multi-process orchestration can be exercised over gloo on a
machine without a GPU.  `stats` "pointers" are NumPy views of the torch CPU
tensors that get all-reduced."""
import numpy as np

from oracle import cavi_oracle as orc

USER, ITEM = 0, 1


class GlooComm:
    """An EXTERNAL collective for `pmf_hip.dist` (the bring-your-own-communicator form): torch.distributed
    over gloo, CPU tensors.  `in_library = False` makes the iteration functions sequence
    accumulate -> all_reduce -> finalize themselves."""

    in_library = False

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_reduce(self, tensor):
        self._dist.all_reduce(tensor, op=self._dist.ReduceOp.SUM, group=self.group)
        return tensor

    def all_reduce_async(self, tensor):
        return self._dist.all_reduce(tensor, op=self._dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_reduce_host(self, values, op="sum"):
        import torch
        t = torch.tensor(np.asarray(values, dtype=np.float64).reshape(-1))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX if op == "max" else self._dist.ReduceOp.SUM, group=self.group)
        return t.numpy().reshape(np.shape(values))

    def barrier(self):
        self._dist.barrier(group=self.group)

    def attach(self, ctx):
        """What pmf_hip.dist.Comm.attach does for a real context: from now on the context's ITEM sweeps
        are distributed (here: by the OracleContext itself, over this communicator)."""
        ctx.comm_attach(self)
        return ctx


class AttachingGlooComm(GlooComm):
    """Looks to the model classes like pmf_hip.dist.Comm: contexts `attach` and then distribute their own ITEM
    sweeps (`in_library = True`), so the iteration functions issue the plain fused calls."""
    in_library = True


class CpuStats:
    def __init__(self, n_elems):
        import torch
        self.tensor = torch.zeros(int(n_elems), dtype=torch.float64)
        self.ptr = self.tensor.numpy()  # shares memory with the tensor


class OracleEngine:
    def __init__(self, n_users, n_items, K, u, i, x):
        self.n_users, self.n_items, self.K = n_users, n_items, K
        self.kpad, self.cov_stride = K, K * K  # stats carry full K x K matrices here
        self.u, self.i, self.x = np.asarray(u, np.int64), np.asarray(i, np.int64), np.asarray(x, np.float64)
        self.idx = (orc.group_positions(self.u, n_users), orc.group_positions(self.i, n_items))
        self.st = {}
        self.n_chunks = {USER: 1, ITEM: 1}
        self._cur = {USER: -1, ITEM: -1}

    # --- row chunks (same contract as pmf_ctx_set_row_chunks / select_chunk) ---
    def _rows(self, side):
        return self.n_users if side == USER else self.n_items

    def set_row_chunks(self, side, n):
        self.n_chunks[side] = max(1, min(int(n), self._rows(side)))

    def chunk_rows(self, side, c):
        r, n = self._rows(side), self.n_chunks[side]
        return r * c // n, r * (c + 1) // n

    def select_chunk(self, side, c):
        assert -1 <= c < self.n_chunks[side]
        self._cur[side] = c

    def _window(self, side):
        return (0, self._rows(side)) if self._cur[side] < 0 else self.chunk_rows(side, self._cur[side])

    def _put(self, side, stats, block):
        """write rows [lo, hi) of `block` ([rows, width]) into the flat stats buffer"""
        lo, hi = self._window(side)
        w = block.shape[1]
        stats[lo * w:hi * w] = block[lo:hi].reshape(-1)

    def _merge(self, side, key, new):
        """take rows [lo, hi) of a finalize result, keep the rest of the state"""
        lo, hi = self._window(side)
        out = self.st[key].copy()
        out[lo:hi] = new[lo:hi]
        return out

    # --- helpers ---------------------------------------------------------
    def _sides(self, side):
        ptr, pos = self.idx[side]
        other_ids = self.i if side == USER else self.u
        return ptr, pos, other_ids

    # --- Poisson / HPF ---------------------------------------------------
    def _gamma_sums(self, side):
        ptr, pos, oid = self._sides(side)
        me, ot = ("E_theta", "E_beta") if side == USER else ("E_beta", "E_theta")
        a, b = orc.gamma_half_sweep_segsum(self.st[me], self.st[ot], ptr, pos, oid, self.x, 0.0, 0.0)
        return a, b

    def _gamma_apply(self, side, a, b, shape_prior, rate_prior, hier, hyper_shape, hyper_rate_prior):
        s = "theta" if side == USER else "beta"
        pr = "E_xi" if side == USER else "E_eta"
        hr = "gamma_b_xi" if side == USER else "gamma_b_eta"
        rp = self.st[pr][:, None] if hier else rate_prior
        new_a, new_b = shape_prior + a, rp + b
        new_E = new_a / new_b
        for key, val in ((f"a_{s}", new_a), (f"b_{s}", new_b), (f"E_{s}", new_E)):
            self.st[key] = self._merge(side, key, val) if key in self.st else val
        if hier:
            new_hr = hyper_rate_prior + new_E.sum(axis=1)
            self.st[hr] = self._merge(side, hr, new_hr) if hr in self.st else new_hr
            self.st[pr] = self._merge(side, pr, hyper_shape / new_hr)

    def gamma_sweep(self, side, *prior):
        a, b = self._gamma_sums(side)
        self._gamma_apply(side, a, b, *prior)

    def gamma_accumulate(self, side, stats):
        a, b = self._gamma_sums(side)
        self._put(side, stats, np.stack([a, b], axis=1).reshape(a.shape[0], -1))

    def gamma_finalize(self, side, stats, *prior):
        rows = self.n_users if side == USER else self.n_items
        s = np.asarray(stats).reshape(rows, 2, self.K)
        self._gamma_apply(side, s[:, 0], s[:, 1], *prior)

    # --- Gaussian ----------------------------------------------------------
    def _gauss_sums(self, side):
        ptr, pos, oid = self._sides(side)
        me, ot = ("theta", "beta") if side == USER else ("beta", "theta")
        bs, bo = ("m_user_bias", "m_item_bias") if side == USER else ("m_item_bias", "m_user_bias")
        n_rows = ptr.size - 1
        o = oid[pos]
        row_of = np.repeat(np.arange(n_rows), np.diff(ptr))
        mo = self.st[f"m_{ot}"][o]
        S = orc._segment_sum((self.st[f"V_{ot}"][o] + np.einsum("nk,nl->nkl", mo, mo)).reshape(len(o), -1), ptr)
        resid = self.x[pos] - self.st[bs][row_of] - self.st[bo][o]
        w = orc._segment_sum(mo * resid[:, None], ptr)
        return S, w

    def _gauss_apply(self, side, S, w, sigma2, eta2):
        me = "theta" if side == USER else "beta"
        K = self.K
        S = S.reshape(-1, K, K)
        live = S[:, 0, 0] != 0
        lo, hi = self._window(side)
        live[:lo] = False
        live[hi:] = False
        V = np.linalg.inv(S[live] / sigma2 + np.eye(K) / eta2)
        self.st[f"V_{me}"] = self.st[f"V_{me}"].copy()
        self.st[f"m_{me}"] = self.st[f"m_{me}"].copy()
        self.st[f"V_{me}"][live] = V
        self.st[f"m_{me}"][live] = np.einsum("nkl,nl->nk", V, w[live]) / sigma2

    def gauss_factor_sweep(self, side, sigma2, eta2):
        S, w = self._gauss_sums(side)
        self._gauss_apply(side, S, w, sigma2, eta2)

    def gauss_factor_accumulate(self, side, stats):
        S, w = self._gauss_sums(side)
        self._put(side, stats, np.concatenate([S, w], axis=1))

    def gauss_factor_finalize(self, side, stats, sigma2, eta2):
        rows = self.n_users if side == USER else self.n_items
        s = np.asarray(stats).reshape(rows, self.K * self.K + self.K)
        self._gauss_apply(side, s[:, :self.K * self.K], s[:, self.K * self.K:], sigma2, eta2)

    def _bias_sums(self, side):
        ptr, pos, oid = self._sides(side)
        me, ot = ("theta", "beta") if side == USER else ("beta", "theta")
        bo = "m_item_bias" if side == USER else "m_user_bias"
        n_rows = ptr.size - 1
        o = oid[pos]
        row_of = np.repeat(np.arange(n_rows), np.diff(ptr))
        resid = self.x[pos] - self.st[bo][o] - np.einsum("nk,nk->n", self.st[f"m_{ot}"][o], self.st[f"m_{me}"][row_of])
        return orc._segment_sum(resid[:, None], ptr)[:, 0], np.diff(ptr).astype(np.float64)

    def _bias_apply(self, side, tot, cnt, sigma2, eta_bias2):
        key = "m_user_bias" if side == USER else "m_item_bias"
        out = self.st[key].copy()
        nz = cnt > 0
        lo, hi = self._window(side)
        nz[:lo] = False
        nz[hi:] = False
        var = 1.0 / (1.0 / eta_bias2 + cnt[nz] / sigma2)
        out[nz] = var / sigma2 * tot[nz]
        self.st[key] = out

    def gauss_bias_sweep(self, side, sigma2, eta_bias2):
        self._bias_apply(side, *self._bias_sums(side), sigma2, eta_bias2)

    def gauss_bias_accumulate(self, side, stats):
        tot, cnt = self._bias_sums(side)
        self._put(side, stats, np.stack([tot, cnt], axis=1))

    def gauss_bias_finalize(self, side, stats, sigma2, eta_bias2):
        rows = self.n_users if side == USER else self.n_items
        s = np.asarray(stats).reshape(rows, 2)
        self._bias_apply(side, s[:, 0], s[:, 1], sigma2, eta_bias2)

    # --- Gaussian MAP / gradient mode (no reference counterpart) -------------
    @property
    def sgd_stats_width(self):
        return self.K + 4

    def _sgd_new(self, side, lr, sigma2, eta2, eta_bias2):
        ptr, pos, oid = self._sides(side)
        me, ot = ("theta", "beta") if side == USER else ("beta", "theta")
        bs, bo = ("m_user_bias", "m_item_bias") if side == USER else ("m_item_bias", "m_user_bias")
        f_new, b_new = orc.gauss_sgd_half_sweep(self.st[f"m_{me}"], self.st[f"m_{ot}"], self.st[bs], self.st[bo],
                                                ptr, pos, oid, self.x, lr, sigma2, eta2, eta_bias2)
        return f"m_{me}", bs, f_new, b_new, np.diff(ptr).astype(np.float64)

    def gauss_sgd_sweep(self, side, lr, sigma2, eta2, eta_bias2):
        kf, kb, f_new, b_new, _ = self._sgd_new(side, lr, sigma2, eta2, eta_bias2)
        self.st[kf], self.st[kb] = f_new, b_new

    def gauss_sgd_accumulate(self, side, stats, lr, sigma2, eta2, eta_bias2):
        kf, kb, f_new, b_new, n = self._sgd_new(side, lr, sigma2, eta2, eta_bias2)
        block = np.zeros((n.size, self.K + 4))
        block[:, :self.K] = n[:, None] * (f_new - self.st[kf])
        block[:, self.K] = n * (b_new - self.st[kb])
        block[:, self.K + 1] = n
        self._put(side, stats, block)

    def gauss_sgd_finalize(self, side, stats):
        kf, kb = ("m_theta", "m_user_bias") if side == USER else ("m_beta", "m_item_bias")
        s = np.asarray(stats).reshape(self._rows(side), self.K + 4)
        lo, hi = self._window(side)
        live = s[:, self.K + 1] > 0
        live[:lo] = False
        live[hi:] = False
        f, b = self.st[kf].copy(), self.st[kb].copy()
        f[live] += s[live, :self.K] / s[live, self.K + 1][:, None]
        b[live] += s[live, self.K] / s[live, self.K + 1]
        self.st[kf], self.st[kb] = f, b


class OracleContext(OracleEngine):
    """A CPU stand-in for `pmf_hip.Context` itself -- TEST ONLY -- so that the model classes' sharded
    `fit` (sharding, attach, monitor all-reduce, gather, the full-size context at the end) can run over
    gloo without a GPU.  With a communicator attached (`GlooComm.attach`) its ITEM sweeps run the product's
    host-sequenced accumulate -> all-reduce -> finalize (`pmf_hip.dist._item_half_sweep`), which is what
    libpmf_hip.so does internally on a GPU."""

    NAMES = {"gamma": {(USER, 0): "E_theta", (ITEM, 0): "E_beta", (USER, 1): "a_theta", (ITEM, 1): "a_beta",
                       (USER, 2): "b_theta", (ITEM, 2): "b_beta", (USER, 3): "E_xi", (ITEM, 3): "E_eta",
                       (USER, 4): "gamma_b_xi", (ITEM, 4): "gamma_b_eta"},
             "gauss": {(USER, 0): "m_theta", (ITEM, 0): "m_beta", (USER, 5): "V_theta", (ITEM, 5): "V_beta",
                       (USER, 6): "m_user_bias", (ITEM, 6): "m_item_bias"}}

    def __init__(self, n_users, n_items, n_factors, dtype="f64", device=0):
        self.n_users, self.n_items, self.K = int(n_users), int(n_items), int(n_factors)
        self.kpad, self.cov_stride, self.np_dtype = self.K, self.K * self.K, np.float64
        self.raw, self.st = {}, {}
        self.n_chunks = {USER: 1, ITEM: 1}
        self._cur = {USER: -1, ITEM: -1}
        self._comm = None
        self._eval = None
        self.nnz = 0

    # ---- data ------------------------------------------------------------
    def set_ratings(self, u, i, x):
        self.u, self.i, self.x = np.asarray(u, np.int64), np.asarray(i, np.int64), np.asarray(x, np.float64)
        assert self.u.min(initial=0) >= 0 and self.u.max(initial=0) < self.n_users
        self.idx = (orc.group_positions(self.u, self.n_users), orc.group_positions(self.i, self.n_items))
        self.nnz = len(self.u)

    def set_array(self, side, array, host):
        self.raw[(side, array)] = np.array(host, dtype=np.float64)

    def get_array(self, side, array):
        return self.raw[(side, array)].copy()

    def set_cov_identity(self, side, scale=1.0):
        rows = self.n_users if side == USER else self.n_items
        self.raw[(side, 5)] = np.tile(scale * np.eye(self.K), (rows, 1, 1))

    def close(self):
        pass

    def comm_attach(self, comm):
        self._comm = comm

    def gather_user_rows(self, array, bounds):
        import torch.distributed as dist
        parts = [None] * self._comm.world
        dist.all_gather_object(parts, self.raw[(USER, array)], group=self._comm.group)
        assert [len(p) for p in parts] == list(np.diff(bounds))
        return np.concatenate(parts, axis=0)

    # ---- sweeps ------------------------------------------------------------
    def _with(self, kind, fn):
        names = self.NAMES[kind]
        self.st = {name: self.raw[key] for key, name in names.items() if key in self.raw}
        if kind == "gauss":
            for key, rows in (("m_user_bias", self.n_users), ("m_item_bias", self.n_items)):
                self.st.setdefault(key, np.zeros(rows))
        fn()
        for key, name in names.items():
            if name in self.st and (key in self.raw or name not in ("m_user_bias", "m_item_bias")):
                self.raw[key] = self.st[name]

    def _item_dist(self, side, width, accumulate, finalize):
        from pmf_hip import dist as pdist
        rows = self.n_items
        stats = CpuStats(rows * width)
        pdist._item_half_sweep(self, self._comm, stats, width, lambda: accumulate(stats.ptr), lambda: finalize(stats.ptr))

    def gamma_sweep(self, side, *prior):
        def run():
            if side == ITEM and self._comm is not None:
                self._item_dist(side, 2 * self.K, lambda s: OracleEngine.gamma_accumulate(self, side, s),
                                lambda s: OracleEngine.gamma_finalize(self, side, s, *prior))
            else:
                OracleEngine.gamma_sweep(self, side, *prior)
        self._with("gamma", run)

    def gamma_ext_sweep(self, side, a0, b0):
        """pmf_gamma_ext_sweep (single context only, as in the library): arrays 0 / 1 / 2 = E / shape / rate of
        the factors, 7 / 8 / 9 = E / shape / rate of the scalar scale."""
        assert self._comm is None
        other = ITEM if side == USER else USER
        ptr, pos = self.idx[0] if side == USER else self.idx[1]
        oid = self.i if side == USER else self.u
        a, b, E, sa, sb, S = orc.gamma_ext_half_sweep_rows(self.raw[(side, 0)], self.raw[(side, 7)], self.raw[(other, 0)],
                                                           self.raw[(other, 7)], ptr, pos, oid, self.x, a0, b0)
        for array, value in ((1, a), (2, b), (0, E), (8, sa), (9, sb), (7, S)):
            self.raw[(side, array)] = value

    def gauss_factor_sweep(self, side, sigma2, eta2):
        def run():
            if side == ITEM and self._comm is not None:
                self._item_dist(side, self.K * self.K + self.K, lambda s: OracleEngine.gauss_factor_accumulate(self, side, s),
                                lambda s: OracleEngine.gauss_factor_finalize(self, side, s, sigma2, eta2))
            else:
                OracleEngine.gauss_factor_sweep(self, side, sigma2, eta2)
        self._with("gauss", run)

    def gauss_bias_sweep(self, side, sigma2, eta_bias2):
        def run():
            if side == ITEM and self._comm is not None:
                saved, self.n_chunks[ITEM] = self.n_chunks[ITEM], 1      # [I x 2]: one message
                try:
                    self._item_dist(side, 2, lambda s: OracleEngine.gauss_bias_accumulate(self, side, s),
                                    lambda s: OracleEngine.gauss_bias_finalize(self, side, s, sigma2, eta_bias2))
                finally:
                    self.n_chunks[ITEM] = saved
            else:
                OracleEngine.gauss_bias_sweep(self, side, sigma2, eta_bias2)
        self._with("gauss", run)

    # ---- predict / monitor (engine.Context semantics) -------------------------
    def predict(self, user_ids, item_ids, use_bias=False, offset=0.0):
        u, i = np.asarray(user_ids, np.int64), np.asarray(item_ids, np.int64)
        ok = (u >= 0) & (u < self.n_users) & (i >= 0) & (i < self.n_items)
        out = np.zeros(len(u))
        A, B = self.raw[(USER, 0)], self.raw[(ITEM, 0)]
        out[ok] = np.einsum("nk,nk->n", A[u[ok]], B[i[ok]])
        if use_bias == 2:      # PREDICT_SCALE: the extended Poisson model's phi_u psi_i theta_u.beta_i
            out[ok] *= self.raw[(USER, 7)][u[ok]] * self.raw[(ITEM, 7)][i[ok]]
        elif use_bias:
            out[ok] += self.raw[(USER, 6)][u[ok]] + self.raw[(ITEM, 6)][i[ok]]
        return out + offset

    def eval_set(self, user_ids, item_ids, y_true, labels=None):
        y = np.asarray(y_true, np.float64)
        labels = np.unique(y) if labels is None else np.asarray(labels, np.float64)
        if len(labels) > 32 or len(y) == 0:
            return False
        self._eval = (np.asarray(user_ids, np.int64), np.asarray(item_ids, np.int64), y, np.searchsorted(labels, y))
        return True

    def eval_sums(self, use_bias=False, offset=0.0):
        u, i, y, lab = self._eval
        err = y - self.predict(u, i, use_bias, offset)
        abs_l, cnt_l = np.zeros(32), np.zeros(32)
        np.add.at(abs_l, lab, np.abs(err))
        np.add.at(cnt_l, lab, 1.0)
        return np.concatenate([[float(len(y)), float(np.sum(err * err))], abs_l, cnt_l])

    @staticmethod
    def metrics_from_sums(sums):
        n, sse = sums[0], sums[1]
        abs_l, cnt_l = sums[2:34], sums[34:66]
        seen = cnt_l > 0
        return float(np.sqrt(sse / n)), float(np.mean(abs_l[seen] / cnt_l[seen]))

    def eval_run(self, use_bias=False, offset=0.0):
        return self.metrics_from_sums(self.eval_sums(use_bias, offset))
