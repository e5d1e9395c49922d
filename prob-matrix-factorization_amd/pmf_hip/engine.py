"""`Context`: a thin Python owner of one `pmf_ctx` (one model instance on one GPU).

Every method is a direct call through the C-ABI of include/pmf_hip.h; the
class adds argument conversion (NumPy -> C pointers), error translation and
lifetime management only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import (ARR_COV, EXCHANGE, F32, F64, ITEM, KERNEL_NAMES, MAX_LABELS, OP_MAX, OP_SUM, TEST_LIB_PATH, UNIQUE_ID_BYTES,
               USER, PmfError, PmfLibraryError, as_f64, as_i32, check, load, ptr)


class Context:
    def __init__(self, n_users, n_items, n_factors, dtype="f32", device=0):
        self._lib = load()
        self._h = C.c_void_p()
        self.n_users, self.n_items, self.K = int(n_users), int(n_items), int(n_factors)
        self.dtype = {"f32": F32, "f64": F64, F32: F32, F64: F64}[dtype]
        self.np_dtype = np.float64 if self.dtype == F64 else np.float32
        check(self._lib.pmf_ctx_create(int(device), self.n_users, self.n_items, self.K, self.dtype,
                                       C.byref(self._h)), "pmf_ctx_create")
        k = C.c_int(0)
        check(self._lib.pmf_ctx_kpad(self._h, C.byref(k)), "pmf_ctx_kpad")
        self.kpad = k.value
        check(self._lib.pmf_ctx_cov_stride(self._h, C.byref(k)), "pmf_ctx_cov_stride")
        self.cov_stride = k.value
        self.nnz = 0
        self.n_chunks = {USER: 1, ITEM: 1}

    # ---- row chunks (multi-GPU pipelining of a half-sweep) --------------
    def set_row_chunks(self, side, n_chunks):
        """Split `side`'s rows into equal ranges; accumulate / finalize calls then act on the
        range chosen with `select_chunk` (-1 = all rows)."""
        n = max(1, min(int(n_chunks), self.rows(side), 1024))
        check(self._lib.pmf_ctx_set_row_chunks(self._h, side, n), "pmf_ctx_set_row_chunks")
        self.n_chunks[side] = n

    def chunk_rows(self, side, chunk):
        lo, hi = C.c_int64(0), C.c_int64(0)
        check(self._lib.pmf_ctx_chunk_rows(self._h, side, int(chunk), C.byref(lo), C.byref(hi)),
              "pmf_ctx_chunk_rows")
        return lo.value, hi.value

    def select_chunk(self, side, chunk):
        check(self._lib.pmf_ctx_select_chunk(self._h, side, int(chunk)), "pmf_ctx_select_chunk")

    # ---- multi-GPU: the communicator lives inside the library -------------
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId (rank 0 creates them, every rank passes them to comm_init)."""
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        check(load().pmf_comm_unique_id(buf), "pmf_comm_unique_id")
        return buf.raw

    def comm_init(self, nranks, rank, unique_id, transport="rccl"):
        """Collective.  transport 'rccl' (one rank per GPU), or 'hostshm' -- the rehearsal transport of the TEST build
        of the library (ranks share one GPU), loaded only when the process asked for it (`pmf_hip.wants_test_library`)."""
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError(f"unique_id must be {UNIQUE_ID_BYTES} bytes")
        if transport == "hostshm":
            if not hasattr(self._lib, "pmf_comm_init_hostshm") or self._lib.pmf_path != TEST_LIB_PATH:
                raise PmfLibraryError("the hostshm transport exists in the test build of the library only: set "
                                      "PMF_COMM_TRANSPORT=hostshm (or PMF_HIP_TEST_LIBRARY=1) before the first engine call")
            fn = self._lib.pmf_comm_init_hostshm
        else:
            fn = {"rccl": self._lib.pmf_comm_init}[transport]
        check(fn(self._h, int(nranks), int(rank), C.c_char_p(bytes(unique_id))), f"pmf_comm_init[{transport}]")

    def comm_set_exchange(self, mode):
        """'auto' (default), 'allreduce' or 'scatter_gather': how an ITEM half-sweep's statistics travel (the same
        on every rank) -- all-reduce + every rank finalises every item, or reduce-scatter -> finalise 1/N of the
        items -> all-gather of the finalised rows."""
        check(self._lib.pmf_comm_set_exchange(self._h, EXCHANGE[mode]), "pmf_comm_set_exchange")

    def comm_attach(self, owner):
        check(self._lib.pmf_comm_attach(self._h, owner._h), "pmf_comm_attach")

    def comm_destroy(self):
        check(self._lib.pmf_comm_destroy(self._h), "pmf_comm_destroy")

    def comm_info(self):
        """(nranks, rank, transport id or -1 without a communicator)"""
        n, r, t = C.c_int(1), C.c_int(0), C.c_int(-1)
        check(self._lib.pmf_comm_info(self._h, C.byref(n), C.byref(r), C.byref(t)), "pmf_comm_info")
        return n.value, r.value, t.value

    def comm_barrier(self):
        check(self._lib.pmf_comm_barrier(self._h), "pmf_comm_barrier")

    def comm_allreduce_host(self, values, op="sum"):
        a = np.array(values, dtype=np.float64, copy=True).reshape(-1)
        check(self._lib.pmf_comm_allreduce_host(self._h, ptr(a, C.c_double), a.size, OP_MAX if op == "max" else OP_SUM),
              "pmf_comm_allreduce_host")
        return a.reshape(np.shape(values))

    def gather_user_rows(self, array, bounds):
        """Every rank's USER rows of `array`, concatenated in rank order (host float64, on every rank)."""
        b = np.ascontiguousarray(np.asarray(bounds, dtype=np.int64))
        total = int(b[-1])
        shape = {2: (total, self.K), 3: (total, self.K, self.K)}.get(len(self._host_shape(USER, array)), (total,))
        out = np.empty(shape, dtype=np.float64)
        check(self._lib.pmf_comm_gather_user_rows(self._h, int(array), ptr(b, C.c_int64), ptr(out, C.c_double)),
              "pmf_comm_gather_user_rows")
        return out

    # ---- lifetime -------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pmf_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def rows(self, side):
        return self.n_users if side == USER else self.n_items

    def sync(self):
        check(self._lib.pmf_ctx_sync(self._h), "pmf_ctx_sync")

    def set_stream(self, hip_stream):
        check(self._lib.pmf_ctx_set_stream(self._h, C.c_void_p(hip_stream or 0)), "pmf_ctx_set_stream")

    def device_bytes(self):
        n = C.c_int64(0)
        check(self._lib.pmf_ctx_device_bytes(self._h, C.byref(n)), "pmf_ctx_device_bytes")
        return n.value

    # ---- data -----------------------------------------------------------
    def set_ratings(self, user_ids, item_ids, ratings):
        u, i, x = as_i32(user_ids, "user_ids"), as_i32(item_ids, "item_ids"), as_f64(ratings)
        if not (len(u) == len(i) == len(x)):
            raise ValueError("user_ids, item_ids and ratings must have the same length")
        check(self._lib.pmf_ctx_set_ratings(self._h, len(u), ptr(u, C.c_int32), ptr(i, C.c_int32),
                                            ptr(x, C.c_double)), "pmf_ctx_set_ratings")
        self.nnz = len(u)

    def _host_shape(self, side, array):
        r = self.rows(side)
        if array in (0, 1, 2):
            return (r, self.K)
        if array == ARR_COV:
            return (r, self.K, self.K)
        return (r,)

    def set_array(self, side, array, host):
        a = as_f64(host)
        if a.shape != self._host_shape(side, array):
            raise ValueError(f"array {array} of side {side}: expected shape "
                             f"{self._host_shape(side, array)}, got {a.shape}")
        check(self._lib.pmf_set_array(self._h, side, array, ptr(a, C.c_double)), "pmf_set_array")

    def get_array(self, side, array):
        out = np.empty(self._host_shape(side, array), dtype=np.float64)
        check(self._lib.pmf_get_array(self._h, side, array, ptr(out, C.c_double)), "pmf_get_array")
        return out

    def get_array_rows(self, side, array, rows):
        """The given rows of `array` as host float64 (len(rows) x K, len(rows) or len(rows) x K x K):
        `V_theta[rows]` without the 32 GB stack."""
        r = np.ascontiguousarray(np.asarray(rows, dtype=np.int64).reshape(-1))
        out = np.empty((len(r),) + self._host_shape(side, array)[1:], dtype=np.float64)
        check(self._lib.pmf_get_array_rows(self._h, side, array, len(r), ptr(r, C.c_int64), ptr(out, C.c_double)),
              "pmf_get_array_rows")
        return out

    def set_array_rows(self, side, array, rows, host):
        r = np.ascontiguousarray(np.asarray(rows, dtype=np.int64).reshape(-1))
        a = as_f64(host)
        want = (len(r),) + self._host_shape(side, array)[1:]
        if a.shape != want:
            raise ValueError(f"array {array} of side {side}: expected shape {want}, got {a.shape}")
        check(self._lib.pmf_set_array_rows(self._h, side, array, len(r), ptr(r, C.c_int64), ptr(a, C.c_double)),
              "pmf_set_array_rows")

    def set_cov_identity(self, side, scale=1.0):
        check(self._lib.pmf_set_cov_identity(self._h, side, float(scale)), "pmf_set_cov_identity")

    # ---- Poisson / HPF --------------------------------------------------
    def gamma_sweep(self, side, shape_prior, rate_prior, hierarchical=False, hyper_shape=0.0,
                    hyper_rate_prior=0.0):
        check(self._lib.pmf_gamma_sweep(self._h, side, float(shape_prior), float(rate_prior),
                                        int(bool(hierarchical)), float(hyper_shape),
                                        float(hyper_rate_prior)), "pmf_gamma_sweep")

    def gamma_ext_sweep(self, side, shape_prior, rate_prior):
        check(self._lib.pmf_gamma_ext_sweep(self._h, side, float(shape_prior), float(rate_prior)),
              "pmf_gamma_ext_sweep")

    def gamma_accumulate(self, side, stats_ptr):
        check(self._lib.pmf_gamma_accumulate(self._h, side, C.c_void_p(stats_ptr)), "pmf_gamma_accumulate")

    def gamma_finalize(self, side, stats_ptr, shape_prior, rate_prior, hierarchical=False,
                       hyper_shape=0.0, hyper_rate_prior=0.0):
        check(self._lib.pmf_gamma_finalize(self._h, side, C.c_void_p(stats_ptr), float(shape_prior),
                                           float(rate_prior), int(bool(hierarchical)),
                                           float(hyper_shape), float(hyper_rate_prior)),
              "pmf_gamma_finalize")

    # ---- Gaussian -------------------------------------------------------
    def gauss_factor_sweep(self, side, sigma2, eta2):
        check(self._lib.pmf_gauss_factor_sweep(self._h, side, float(sigma2), float(eta2)),
              "pmf_gauss_factor_sweep")

    def gauss_bias_sweep(self, side, sigma2, eta_bias2):
        check(self._lib.pmf_gauss_bias_sweep(self._h, side, float(sigma2), float(eta_bias2)),
              "pmf_gauss_bias_sweep")

    def gauss_factor_accumulate(self, side, stats_ptr):
        check(self._lib.pmf_gauss_factor_accumulate(self._h, side, C.c_void_p(stats_ptr)),
              "pmf_gauss_factor_accumulate")

    def gauss_factor_finalize(self, side, stats_ptr, sigma2, eta2):
        check(self._lib.pmf_gauss_factor_finalize(self._h, side, C.c_void_p(stats_ptr), float(sigma2),
                                                  float(eta2)), "pmf_gauss_factor_finalize")

    def gauss_bias_accumulate(self, side, stats_ptr):
        check(self._lib.pmf_gauss_bias_accumulate(self._h, side, C.c_void_p(stats_ptr)),
              "pmf_gauss_bias_accumulate")

    def gauss_bias_finalize(self, side, stats_ptr, sigma2, eta_bias2):
        check(self._lib.pmf_gauss_bias_finalize(self._h, side, C.c_void_p(stats_ptr), float(sigma2),
                                                float(eta_bias2)), "pmf_gauss_bias_finalize")

    # ---- Gaussian MAP by gradient steps (no reference counterpart) -------
    def gauss_sgd_sweep(self, side, lr, sigma2, eta2, eta_bias2=1.0):
        check(self._lib.pmf_gauss_sgd_sweep(self._h, side, float(lr), float(sigma2), float(eta2), float(eta_bias2)),
              "pmf_gauss_sgd_sweep")

    @property
    def sgd_stats_width(self):
        w = C.c_int(0)
        check(self._lib.pmf_ctx_sgd_stats_width(self._h, C.byref(w)), "pmf_ctx_sgd_stats_width")
        return w.value

    def gauss_sgd_accumulate(self, side, stats_ptr, lr, sigma2, eta2, eta_bias2=1.0):
        check(self._lib.pmf_gauss_sgd_accumulate(self._h, side, C.c_void_p(stats_ptr), float(lr), float(sigma2),
                                                 float(eta2), float(eta_bias2)), "pmf_gauss_sgd_accumulate")

    def gauss_sgd_finalize(self, side, stats_ptr):
        check(self._lib.pmf_gauss_sgd_finalize(self._h, side, C.c_void_p(stats_ptr)), "pmf_gauss_sgd_finalize")

    # ---- predict / evaluate --------------------------------------------
    @staticmethod
    def _clip_ids(ids):
        """Ids beyond int32 cannot be valid rows; map them to an id that is
        out of range for every table so they predict 0 like the reference."""
        a = np.asarray(ids, dtype=np.int64)
        big = np.iinfo(np.int32).max
        return np.ascontiguousarray(np.where((a > big) | (a < 0), big, a).astype(np.int32))

    def predict(self, user_ids, item_ids, use_bias=False, offset=0.0):
        u, i = self._clip_ids(user_ids), self._clip_ids(item_ids)
        if len(u) != len(i):
            raise ValueError("user_ids and item_ids must have the same length")
        out = np.zeros(len(u), dtype=np.float64)
        if len(u):
            check(self._lib.pmf_predict(self._h, len(u), ptr(u, C.c_int32), ptr(i, C.c_int32),
                                        int(use_bias), float(offset), ptr(out, C.c_double)),
                  "pmf_predict")
        return out

    def eval_set(self, user_ids, item_ids, y_true, labels=None):
        """Store a validation set on the device.  Returns False (and stores
        nothing) when y_true has more than MAX_LABELS distinct values -- the
        caller then evaluates through `predict`.  `labels` (sorted distinct
        values) fixes the label set, e.g. to the global one of a sharded run."""
        y = as_f64(y_true)
        if labels is None:
            labels, inverse = np.unique(y, return_inverse=True)
        else:
            labels = np.asarray(labels, dtype=np.float64)
            inverse = np.searchsorted(labels, y)
        if len(labels) > MAX_LABELS or len(y) == 0:
            return False
        u, i = self._clip_ids(user_ids), self._clip_ids(item_ids)
        lab = np.ascontiguousarray(inverse.astype(np.int32))
        check(self._lib.pmf_eval_set(self._h, len(y), ptr(u, C.c_int32), ptr(i, C.c_int32),
                                     ptr(y, C.c_double), ptr(lab, C.c_int32), len(labels)),
              "pmf_eval_set")
        self._eval_n, self._eval_labels = len(y), len(labels)
        return True

    def eval_sums(self, use_bias=False, offset=0.0):
        """Raw reductions over the stored validation set: [n, sum sq err, per-label
        sum |err| (MAX_LABELS), per-label count (MAX_LABELS)] as one float64 vector
        (additive across the shards of a multi-GPU run)."""
        sse = C.c_double(0.0)
        abs_l = np.zeros(MAX_LABELS, dtype=np.float64)
        cnt_l = np.zeros(MAX_LABELS, dtype=np.int64)
        check(self._lib.pmf_eval_run(self._h, int(use_bias), float(offset), C.byref(sse),
                                     ptr(abs_l, C.c_double), ptr(cnt_l, C.c_int64)), "pmf_eval_run")
        return np.concatenate([[float(self._eval_n), sse.value], abs_l, cnt_l.astype(np.float64)])

    @staticmethod
    def metrics_from_sums(sums):
        """(rmse, macro_mae) from an `eval_sums` vector (metrics.py:6-10, :37-51)."""
        n, sse = sums[0], sums[1]
        abs_l, cnt_l = sums[2:2 + MAX_LABELS], sums[2 + MAX_LABELS:2 + 2 * MAX_LABELS]
        seen = cnt_l > 0
        return float(np.sqrt(sse / n)), float(np.mean(abs_l[seen] / cnt_l[seen]))

    def eval_run(self, use_bias=False, offset=0.0):
        """(rmse, macro_mae) over the stored validation set."""
        return self.metrics_from_sums(self.eval_sums(use_bias, offset))

    def topk_items(self, user_ids, k, use_bias=False):
        """`use_bias`: False / 0, PREDICT_BIAS (scores + b_u + b_i) or PREDICT_SCALE (scores * s_u * s_i):
        the same flag `predict` takes, so the ranking follows the model's own score."""
        u = as_i32(user_ids, "user_ids")
        items = np.empty((len(u), k), dtype=np.int32)
        scores = np.empty((len(u), k), dtype=np.float64)
        check(self._lib.pmf_topk_items(self._h, len(u), ptr(u, C.c_int32), int(k), int(use_bias),
                                       ptr(items, C.c_int32), ptr(scores, C.c_double)), "pmf_topk_items")
        return items, scores

    # ---- profiling ------------------------------------------------------
    def prof_enable(self, on=True):
        check(self._lib.pmf_prof_enable(self._h, int(bool(on))), "pmf_prof_enable")

    def prof_reset(self):
        check(self._lib.pmf_prof_reset(self._h), "pmf_prof_reset")

    def gather_ceiling_ms(self, side, repeats=5):
        """Average ms of the gather-only twin of the Poisson/HPF half-sweep of `side` (its roofline)."""
        ms = C.c_double(0.0)
        check(self._lib.pmf_prof_gather_ceiling(self._h, side, int(repeats), C.byref(ms)), "pmf_prof_gather_ceiling")
        return ms.value

    def prof_get(self):
        """{kernel_name: (total_ms, launches)} since the last reset."""
        out = {}
        for k, name in enumerate(KERNEL_NAMES):
            ms, n = C.c_double(0.0), C.c_int64(0)
            check(self._lib.pmf_prof_get(self._h, k, C.byref(ms), C.byref(n)), "pmf_prof_get")
            out[name] = (ms.value, n.value)
        return out


__all__ = ["Context", "PmfError", "USER", "ITEM"]
