"""CPU oracle for the latent-factor update loop -- TEST INFRASTRUCTURE ONLY.

A NumPy restatement of the arithmetic in the reference's `src/models/*.py`
(citations are relative to /root/reference).  It exists to check the HIP path:
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  Nothing under `prob-matrix-factorization_amd/`
imports it, and the product never falls back to it.

Parity is PINNED: `tests/test_oracle_golden.py` checks every function here
against the vectors in `tests/golden/*.npz`, which were produced by running
the reference itself (`tests/golden/make_golden.py`).

Two forms of every half-sweep are provided:

  * `*_rows`    -- a per-row loop in the reference's summation order (one
                   NumPy call sequence per row, as `hpf_cavi.py:126-151`
                   does).  This is the "port" timed as `cpu_baseline`.
  * `*_segsum`  -- the same update as a segmented sum over the CSR-ordered
                   ratings (vectorised), used to check mid-size problems in
                   seconds.

State is a plain dict of float64 arrays; index structure is a CSR-like
`(ptr, pos)` pair per side where `pos` lists rating positions in their
original order (the reference's `_build_index_lists`, `hpf_cavi.py:97-107`).
"""
from __future__ import annotations

import numpy as np

RATE_FLOOR = 1e-10  # hpf_cavi.py:141, poisson_mf_cavi.py:155


# --------------------------------------------------------------------------
# index structure and dimensions
# --------------------------------------------------------------------------
def infer_dims(u, i):
    """n_users = max(u)+1, n_items = max(i)+1 from the TRAINING ids only
    (hpf_cavi.py:60-64; identical in every model)."""
    return int(np.max(u)) + 1, int(np.max(i)) + 1


def group_positions(ids, n_rows):
    """Positions of the ratings of each row, ascending within a row.

    Equivalent of `_build_index_lists` (hpf_cavi.py:97-107): the reference
    appends positions while scanning the ratings once, so each row's list is
    in ascending position order and duplicates are kept.  A stable sort gives
    the same lists without the O(nnz) interpreter loop."""
    ids = np.asarray(ids, dtype=np.int64)
    pos = np.argsort(ids, kind="stable")
    ptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(ids, minlength=n_rows), out=ptr[1:])
    return ptr, pos


# --------------------------------------------------------------------------
# initialisation (RNG draw order is part of the contract)
# --------------------------------------------------------------------------
def init_poisson(n_users, n_items, K, a0, b0, seed):
    """poisson_mf_cavi.py:50-71 -- two gamma draws: users then items."""
    rng = np.random.default_rng(seed)
    st = {}
    st["a_theta"] = a0 + rng.gamma(1.0, 0.1, size=(n_users, K))
    st["a_beta"] = a0 + rng.gamma(1.0, 0.1, size=(n_items, K))
    st["b_theta"] = b0 * np.ones((n_users, K))
    st["b_beta"] = b0 * np.ones((n_items, K))
    st["E_theta"] = st["a_theta"] / st["b_theta"]
    st["E_beta"] = st["a_beta"] / st["b_beta"]
    return st


def init_poisson_ext(n_users, n_items, K, a0, b0, seed):
    """poisson_mf_extended_cavi.py:57-75 -- four gamma draws: a_theta, a_beta,
    a_phi, a_psi; all rates start at b0."""
    rng = np.random.default_rng(seed)
    st = {}
    st["a_theta"] = a0 + rng.gamma(1.0, 0.1, size=(n_users, K))
    st["a_beta"] = a0 + rng.gamma(1.0, 0.1, size=(n_items, K))
    st["a_phi"] = a0 + rng.gamma(1.0, 0.1, size=n_users)
    st["a_psi"] = a0 + rng.gamma(1.0, 0.1, size=n_items)
    st["b_theta"] = b0 * np.ones((n_users, K))
    st["b_beta"] = b0 * np.ones((n_items, K))
    st["b_phi"] = b0 * np.ones(n_users)
    st["b_psi"] = b0 * np.ones(n_items)
    for name in ("theta", "beta", "phi", "psi"):
        st[f"E_{name}"] = st[f"a_{name}"] / st[f"b_{name}"]
    return st


def init_hpf(n_users, n_items, K, a, a_prime, b_prime, c, c_prime, d_prime, seed):
    """hpf_cavi.py:66-89 -- four gamma draws in the order a_theta, b_theta,
    a_beta, b_beta; xi/eta shapes are scalars (Appendix A.3 of SURVEY.md)."""
    rng = np.random.default_rng(seed)
    st = {}
    st["gamma_a_theta"] = a + rng.gamma(1.0, 0.1, size=(n_users, K))
    st["gamma_b_theta"] = b_prime + rng.gamma(1.0, 0.1, size=(n_users, K))
    st["gamma_a_beta"] = c + rng.gamma(1.0, 0.1, size=(n_items, K))
    st["gamma_b_beta"] = d_prime + rng.gamma(1.0, 0.1, size=(n_items, K))
    st["gamma_a_xi"] = a_prime + K * a
    st["gamma_b_xi"] = b_prime * np.ones(n_users)
    st["gamma_a_eta"] = c_prime + K * c
    st["gamma_b_eta"] = d_prime * np.ones(n_items)
    st["E_theta"] = st["gamma_a_theta"] / st["gamma_b_theta"]
    st["E_beta"] = st["gamma_a_beta"] / st["gamma_b_beta"]
    st["E_xi"] = st["gamma_a_xi"] / st["gamma_b_xi"]
    st["E_eta"] = st["gamma_a_eta"] / st["gamma_b_eta"]
    return st


def init_gaussian(n_users, n_items, K, seed, bias=True):
    """gaussian_mf_cavi_bias.py:52-67 (no-bias twin gaussian_mf_cavi.py:47-58):
    means 0.1*N(0,1) users then items, identity covariances, zero biases."""
    rng = np.random.default_rng(seed)
    st = {}
    st["m_theta"] = 0.1 * rng.standard_normal((n_users, K))
    st["m_beta"] = 0.1 * rng.standard_normal((n_items, K))
    st["V_theta"] = np.broadcast_to(np.eye(K), (n_users, K, K)).copy()
    st["V_beta"] = np.broadcast_to(np.eye(K), (n_items, K, K)).copy()
    if bias:
        st["m_user_bias"] = np.zeros(n_users)
        st["m_item_bias"] = np.zeros(n_items)
    return st


# --------------------------------------------------------------------------
# Poisson / HPF half-sweep  (rows a5 / a6 of SURVEY.md section 8)
# --------------------------------------------------------------------------
def gamma_half_sweep_rows(E_self, E_other, ptr, pos, other_ids, x, shape_prior, rate_prior):
    """One block-Jacobi half-sweep of the simplified Poisson CAVI update.

    For row r with observations Omega_r (positions pos[ptr[r]:ptr[r+1]]):
        rate_j   = max(E_other[o_j] . E_self[r], 1e-10)
        shape[r] = shape_prior + sum_j (x_j / rate_j) * E_other[o_j] * E_self[r]
        rate[r]  = rate_prior[r] + sum_j E_other[o_j]
    Empty rows fall back to the priors.  Follows poisson_mf_cavi.py:135-167 /
    :173-197 and hpf_cavi.py:126-151 / :162-185 (the four loops are one
    routine with the roles of the two sides swapped).  `rate_prior` is the
    scalar b0 (Poisson) or the vector E_xi / E_eta (HPF)."""
    n_rows, K = E_self.shape
    rate_prior = np.broadcast_to(np.asarray(rate_prior, dtype=np.float64), (n_rows,))
    shape = np.empty((n_rows, K))
    rate = np.empty((n_rows, K))
    for r in range(n_rows):
        sel = pos[ptr[r]:ptr[r + 1]]
        if sel.size == 0:
            shape[r] = shape_prior
            rate[r] = rate_prior[r]
            continue
        other = E_other[other_ids[sel]]
        mine = E_self[r]
        lam = other @ mine
        lam[lam < RATE_FLOOR] = RATE_FLOOR
        share = (x[sel][:, None] / lam[:, None]) * other * mine[None, :]
        shape[r] = shape_prior + np.sum(share, axis=0)
        rate[r] = rate_prior[r] + np.sum(other, axis=0)
    return shape, rate


def _segment_sum(values, ptr):
    """Sum `values` (nnz, K) over CSR segments; empty segments give zeros."""
    n_rows = ptr.size - 1
    out = np.zeros((n_rows,) + values.shape[1:], dtype=values.dtype)
    nonempty = ptr[1:] > ptr[:-1]
    if values.shape[0]:
        out[nonempty] = np.add.reduceat(values, ptr[:-1][nonempty], axis=0)
    return out


def gamma_half_sweep_segsum(E_self, E_other, ptr, pos, other_ids, x, shape_prior, rate_prior):
    """Vectorised form of `gamma_half_sweep_rows` (same update, reduceat
    summation order)."""
    n_rows, K = E_self.shape
    rate_prior = np.broadcast_to(np.asarray(rate_prior, dtype=np.float64), (n_rows,))
    row_of = np.repeat(np.arange(n_rows), np.diff(ptr))
    other = E_other[other_ids[pos]]
    mine = E_self[row_of]
    lam = np.einsum("nk,nk->n", other, mine)
    lam = np.maximum(lam, RATE_FLOOR)
    share = (x[pos] / lam)[:, None] * other * mine
    shape = shape_prior + _segment_sum(share, ptr)
    rate = rate_prior[:, None] + _segment_sum(other, ptr)
    return shape, rate


def gamma_ext_half_sweep_rows(E_self, S_self, E_other, S_other, ptr, pos, other_ids, x, a0, b0):
    """One half-sweep of the extended Poisson model x ~ Poisson(phi psi theta.beta)
    (poisson_mf_extended_cavi.py:108-160 users, :163-215 items).  Per non-empty row:
        a[r]   = a0 + sum_j x_j * E_other[o_j] * E_self[r] / (E_other[o_j].E_self[r])   (no clamp)
        b[r]   = b0 + sum_j S_other[o_j] * E_other[o_j]
        E[r]   = a[r] / b[r]                      (row-local Gauss-Seidel: used right below)
        sa[r]  = a0 + sum_j x_j ;  sb[r] = b0 + sum_j S_other[o_j] * (E_other[o_j].E[r]) ;  S[r] = sa/sb
    Empty rows: a, b, sa, sb fall back to the priors but E[r] and S[r] KEEP their
    values (the reference `continue`s before touching them).
    Returns (a, b, E, sa, sb, S)."""
    n_rows, K = E_self.shape
    a, b, E = np.empty((n_rows, K)), np.empty((n_rows, K)), E_self.copy()
    sa, sb, S = np.empty(n_rows), np.empty(n_rows), S_self.copy()
    for r in range(n_rows):
        sel = pos[ptr[r]:ptr[r + 1]]
        if sel.size == 0:
            a[r], b[r], sa[r], sb[r] = a0, b0, a0, b0
            continue
        other = E_other[other_ids[sel]]
        scale = S_other[other_ids[sel]]
        mine = E_self[r]
        dots = other @ mine
        a[r] = a0 + np.sum((x[sel][:, None] / dots[:, None]) * other * mine[None, :], axis=0)
        b[r] = b0 + np.sum(other * scale[:, None], axis=0)
        E[r] = a[r] / b[r]
        sa[r] = a0 + np.sum(x[sel])
        sb[r] = b0 + np.sum(scale * (other @ E[r]))
        S[r] = sa[r] / sb[r]
    return a, b, E, sa, sb, S


def poisson_ext_iteration(st, idx, u, i, x, a0, b0):
    """poisson_mf_extended_cavi.py:105-215: users (theta, phi) then items (beta, psi)."""
    (uptr, upos), (iptr, ipos) = idx
    (st["a_theta"], st["b_theta"], st["E_theta"], st["a_phi"], st["b_phi"], st["E_phi"]) = gamma_ext_half_sweep_rows(
        st["E_theta"], st["E_phi"], st["E_beta"], st["E_psi"], uptr, upos, i, x, a0, b0)
    (st["a_beta"], st["b_beta"], st["E_beta"], st["a_psi"], st["b_psi"], st["E_psi"]) = gamma_ext_half_sweep_rows(
        st["E_beta"], st["E_psi"], st["E_theta"], st["E_phi"], iptr, ipos, u, x, a0, b0)
    return st


def poisson_iteration(st, idx, u, i, x, a0, b0, sweep=gamma_half_sweep_rows):
    """One full PoissonMFCAVI iteration (poisson_mf_cavi.py:135-200): users
    from the old E_theta and current E_beta, then items from the NEW E_theta."""
    (uptr, upos), (iptr, ipos) = idx
    st["a_theta"], st["b_theta"] = sweep(st["E_theta"], st["E_beta"], uptr, upos, i, x, a0, b0)
    st["E_theta"] = st["a_theta"] / st["b_theta"]
    st["a_beta"], st["b_beta"] = sweep(st["E_beta"], st["E_theta"], iptr, ipos, u, x, a0, b0)
    st["E_beta"] = st["a_beta"] / st["b_beta"]
    return st


def hpf_iteration(st, idx, u, i, x, a, b_prime, c, d_prime, sweep=gamma_half_sweep_rows):
    """One full HPF_CAVI iteration (hpf_cavi.py:126-193): theta sweep with
    rate prior E_xi, then xi; beta sweep with rate prior E_eta, then eta.
    The xi/eta rate priors are b_prime / d_prime themselves (Appendix A.4)."""
    (uptr, upos), (iptr, ipos) = idx
    st["gamma_a_theta"], st["gamma_b_theta"] = sweep(
        st["E_theta"], st["E_beta"], uptr, upos, i, x, a, st["E_xi"])
    st["E_theta"] = st["gamma_a_theta"] / st["gamma_b_theta"]
    st["gamma_b_xi"] = b_prime + np.sum(st["E_theta"], axis=1)
    st["E_xi"] = st["gamma_a_xi"] / st["gamma_b_xi"]
    st["gamma_a_beta"], st["gamma_b_beta"] = sweep(
        st["E_beta"], st["E_theta"], iptr, ipos, u, x, c, st["E_eta"])
    st["E_beta"] = st["gamma_a_beta"] / st["gamma_b_beta"]
    st["gamma_b_eta"] = d_prime + np.sum(st["E_beta"], axis=1)
    st["E_eta"] = st["gamma_a_eta"] / st["gamma_b_eta"]
    return st


# --------------------------------------------------------------------------
# Gaussian half-sweeps  (rows a8 / a9)
# --------------------------------------------------------------------------
def gauss_factor_sweep_rows(m_self, V_self, m_other, V_other, ptr, pos, other_ids, x,
                            bias_self, bias_other, sigma2, eta2):
    """Mean/covariance update of one side, in place on copies.

    For a non-empty row r:  S = sum_j (V_other[o_j] + m_other[o_j] m_other[o_j]^T),
    P = I/eta2 + S/sigma2,  V[r] = inv(P),
    m[r] = (1/sigma2) * V[r] @ sum_j m_other[o_j] * (x_j - bias_self[r] - bias_other[o_j]).
    Empty rows keep their previous mean and covariance.  Follows
    gaussian_mf_cavi_bias.py:132-165 / :170-201; with zero bias vectors it is
    gaussian_mf_cavi.py:121-147 / :152-178."""
    n_rows, K = m_self.shape
    m_new, V_new = m_self.copy(), V_self.copy()
    eye = np.eye(K)
    for r in range(n_rows):
        sel = pos[ptr[r]:ptr[r + 1]]
        if sel.size == 0:
            continue
        o = other_ids[sel]
        resid = x[sel] - bias_self[r] - bias_other[o]
        mo = m_other[o]
        second = V_other[o] + np.einsum("nk,nl->nkl", mo, mo)
        P = (1.0 / eta2) * eye + (1.0 / sigma2) * second.sum(axis=0)
        V = np.linalg.inv(P)
        V_new[r] = V
        m_new[r] = (1.0 / sigma2) * V @ (mo * resid[:, None]).sum(axis=0)
    return m_new, V_new


def gauss_factor_sweep_segsum(m_self, V_self, m_other, V_other, ptr, pos, other_ids, x,
                              bias_self, bias_other, sigma2, eta2):
    """Vectorised normal-equation build + batched inverse (same update)."""
    n_rows, K = m_self.shape
    o = other_ids[pos]
    row_of = np.repeat(np.arange(n_rows), np.diff(ptr))
    mo = m_other[o]
    # sum_j m m^T per segment without materialising (nnz, K, K): loop over k
    S = _segment_sum(V_other[o].reshape(len(o), K * K), ptr).reshape(n_rows, K, K)
    for k in range(K):
        S[:, k, :] += _segment_sum(mo * mo[:, k:k + 1], ptr)
    resid = x[pos] - bias_self[row_of] - bias_other[o]
    w = _segment_sum(mo * resid[:, None], ptr)
    nonempty = np.diff(ptr) > 0
    P = S[nonempty] / sigma2 + np.eye(K) / eta2
    V = np.linalg.inv(P)
    m_new, V_new = m_self.copy(), V_self.copy()
    V_new[nonempty] = V
    m_new[nonempty] = (1.0 / sigma2) * np.einsum("nkl,nl->nk", V, w[nonempty])
    return m_new, V_new


def gauss_bias_sweep_rows(bias_self, bias_other, m_self, m_other, ptr, pos, other_ids, x,
                          sigma2, eta_bias2):
    """Scalar bias update of one side (gaussian_mf_cavi_bias.py:206-232 /
    :237-263): b[r] = var/sigma2 * sum_j (x_j - bias_other[o_j] - m_other[o_j].m_self[r]),
    var = 1/(1/eta_bias2 + n_r/sigma2).  Empty rows keep their value."""
    out = bias_self.copy()
    for r in range(bias_self.shape[0]):
        sel = pos[ptr[r]:ptr[r + 1]]
        if sel.size == 0:
            continue
        o = other_ids[sel]
        resid = x[sel] - bias_other[o] - m_other[o] @ m_self[r]
        var = 1.0 / ((1.0 / eta_bias2) + (len(sel) / sigma2))
        out[r] = (var / sigma2) * resid.sum()
    return out


def gauss_bias_sweep_segsum(bias_self, bias_other, m_self, m_other, ptr, pos, other_ids, x,
                            sigma2, eta_bias2):
    n_rows = bias_self.shape[0]
    cnt = np.diff(ptr)
    row_of = np.repeat(np.arange(n_rows), cnt)
    o = other_ids[pos]
    resid = x[pos] - bias_other[o] - np.einsum("nk,nk->n", m_other[o], m_self[row_of])
    tot = _segment_sum(resid[:, None], ptr)[:, 0]
    var = 1.0 / ((1.0 / eta_bias2) + (cnt / sigma2))
    out = bias_self.copy()
    nz = cnt > 0
    out[nz] = (var[nz] / sigma2) * tot[nz]
    return out


def gaussian_iteration(st, idx, u, i, x, sigma2, eta_theta2, eta_beta2, eta_bias2=None,
                       vectorised=False):
    """One full GaussianMFCAVI iteration: theta sweep (old biases), beta sweep
    (new theta), then -- bias model only -- user biases, item biases (new user
    biases).  Order as gaussian_mf_cavi_bias.py:129-263."""
    (uptr, upos), (iptr, ipos) = idx
    fsweep = gauss_factor_sweep_segsum if vectorised else gauss_factor_sweep_rows
    bsweep = gauss_bias_sweep_segsum if vectorised else gauss_bias_sweep_rows
    has_bias = eta_bias2 is not None
    bu = st["m_user_bias"] if has_bias else np.zeros(st["m_theta"].shape[0])
    bi = st["m_item_bias"] if has_bias else np.zeros(st["m_beta"].shape[0])
    st["m_theta"], st["V_theta"] = fsweep(st["m_theta"], st["V_theta"], st["m_beta"], st["V_beta"],
                                          uptr, upos, i, x, bu, bi, sigma2, eta_theta2)
    st["m_beta"], st["V_beta"] = fsweep(st["m_beta"], st["V_beta"], st["m_theta"], st["V_theta"],
                                        iptr, ipos, u, x, bi, bu, sigma2, eta_beta2)
    if has_bias:
        st["m_user_bias"] = bsweep(bu, bi, st["m_theta"], st["m_beta"], uptr, upos, i, x,
                                   sigma2, eta_bias2)
        st["m_item_bias"] = bsweep(bi, st["m_user_bias"], st["m_beta"], st["m_theta"], iptr, ipos,
                                   u, x, sigma2, eta_bias2)
    return st


# --------------------------------------------------------------------------
# Gaussian MF, MAP by gradient steps  (SURVEY.md 8(f) rank 4 -- NO reference
# counterpart: this restates the ENGINE's own definition, include/pmf_hip.h
# `pmf_gauss_sgd_sweep`, so parity of this mode is UNPINNED)
# --------------------------------------------------------------------------
def sgd_pieces(n, chunk=256):
    """Lengths of the pieces a row of n ratings is cut into (the engine's task
    builder): one piece up to `chunk`, else q = ceil(n / chunk) nearly equal ones."""
    if n <= chunk:
        return [n]
    q = -(-n // chunk)
    base, extra = divmod(n, q)
    return [base + (1 if c < extra else 0) for c in range(q)]


def gauss_sgd_half_sweep(f_self, f_other, b_self, b_other, ptr, pos, other_ids, x, lr, sigma2, eta2,
                         eta_bias2, chunk=256):
    """Every row walks through its ratings in input order with the other side fixed:
    e = x - b_r - b_o - f_r.f_o; f_r += lr (e f_o / sigma2 - f_r / (eta2 n_r));
    b_r += lr (e / sigma2 - b_r / (eta_bias2 n_r)).  Pieces of a long row start from the
    row's old value; the row moves by the length-weighted mean of their displacements.
    Returns new (f_self, b_self); b_* may be None (model without biases)."""
    f_new = f_self.copy()
    b_new = None if b_self is None else b_self.copy()
    for r in range(ptr.size - 1):
        lo, hi = int(ptr[r]), int(ptr[r + 1])
        n = hi - lo
        if n == 0:
            continue
        d_f, d_b, at = np.zeros(f_self.shape[1]), 0.0, lo
        for length in sgd_pieces(n, chunk):
            th = f_self[r].copy()
            b = 0.0 if b_self is None else float(b_self[r])
            for p in pos[at:at + length]:
                o = other_ids[p]
                be = f_other[o]
                e = x[p] - b - (0.0 if b_other is None else b_other[o]) - float(th @ be)
                th = th + lr * (e * be / sigma2 - th / (eta2 * n))
                if b_self is not None:
                    b = b + lr * (e / sigma2 - b / (eta_bias2 * n))
            d_f += length * (th - f_self[r])
            d_b += length * (b - (0.0 if b_self is None else float(b_self[r])))
            at += length
        f_new[r] = f_self[r] + d_f / n
        if b_self is not None:
            b_new[r] = b_self[r] + d_b / n
    return f_new, b_new


def gauss_sgd_epoch(st, idx, u, i, x, lr, sigma2, eta_theta2, eta_beta2, eta_bias2=None):
    """Users with the items fixed, then items with the new users (the CAVI order)."""
    (uptr, upos), (iptr, ipos) = idx
    bias = eta_bias2 is not None
    bu, bi = (st["m_user_bias"], st["m_item_bias"]) if bias else (None, None)
    st["m_theta"], new_bu = gauss_sgd_half_sweep(st["m_theta"], st["m_beta"], bu, bi, uptr, upos, i, x, lr, sigma2,
                                                 eta_theta2, eta_bias2 if bias else 1.0)
    if bias:
        st["m_user_bias"] = bu = new_bu
    st["m_beta"], new_bi = gauss_sgd_half_sweep(st["m_beta"], st["m_theta"], bi, bu, iptr, ipos, u, x, lr, sigma2,
                                                eta_beta2, eta_bias2 if bias else 1.0)
    if bias:
        st["m_item_bias"] = new_bi
    return st


# --------------------------------------------------------------------------
# predict / metrics  (rows a10 / a11)
# --------------------------------------------------------------------------
def predict_dot(A, B, user_ids, item_ids, bias_u=None, bias_i=None, offset=0.0):
    """hpf_cavi.py:215-231 / gaussian_mf_cavi_bias.py:291-316: ids outside the
    trained dimensions predict 0; `offset` (global_mean) is added to ALL rows."""
    user_ids = np.asarray(user_ids, dtype=np.int64)
    item_ids = np.asarray(item_ids, dtype=np.int64)
    out = np.zeros(len(user_ids))
    ok = (user_ids < A.shape[0]) & (item_ids < B.shape[0])
    uu, ii = user_ids[ok], item_ids[ok]
    val = np.sum(A[uu] * B[ii], axis=1)
    if bias_u is not None:
        val = bias_u[uu] + bias_i[ii] + val
    out[ok] = val
    return out + offset


def rmse(y_true, y_pred):
    """metrics.py:6-10."""
    return float(np.sqrt(np.mean((y_true - y_pred) ** 2)))


def mae(y_true, y_pred):
    """metrics.py:12-16."""
    return float(np.mean(np.abs(y_true - y_pred)))


def macro_mae(y_true, y_pred):
    """metrics.py:37-51: mean over distinct true labels (exact float equality)
    of the per-label mean absolute error."""
    per_label = [np.mean(np.abs(y_true[y_true == lab] - y_pred[y_true == lab]))
                 for lab in np.unique(y_true)]
    return float(np.mean(per_label))


def gaussian_eval(st, val_u, val_i, val_x, global_mean, bias=True):
    """gaussian_mf_cavi_bias.py:318-348: drop rows with unseen ids, compare on
    the original scale."""
    keep = (val_u < st["m_theta"].shape[0]) & (val_i < st["m_beta"].shape[0])
    if not np.any(keep):
        return float("nan"), float("nan")
    y = val_x[keep] + global_mean
    p = predict_dot(st["m_theta"], st["m_beta"], val_u[keep], val_i[keep],
                    st["m_user_bias"] if bias else None, st["m_item_bias"] if bias else None,
                    global_mean)
    return rmse(y, p), macro_mae(y, p)


def ext_predict(st, user_ids, item_ids):
    """poisson_mf_extended_cavi.py:239-259: phi_u psi_i theta_u.beta_i, 0 for unseen ids."""
    user_ids = np.asarray(user_ids, dtype=np.int64)
    item_ids = np.asarray(item_ids, dtype=np.int64)
    out = np.zeros(len(user_ids))
    ok = (user_ids < st["E_theta"].shape[0]) & (item_ids < st["E_beta"].shape[0])
    uu, ii = user_ids[ok], item_ids[ok]
    out[ok] = st["E_phi"][uu] * st["E_psi"][ii] * np.sum(st["E_theta"][uu] * st["E_beta"][ii], axis=1)
    return out


def gamma_eval(st, val_u, val_i, val_x):
    """hpf_cavi.py:233-241 / poisson_mf_cavi.py:243-251: no filtering -- rows
    with unseen ids count with a prediction of 0."""
    p = predict_dot(st["E_theta"], st["E_beta"], val_u, val_i)
    return rmse(val_x, p), macro_mae(val_x, p)


# --------------------------------------------------------------------------
# full fits with validation monitoring and the early-stop rules (row a12)
# --------------------------------------------------------------------------
def _stop_gamma(improvement, tol):
    """poisson_mf_cavi.py:213, hpf_cavi.py:207 -- fires on negative improvement too."""
    return tol is not None and improvement < tol


def _stop_gauss(improvement, tol):
    """gaussian_mf_cavi_bias.py:279 -- only a small NON-NEGATIVE improvement stops."""
    return 0 <= improvement < tol


def fit(kind, u, i, x, cfg, val=None, global_mean=0.0, vectorised=False):
    """Run `cfg['max_iter']` iterations of model `kind` ('poisson' | 'hpf' |
    'gauss_bias' | 'gauss' | 'poisson_ext'); returns (state, history).  `val` = (u, i, rating)."""
    u = np.asarray(u, dtype=np.int64)
    i = np.asarray(i, dtype=np.int64)
    x = np.asarray(x, dtype=np.float64)
    U, I = infer_dims(u, i)
    K, seed = cfg["n_factors"], cfg["random_state"]
    idx = (group_positions(u, U), group_positions(i, I))
    gsweep = gamma_half_sweep_segsum if vectorised else gamma_half_sweep_rows
    if kind == "poisson":
        st = init_poisson(U, I, K, cfg["a0"], cfg["b0"], seed)
    elif kind == "hpf":
        st = init_hpf(U, I, K, cfg["a"], cfg["a_prime"], cfg["b_prime"], cfg["c"],
                      cfg["c_prime"], cfg["d_prime"], seed)
    elif kind == "poisson_ext":
        st = init_poisson_ext(U, I, K, cfg["a0"], cfg["b0"], seed)
    else:
        st = init_gaussian(U, I, K, seed, bias=(kind == "gauss_bias"))
    hist = {"val_rmse": [], "val_macro_mae": [], "stopped_early": False, "iterations": 0}
    prev = None
    for it in range(1, cfg["max_iter"] + 1):
        if kind == "poisson":
            poisson_iteration(st, idx, u, i, x, cfg["a0"], cfg["b0"], gsweep)
        elif kind == "hpf":
            hpf_iteration(st, idx, u, i, x, cfg["a"], cfg["b_prime"], cfg["c"], cfg["d_prime"], gsweep)
        elif kind == "poisson_ext":
            poisson_ext_iteration(st, idx, u, i, x, cfg["a0"], cfg["b0"])
        else:
            gaussian_iteration(st, idx, u, i, x, cfg["sigma2"], cfg["eta_theta2"], cfg["eta_beta2"],
                               cfg.get("eta_bias2") if kind == "gauss_bias" else None, vectorised)
        hist["iterations"] = it
        if val is None:
            continue
        vu, vi, vx = (np.asarray(val[0], dtype=np.int64), np.asarray(val[1], dtype=np.int64),
                      np.asarray(val[2], dtype=np.float64))
        if kind == "poisson_ext":
            r, mm = rmse(vx, ext_predict(st, vu, vi)), float("nan")
            stop = _stop_gamma
        elif kind in ("poisson", "hpf"):
            r, mm = gamma_eval(st, vu, vi, vx)
            stop = _stop_gamma
        else:
            r, mm = gaussian_eval(st, vu, vi, vx, global_mean, bias=(kind == "gauss_bias"))
            stop = _stop_gauss
        hist["val_rmse"].append(r)
        hist["val_macro_mae"].append(mm)
        if prev is not None and stop(prev - r, cfg["tol"]):
            hist["stopped_early"] = True
            break
        prev = r
    return st, hist
