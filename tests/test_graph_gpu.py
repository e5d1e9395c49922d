"""HIP-graph capture / replay of sweep sequences (pmf_graph_*): a replayed iteration is the same
kernels with the same arguments, so results must be bit-identical to issuing the calls."""
import numpy as np
import pandas as pd
import pytest

from helpers import skewed_problem

pytestmark = pytest.mark.gpu


def _hpf_ctx(dtype="f32"):
    import pmf_hip
    from pmf_hip import ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER
    U, I, N, K = 2000, 300, 40000, 20
    u, i, x = skewed_problem(12, U, I, N)
    rng = np.random.default_rng(2)
    ctx = pmf_hip.Context(U, I, K, dtype=dtype)
    ctx.set_ratings(u, i, x)
    ctx.set_array(USER, ARR_FACTOR, rng.gamma(1.0, 0.3, (U, K)) + 0.1)
    ctx.set_array(ITEM, ARR_FACTOR, rng.gamma(1.0, 0.3, (I, K)) + 0.1)
    ctx.set_array(USER, ARR_PRIOR_RATE, np.ones(U)); ctx.set_array(ITEM, ARR_PRIOR_RATE, np.ones(I))
    return ctx, K


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_replayed_iterations_equal_issued_iterations(dtype):
    from pmf_hip import ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER

    def run(use_graph):
        ctx, K = _hpf_ctx(dtype)

        def it():
            ctx.gamma_sweep(USER, 0.3, 0.0, True, 0.3 + K * 0.3, 1.0)
            ctx.gamma_sweep(ITEM, 0.3, 0.0, True, 0.3 + K * 0.3, 1.0)
        it()
        if use_graph:
            with ctx.capture() as g:
                it()
            for _ in range(4):
                g.launch()
        else:
            for _ in range(4):
                it()
        out = [ctx.get_array(s, a) for s in (USER, ITEM) for a in (ARR_FACTOR, ARR_PRIOR_RATE)]
        ctx.close()
        return out
    for a, b in zip(run(False), run(True)):
        assert np.array_equal(a, b)


def test_capture_refuses_allocation_and_leaves_the_context_usable():
    import pmf_hip
    from pmf_hip import ARR_FACTOR, ITEM, USER
    ctx, K = _hpf_ctx()
    with pytest.raises(pmf_hip.PmfError, match="before capturing"):
        with ctx.capture():
            ctx.gamma_sweep(USER, 0.3, 0.0, True, 0.3 + K * 0.3, 1.0)   # first call: SHAPE / RATE not allocated yet
    ctx.gamma_sweep(USER, 0.3, 0.0, True, 0.3 + K * 0.3, 1.0)           # the stream is out of capture mode again
    assert np.isfinite(ctx.get_array(USER, ARR_FACTOR)).all()
    with pytest.raises(pmf_hip.PmfError):
        ctx._lib  # noqa: B018 (keep flake quiet)
        pmf_hip.check(ctx._lib.pmf_graph_launch(ctx._h, 7), "pmf_graph_launch")
    ctx.close()
    del ITEM


@pytest.mark.parametrize("kind", ["hpf", "gauss"])
def test_model_fit_with_graph_replay_equals_fit_without(kind, monkeypatch):
    from pmf_hip.synth import synth_ratings, train_val_split
    u, i, r = synth_ratings(3000, 400, 60000, seed=21)
    (tu, ti, tr), (vu, vi, vr) = train_val_split(u, i, r)
    shift = 1.0 if kind == "hpf" else -float(tr.mean())
    train = pd.DataFrame({"u": tu, "i": ti, "rating": tr + shift})
    val = pd.DataFrame({"u": vu, "i": vi, "rating": vr + shift})

    def fit():
        if kind == "hpf":
            from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
            m = HPF_CAVI(HPF_CAVI_Config(n_factors=12, max_iter=8, tol=None, verbose=False)).fit(train, val_df=val)
            return [m.E_theta, m.E_beta, m.E_xi, np.array(m.history_["val_rmse"])]
        from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
        m = GaussianMFCAVI(GaussianMFCAVIConfig(n_factors=12, max_iter=6, tol=-1.0, verbose=False)).fit(train, val_df=val)
        return [m.m_theta, m.m_beta, m.m_user_bias, m.V_beta, np.array(m.history_["val_rmse"])]
    monkeypatch.setenv("PMF_HIP_GRAPH", "1")
    with_graph = fit()
    monkeypatch.setenv("PMF_HIP_GRAPH", "0")
    without = fit()
    for a, b in zip(with_graph, without):
        assert np.array_equal(a, b)
