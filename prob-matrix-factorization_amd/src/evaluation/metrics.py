"""Host-side metrics with the reference's names and semantics
(reference: src/evaluation/metrics.py).  Plain NumPy on host arrays; the
per-iteration validation monitor inside `fit` uses the fused device reduction
(`pmf_eval_run`) instead and only finishes the two divisions here."""
import numpy as np


def rmse(y_true, y_pred):
    """sqrt(mean squared error)  -- reference metrics.py:6-10."""
    diff = np.asarray(y_true) - np.asarray(y_pred)
    return np.sqrt(np.mean(diff * diff))


def mae(y_true, y_pred):
    """mean absolute error  -- reference metrics.py:12-16."""
    return np.mean(np.abs(np.asarray(y_true) - np.asarray(y_pred)))


def macro_mae(y_true, y_pred):
    """Mean over the distinct true labels of the per-label MAE (labels compared
    by exact float equality)  -- reference metrics.py:37-51."""
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    labels, inverse = np.unique(y_true, return_inverse=True)
    abs_err = np.abs(y_true - y_pred)
    per_label = [np.mean(abs_err[inverse == k]) for k in range(len(labels))]
    return np.mean(per_label)


def GaussianLogPredictiveLikelihood(df, theta, beta, sigma):
    """Sum of log N(rating | theta_u . beta_i, sigma^2)  -- reference metrics.py:18-35."""
    mean = np.einsum("nk,nk->n", theta[df.u], beta[df.i])
    var = sigma ** 2
    return np.sum(-0.5 * np.log(2 * np.pi * var) - (df.rating - mean) ** 2 / (2 * var))


def PoissonLogPredictiveLikelihood(df, theta, beta, epsilon=1e-10):
    """Sum of log Poisson(rating | max(theta_u . beta_i, eps))  -- reference metrics.py:53-66."""
    from scipy.special import gammaln
    lam = np.maximum(np.einsum("nk,nk->n", theta[df.u], beta[df.i]), epsilon)
    return np.sum(df.rating * np.log(lam) - lam - gammaln(df.rating + 1))
