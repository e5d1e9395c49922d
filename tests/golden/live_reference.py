#!/usr/bin/env python3
"""Run THE REFERENCE on a batch of problems (build container only; `/root/reference` does not exist on
the GPU box).  Helper of tests/test_live_reference_cpu.py, which starts it as a subprocess so that the
reference's `src` package and this repo's `src` package never meet in one interpreter.

    python tests/golden/live_reference.py problems.npz out.npz

`problems.npz`: for trial t the arrays `t{t}_u`, `t{t}_i`, `t{t}_x`, `t{t}_vu`, `t{t}_vi`, `t{t}_vx`,
`t{t}_qu`, `t{t}_qi` and a JSON string `t{t}_cfg` = {"kind", "config", "global_mean", "validate"}.
`out.npz`: the fitted attributes, `predict` on (qu, qi), the evaluate_* values and the captured stdout
of `fit` (verbose=True) per trial.  Data only -- no code travels."""
import contextlib
import io
import json
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, "/root/reference")
from src.models.gaussian_mf_cavi import GaussianMFCAVI as Gauss, GaussianMFCAVIConfig as GaussCfg  # noqa: E402
from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI as GaussBias, GaussianMFCAVIConfig as GaussBiasCfg  # noqa: E402
from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config  # noqa: E402
from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig  # noqa: E402
from src.models.poisson_mf_extended_cavi import PoissonMFExtendedCAVI, PoissonMFExtendedCAVIConfig  # noqa: E402

KINDS = {"hpf": (HPF_CAVI, HPF_CAVI_Config, ["gamma_a_theta", "gamma_b_theta", "gamma_a_beta", "gamma_b_beta", "E_theta",
                                             "E_beta", "E_xi", "E_eta"]),
         "poisson": (PoissonMFCAVI, PoissonMFCAVIConfig, ["a_theta", "b_theta", "a_beta", "b_beta", "E_theta", "E_beta"]),
         "poisson_ext": (PoissonMFExtendedCAVI, PoissonMFExtendedCAVIConfig,
                         ["a_theta", "b_theta", "a_beta", "b_beta", "a_phi", "b_phi", "a_psi", "b_psi", "E_theta", "E_beta", "E_phi",
                          "E_psi"]),
         "gauss_bias": (GaussBias, GaussBiasCfg, ["m_theta", "m_beta", "V_theta", "V_beta", "m_user_bias", "m_item_bias"]),
         "gauss": (Gauss, GaussCfg, ["m_theta", "m_beta", "V_theta", "V_beta"])}


def torch_case(d, t, meta, out):
    """The PyTorch HPF model: parameters drawn under torch.manual_seed, loss and its four gradient tables on a batch,
    a few Adam steps (the external loop of train_hpf_pytorch_full.py:96-108 without the shuffle), predict."""
    import torch
    from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config
    torch.manual_seed(meta["seed"])
    m = HPF_PyTorch(meta["n_users"], meta["n_items"], d[f"t{t}_user_counts"], d[f"t{t}_item_counts"],
                    HPF_PyTorch_Config(**meta["config"]))
    bu, bi = torch.from_numpy(d[f"t{t}_bu"]), torch.from_numpy(d[f"t{t}_bi"])
    br = torch.from_numpy(d[f"t{t}_br"])
    for name in ("theta_uncons", "beta_uncons", "xi_uncons", "eta_uncons"):
        out[f"t{t}_{name}"] = getattr(m, name).detach().numpy().copy()
    loss = m.loss(bu, bi, br)
    loss.backward()
    out[f"t{t}_loss"] = np.float64(loss.item())
    for name in ("theta_uncons", "beta_uncons", "xi_uncons", "eta_uncons"):
        out[f"t{t}_grad_{name}"] = getattr(m, name).grad.numpy().copy()
    opt = torch.optim.Adam(m.parameters(), lr=meta["config"]["lr"])
    for _ in range(meta["steps"]):
        opt.zero_grad()
        m.loss(bu, bi, br).backward()
        opt.step()
    m.eval()
    out[f"t{t}_predict"] = np.asarray(m.predict(d[f"t{t}_qu"], d[f"t{t}_qi"]), dtype=np.float64)
    out[f"t{t}_theta_after"] = m.theta.detach().numpy().copy()


def main(src, dst):
    d = np.load(src, allow_pickle=False)
    n = int(d["n_trials"])
    out = {}
    for t in range(n):
        meta = json.loads(str(d[f"t{t}_cfg"]))
        if meta["kind"] == "hpf_torch":
            torch_case(d, t, meta, out)
            continue
        if meta["kind"] == "metrics":
            from src.evaluation import metrics as M
            y, p = d[f"t{t}_y"], d[f"t{t}_p"]
            frame = pd.DataFrame({"u": d[f"t{t}_u"], "i": d[f"t{t}_i"], "rating": d[f"t{t}_x"]})
            th, be = d[f"t{t}_theta"], d[f"t{t}_beta"]
            out[f"t{t}_values"] = np.asarray([M.rmse(y, p), M.mae(y, p), M.macro_mae(y, p),
                                              M.GaussianLogPredictiveLikelihood(frame, th, be, meta["sigma"]),
                                              M.PoissonLogPredictiveLikelihood(frame, np.abs(th), np.abs(be))], dtype=np.float64)
            continue
        cls, cfg_cls, keys = KINDS[meta["kind"]]
        train = pd.DataFrame({"u": d[f"t{t}_u"], "i": d[f"t{t}_i"], "rating": d[f"t{t}_x"]})
        val = pd.DataFrame({"u": d[f"t{t}_vu"], "i": d[f"t{t}_vi"], "rating": d[f"t{t}_vx"]}) if meta["validate"] else None
        model = cls(cfg_cls(**meta["config"]))
        gm = meta["global_mean"]
        gauss = meta["kind"].startswith("gauss")
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            if gauss:
                model.fit(train, val_df=val, global_mean=gm)
            else:
                model.fit(train, val_df=val)
        for k in keys:
            out[f"t{t}_{k}"] = np.asarray(getattr(model, k), dtype=np.float64)
        qu, qi = d[f"t{t}_qu"], d[f"t{t}_qi"]
        out[f"t{t}_predict"] = model.predict(qu, qi, gm) if gauss else model.predict(qu, qi)
        vdf = pd.DataFrame({"u": d[f"t{t}_vu"], "i": d[f"t{t}_vi"], "rating": d[f"t{t}_vx"]})
        ebuf = io.StringIO()
        has_macro = hasattr(model, "evaluate_macro_mae")       # (the bias-free Gaussian class has no MacroMAE method)
        with contextlib.redirect_stdout(ebuf), np.errstate(all="ignore"):
            ev = [model.evaluate_rmse(vdf, gm) if gauss else model.evaluate_rmse(vdf)]
            if has_macro:
                ev.append(model.evaluate_macro_mae(vdf, gm) if gauss else model.evaluate_macro_mae(vdf))
        out[f"t{t}_evaluate"] = np.asarray(ev, dtype=np.float64)
        out[f"t{t}_stdout"] = np.asarray(buf.getvalue())
        out[f"t{t}_eval_stdout"] = np.asarray(ebuf.getvalue())
    np.savez(dst, **out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
