"""``utils.plotting`` of the reference (utils/plotting.py:16-319), result-dump half only.  The reference functions do two things: save
the test-set arrays as ``results_<model>/*.npy`` (the input of its evaluation notebooks) and draw matplotlib / t-SNE panels.  The
panels are visualisation (out of scope); the ``.npy`` files are written here under the reference's names
so that the notebooks' loaders find them.  Signatures are the reference's."""
import os

import numpy as np

__all__ = ["individual_cvs", "individual_challenge", "individual_proc", "visualize_latent"]


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def _dump(config, is_post, is_test, results, solution_xt, z, times, **named):
    if not is_test:
        return
    out = "results_{}".format(config.model)
    os.makedirs(out, exist_ok=True)
    tag = "post" if is_post else "prior"
    for k, v in named.items():
        np.save(os.path.join(out, k), _np(v))
    np.save(os.path.join(out, "times"), _np(times))
    get = (lambda k: results[k]) if isinstance(results, dict) else (lambda k: getattr(results, k))
    for k in ("mu_50", "mu_75", "mu_25"):
        np.save(os.path.join(out, "%s_%s" % (k, tag)), _np(get(k)))
    np.save(os.path.join(out, "solution_xt_%s" % tag), _np(solution_xt))
    np.save(os.path.join(out, "z_%s" % tag), _np(z))


def individual_cvs(observations, results, iext, rtpr, times, epoch, is_post, is_test, solution_xt, z, config):
    """utils/plotting.py:16-41 -> :117-126 (arrays saved for the test set; no figure)."""
    _dump(config, is_post, is_test, results, solution_xt, z, times, observations=observations, iext=iext, rtpr=rtpr)


def individual_challenge(observations, results, shedding, symptoms, times, epoch, is_post, is_test, solution_xt, z, config):
    """utils/plotting.py:44-72 -> :174-183."""
    _dump(config, is_post, is_test, results, solution_xt, z, times, observations=observations, shedding=shedding, symptoms=symptoms)


def individual_proc(results, observations, treatments, devices, config, epoch, times, is_post, is_test, z, solution_xt):
    """utils/plotting.py:203-243 (:217-227 are the saved arrays)."""
    _dump(config, is_post, is_test, results, solution_xt, z, times, observations=observations, treatments=treatments, devices=devices)


def visualize_latent(z_prior, z_post, config, epoch):
    """utils/plotting.py:302-319 draws a t-SNE scatter of prior vs posterior latents: visualisation only, nothing to save."""
    return None
