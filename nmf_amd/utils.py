"""Host-side utilities of the reference API surface (nmf/utils.py).

Only `nndsvd` (initialisation, SURVEY 2b K21: host LAPACK, not on the
per-iteration path) and `save_results` (the .npz on-disk format) live on the
host.  The per-iteration objective and the convergence test run on the device
(csrc/kernels_small.h); `convergence_message` only reproduces the reference's
stdout line for a stop rule the device already decided.
"""
import os

import numpy as np

QUIET = os.environ.get("NMF_AMD_QUIET", "0") == "1"


def say(*a):
    if not QUIET:
        print(*a)


def convergence_message(rule):
    """stdout of nmf/utils.py:8-11 for stop rule 1 or 2."""
    if rule in (1, 2):
        say('Algorithm converged ({}).'.format(rule))


def tol_digits(tol1, tol2):
    """Decimal places of the per-iteration print (nmf/mur.py:93-95)."""
    tol = min(tol1, tol2)
    return int(format(tol, 'e').split('-')[1]) if tol < 1 else 2


def _nndsvd_from_triplets(x, u, s, vt, rank, variant):
    """Boutsidis & Gallopoulos construction from the leading `rank` singular triplets
    (nmf/utils.py:51-93; invariant under the sign choice of a triplet: the positive and
    negative parts swap roles)."""
    w = np.zeros((x.shape[0], rank))
    h = np.zeros((rank, x.shape[1]))
    root = np.sqrt(s[:rank])
    w[:, 0] = root[0] * np.abs(u[:, 0])
    h[0, :] = root[0] * np.abs(vt[0, :])
    for c in range(1, rank):
        col, row = u[:, c], vt[c, :]
        cp, cn = np.maximum(col, 0), np.maximum(-col, 0)
        rp, rn = np.maximum(row, 0), np.maximum(-row, 0)
        ncp, ncn = np.linalg.norm(cp), np.linalg.norm(cn)
        nrp, nrn = np.linalg.norm(rp), np.linalg.norm(rn)
        if ncp * nrp >= ncn * nrn:
            scale = np.sqrt(s[c] * ncp * nrp)
            w[:, c], h[c, :] = scale / ncp * cp, scale / nrp * rp
        else:
            scale = np.sqrt(s[c] * ncn * nrn)
            w[:, c], h[c, :] = scale / ncn * cn, scale / nrn * rn
    if variant == 'mean':
        mu = np.mean(x)
        w[w == 0] = mu
        h[h == 0] = mu
    elif variant == 'random':
        mu = np.mean(x)
        fill = mu * np.random.random_sample(w.shape) / 100
        w = np.where(w == 0, fill, w)
        fill = mu * np.random.random_sample(h.shape) / 100
        h = np.where(h == 0, fill, h)
    return w, h


def nndsvd(x, rank=None, variant='zero'):
    """SVD based initialisation (Boutsidis & Gallopoulos), same call signature
    and RNG consumption as nmf/utils.py:36-93.  Returns float64 (w, h)."""
    u, s, vt = np.linalg.svd(x, full_matrices=False)
    if rank is None:
        rank = x.shape[1]
    return _nndsvd_from_triplets(x, u, s, vt, rank, variant)


def nndsvd_device(eng, x, rank, variant='zero'):
    """NNDSVD whose singular triplets come from the device (Engine.topk_svd on the uploaded V,
    f64 subspace iteration) instead of a full LAPACK SVD on the host: the only part of
    nmf/utils.py:36-93 whose cost grows like m n min(m, n).

    NMFX_NNDSVD=device: the device or an error (sweep cap 4000).  Otherwise ("auto"): a few hundred sweeps are
    allowed -- the filtered iteration needs 4 to 14 on matrices with any gap behind the rank-th singular value -- and
    when they do not converge (rank beyond the numerical rank: sigma_k sits in the noise bulk) the start is computed by
    the host's LAPACK SVD like the reference does, with a warning about the minutes that takes on a large matrix."""
    forced = os.environ.get("NMFX_NNDSVD", "auto") == "device"
    u, s, vt, sweeps, resid = eng.topk_svd(rank, max_sweeps=0 if forced else 300)
    if not resid <= 1e-9:
        if forced:
            raise RuntimeError(f'device SVD did not converge ({sweeps} sweeps, residual {resid:.2e}); '
                               'set NMFX_NNDSVD=host to use LAPACK on the host')
        import logging
        logging.warning('device SVD did not converge in %d sweeps (residual %.1e: rank %d lies beyond the numerical rank); '
                        'falling back to the full LAPACK SVD on the host like the reference -- slow for a %dx%d matrix',
                        sweeps, resid, rank, x.shape[0], x.shape[1])
        return nndsvd(np.asarray(x), rank, variant=variant)
    return _nndsvd_from_triplets(x, u, s, vt, rank, variant)


def nndsvd_on_device(x, k):
    """Where the singular triplets of NNDSVD are computed.  NMFX_NNDSVD = host | device | auto
    (default): auto takes the device for matrices of 2^22 elements and more, where the host's
    full LAPACK SVD (what nmf/utils.py:50 does) costs seconds to minutes, and keeps small
    problems on the host LAPACK path (bit-identical to the reference's own initialisation)."""
    mode = os.environ.get("NMFX_NNDSVD", "auto")
    if mode == "host":
        return False
    if mode == "device":
        return True
    return x.shape[0] * x.shape[1] >= (1 << 22) and k <= 128


def initial_factors(x, k, nndsvd_init, uniform=False, defer_device=False):
    """W then H from the GLOBAL numpy RNG in the reference's order
    (nmf/mur.py:105-109, nmf/ao_admm.py:19-23, nmf/admm.py:20-24 use |randn|;
    nmf/anls.py:101-105 uses rand), so identical seeds give identical starts.
    With defer_device=True returns None when the NNDSVD is to be computed from the device
    SVD once V is uploaded (`device_initial_factors`)."""
    if nndsvd_init[0]:
        if defer_device and nndsvd_on_device(x, k):
            return None
        return nndsvd(x, k, variant=nndsvd_init[1])
    if uniform:
        w = np.random.rand(x.shape[0], k)
        h = np.random.rand(k, x.shape[1])
    else:
        w = np.abs(np.random.randn(x.shape[0], k))
        h = np.abs(np.random.randn(k, x.shape[1]))
    return w, h


def device_initial_factors(eng, x, k, nndsvd_init, init):
    """`init` from initial_factors(..., defer_device=True), completed on the device if deferred."""
    return init if init is not None else nndsvd_device(eng, x, k, variant=nndsvd_init[1])


def save_results(save_str, w, h, i, obj_history, experiment):
    """Same .npz keys as nmf/utils.py:96-105 (w, h, i, obj_history, experiment)."""
    np.savez(save_str, w=w, h=h, i=i, obj_history=obj_history, experiment=experiment)
    say('Results saved in {}.'.format(save_str))
