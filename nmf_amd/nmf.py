"""`NMF(data, components).factorize(method=...)` -- the reference's class API
(nmf/nmf.py:7-135) on top of the MI355X engine."""
import os
from importlib import import_module

from . import utils

_METHODS = ('mur', 'anls', 'admm', 'ao_admm')


class NMF:
    """Non-negative matrix factorisation of 2-D `data` into `factors`
    components by MUR, ANLS, ADMM or AO-ADMM.

        nmf = NMF(data, factors)
        nmf.factorize(method='mur', **method_params)
        nmf.w, nmf.h            # also nmf.results.w / .h / .i / .obj_history
    """

    def __init__(self, data=None, factors=None, saving=True, param_file=None):
        self.data = data
        self.factors = factors
        self.saving = saving
        self.results = None
        self.w = None
        self.h = None
        if param_file is not None:
            try:
                self.method_params = import_module(param_file).method_params
            except ImportError:
                print('No parameter file found.')

    def factorize(self, method='mur', saving=False, **method_params):
        if method not in _METHODS:
            raise Exception('Method not known. Choose one from: mur anls admm ao_admm')
        solver = getattr(import_module('.' + method, __package__), method)
        self.results = solver(self.data, self.factors, **method_params)
        # README.md:22 promises nmf.w / nmf.h; the reference only sets .results
        self.w, self.h = self.results.w, self.results.h
        print('Factorization done.')
        if saving:
            self.save_factorization()

    def save_factorization(self, save_dir='./results', save_name=None):
        """Write results to `save_dir`/`save_name`.npz; the default name follows
        the grammar of nmf/nmf.py:95-126, e.g. nmf_ao_admm_3_eu_0:nn_0.5:l1n_random."""
        os.makedirs(save_dir, exist_ok=True)
        exp = self.results.experiment
        if save_name is None:
            with_prox = exp.method in ('admm', 'ao_admm')
            parts = ['nmf', str(exp.method), str(self.factors), str(exp.distance_type)]
            if exp.method == 'admm':
                parts.append(str(exp.rho))
            parts.append(f'{exp.lambda_w}:{exp.prox_w}' if with_prox else str(exp.lambda_w))
            parts.append(f'{exp.lambda_h}:{exp.prox_h}' if with_prox else str(exp.lambda_h))
            parts.append('nndsvd' + exp.nndsvd_init[1][0] if exp.nndsvd_init[0] else 'random')
            if exp.method == 'anls' and exp.fcnnls:
                parts.append('fcnnls')
            save_name = '_'.join(parts)
        utils.save_results(os.path.join(save_dir, save_name), w=self.results.w, h=self.results.h,
                           i=self.results.i, obj_history=self.results.obj_history,
                           experiment=exp._asdict())
