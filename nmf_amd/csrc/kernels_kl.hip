// MUR with the Kullback-Leibler divergence (nmf/mur.py:24-27,40-43; utils.py:21-26).
#include "nmfx_internal.h"
#include "kernels_small.h"

int nmfx_mur_kl_phase_a(nmfx_engine* E, double, int64_t) { E->err = "MUR-KL: not built yet"; return NMFX_E_ARG; }
int nmfx_mur_kl_phase_b(nmfx_engine* E, double, int64_t, double, double, int64_t) { E->err = "MUR-KL: not built yet"; return NMFX_E_ARG; }
int nmfx_mur_kl_finish_a(nmfx_engine* E, int64_t) { E->err = "MUR-KL: not built yet"; return NMFX_E_ARG; }
