"""nmf_amd -- MI355X-native NMF solver engine with the raleng/nmf Python API.

    from nmf_amd import NMF
    nmf = NMF(data, components)
    nmf.factorize(method='mur', distance_type='eu')

The solver loops run as hand-written gfx950 HIP kernels behind the C ABI of
include/nmfx.h (libnmfx.so, built in-tree by `python -m nmf_amd.build`).
"""
from .nmf import NMF  # noqa: F401

__all__ = ['NMF']
