"""Build PRODUCT-side containers (cwn_amd.complex) from tests/golden/dummy_complexes.npz."""
import torch

from cwn_amd.complex import Cochain, Complex, ComplexBatch
from tests._golden import load, T

KEYS = ('x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries',
        'boundary_index', 'y')


def dummy_complex(name: str, device=None) -> Complex:
    g = load('dummy_complexes.npz')
    dim = int(g[f'{name}/dimension'])
    cochains = []
    for d in range(dim + 1):
        kw = {k: T(g[f'{name}/{d}/{k}']).clone() for k in KEYS if f'{name}/{d}/{k}' in g}
        cochains.append(Cochain(dim=d, **kw))
    y = T(g[f'{name}/y']).clone() if f'{name}/y' in g else None
    cx = Complex(*cochains, y=y)
    return cx.to(device) if device is not None else cx


def dummy_batch(names, max_dim=2, device=None) -> ComplexBatch:
    b = ComplexBatch.from_complex_list([dummy_complex(n) for n in names], max_dim=max_dim)
    return b.to(device) if device is not None else b


def list_names(which: str):
    return [str(n) for n in load('dummy_complexes.npz')[f'lists/{which}']]
