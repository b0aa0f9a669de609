"""Loaders for tests/golden/*.npz (data written by oracle/gen_golden.py from the live reference).
Builds the ORACLE's dict data model; product-side loaders live in tests/_product.py."""
import os
from functools import lru_cache

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
KEYS = ('x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries',
        'boundary_index', 'y', 'batch')


@lru_cache(maxsize=None)
def load(name):
    with np.load(os.path.join(GOLD, name)) as z:
        return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.asarray(a))


def complex_dict(arrs, prefix):
    """-> oracle complex dict (see oracle/cwn_oracle.py)."""
    dim = int(arrs[f'{prefix}/dimension'])
    cochains = []
    for d in range(dim + 1):
        c = {k: (T(arrs[f'{prefix}/{d}/{k}']) if f'{prefix}/{d}/{k}' in arrs else None) for k in KEYS}
        c['dim'] = d
        for k in ('num_cells', 'num_cells_up', 'num_cells_down'):
            key = f'{prefix}/{d}/{k}'
            c[k] = int(arrs[key]) if key in arrs else None
        cochains.append(c)
    y = T(arrs[f'{prefix}/y']) if f'{prefix}/y' in arrs else None
    return {'dimension': dim, 'y': y, 'cochains': cochains}


def dummy_complex(name):
    return complex_dict(load('dummy_complexes.npz'), name)


def params_dict(arrs, prefix):
    out = {}
    for k in ('x', 'up_index', 'down_index', 'boundary_index', 'up_attr', 'down_attr', 'boundary_attr'):
        key = f'{prefix}/{k}'
        out[k] = T(arrs[key]) if key in arrs else None
    return out


def state_dict(arrs, prefix):
    pre = prefix + '/'
    return {k[len(pre):]: T(v) for k, v in arrs.items() if k.startswith(pre)}
