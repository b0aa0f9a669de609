"""Stand-in for torch-scatter 2.0.5 `scatter` (sum/mean/max/min) on dim=-2/0 with COO indices.
Semantics restated from the torch-scatter 2.0.5 documentation: output rows that receive nothing
are 0 for every reduce; mean divides by max(count, 1)."""
import torch


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce='sum'):
    assert out is None
    if dim < 0:
        dim = src.dim() + dim
    assert dim == 0 and src.dim() == 2, 'stand-in covers [E,F] row scatters only'
    n = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    res = torch.zeros(n, src.size(1), dtype=src.dtype, device=src.device)
    if reduce in ('sum', 'add'):
        return res.index_add_(0, index, src)
    if reduce == 'mean':
        res.index_add_(0, index, src)
        cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype))
        return res / cnt.clamp(min=1).unsqueeze(-1)
    if reduce in ('max', 'min'):
        idx = index.unsqueeze(-1).expand_as(src)
        return res.scatter_reduce_(0, idx, src, reduce='amax' if reduce == 'max' else 'amin',
                                   include_self=False)
    raise ValueError(reduce)


def segment_csr(*a, **k):
    raise NotImplementedError('CSR branch is dead code in the reference data pipeline')


def gather_csr(*a, **k):
    raise NotImplementedError('CSR branch is dead code in the reference data pipeline')
