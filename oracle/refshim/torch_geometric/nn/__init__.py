import torch
from .inits import reset  # noqa


def _pool(x, batch, size, mean):
    size = int(batch.max().item() + 1) if size is None else int(size)
    out = torch.zeros(size, x.size(-1), dtype=x.dtype, device=x.device)
    out.index_add_(0, batch, x)
    if mean:
        cnt = torch.zeros(size, dtype=x.dtype, device=x.device)
        cnt.index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
        out = out / cnt.clamp(min=1).unsqueeze(-1)
    return out


def global_add_pool(x, batch, size=None):
    return _pool(x, batch, size, False)


def global_mean_pool(x, batch, size=None):
    return _pool(x, batch, size, True)


class JumpingKnowledge(torch.nn.Module):
    def __init__(self, mode, channels=None, num_layers=None):
        super().__init__()
        self.mode = mode.lower()
        assert self.mode in ['cat', 'max']

    def reset_parameters(self):
        pass

    def forward(self, xs):
        if self.mode == 'cat':
            return torch.cat(xs, dim=-1)
        return torch.stack(xs, dim=-1).max(dim=-1)[0]


class GINEConv(torch.nn.Module):  # imported by mp/molec_models.py, unused on the hot path
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError


class GINConv(GINEConv):
    pass
