import torch


class _SumEmbed(torch.nn.Module):
    def __init__(self, emb_dim, dims):
        super().__init__()
        self.embs = torch.nn.ModuleList([torch.nn.Embedding(d, emb_dim) for d in dims])

    def forward(self, x):
        return sum(e(x[:, i]) for i, e in enumerate(self.embs))


class AtomEncoder(_SumEmbed):
    def __init__(self, emb_dim):
        super().__init__(emb_dim, [119, 4, 12, 12, 10, 6, 6, 2, 2])


class BondEncoder(_SumEmbed):
    def __init__(self, emb_dim):
        super().__init__(emb_dim, [5, 6, 2])
