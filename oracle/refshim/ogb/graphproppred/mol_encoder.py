"""Stand-in for ogb.graphproppred.mol_encoder (absent here).  Published OGB convention: an atom has
9 integer feature columns, a bond 3; the encoder is the SUM of one Embedding per column.  Parameter
names follow ogb (atom_embedding_list / bond_embedding_list) so state_dicts are interchangeable."""
import torch

ATOM_DIMS = [119, 4, 12, 12, 10, 6, 6, 2, 2]
BOND_DIMS = [5, 6, 2]


class AtomEncoder(torch.nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        self.atom_embedding_list = torch.nn.ModuleList()
        for d in ATOM_DIMS:
            emb = torch.nn.Embedding(d, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)

    def forward(self, x):
        return sum(self.atom_embedding_list[i](x[:, i]) for i in range(x.shape[1]))


class BondEncoder(torch.nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        self.bond_embedding_list = torch.nn.ModuleList()
        for d in BOND_DIMS:
            emb = torch.nn.Embedding(d, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            self.bond_embedding_list.append(emb)

    def forward(self, edge_attr):
        return sum(self.bond_embedding_list[i](edge_attr[:, i]) for i in range(edge_attr.shape[1]))
