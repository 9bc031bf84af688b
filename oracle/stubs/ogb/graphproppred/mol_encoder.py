"""Stub of ogb 1.2.6 `AtomEncoder`/`BondEncoder` (requirement.yml:75; used at dataset/mol.py:83-84).

Published behaviour restated: one nn.Embedding per categorical feature column (xavier-uniform
initialised), output = sum over columns of embedding[column value].  Feature cardinalities are
OGB's `get_atom_feature_dims()` / `get_bond_feature_dims()`.
Test infrastructure only (see oracle/stubs/README.md).
"""
import torch

full_atom_feature_dims = [119, 4, 12, 12, 10, 6, 6, 2, 2]
full_bond_feature_dims = [5, 6, 2]


class AtomEncoder(torch.nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        self.atom_embedding_list = torch.nn.ModuleList()
        for dim in full_atom_feature_dims:
            emb = torch.nn.Embedding(dim, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)

    def forward(self, x):
        out = 0
        for i in range(x.shape[1]):
            out = out + self.atom_embedding_list[i](x[:, i])
        return out


class BondEncoder(torch.nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        self.bond_embedding_list = torch.nn.ModuleList()
        for dim in full_bond_feature_dims:
            emb = torch.nn.Embedding(dim, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            self.bond_embedding_list.append(emb)

    def forward(self, edge_attr):
        out = 0
        for i in range(edge_attr.shape[1]):
            out = out + self.bond_embedding_list[i](edge_attr[:, i])
        return out
