"""Input encoders on the hot path, same parameter names / state_dict keys
as the reference's: ASTNodeEncoder (dataset/utils.py:8-30), ogb AtomEncoder / BondEncoder
(ogb==1.2.6, used at dataset/mol.py:83-84; ogb is not installed here, so these are our own
modules with ogb's published layout: one nn.Embedding per categorical column, summed).
"""
import torch

ATOM_FEATURE_DIMS = [119, 4, 12, 12, 10, 6, 6, 2, 2]
BOND_FEATURE_DIMS = [5, 6, 2]


class ASTNodeEncoder(torch.nn.Module):
    """type_emb[x[:,0]] + attr_emb[x[:,1]] + depth_emb[min(depth, max_depth)].  Unlike the
    reference (dataset/utils.py:29) the caller's `depth` tensor is not clamped in place."""

    def __init__(self, emb_dim, num_nodetypes, num_nodeattributes, max_depth):
        super().__init__()
        self.max_depth = max_depth
        self.type_encoder = torch.nn.Embedding(num_nodetypes, emb_dim)
        self.attribute_encoder = torch.nn.Embedding(num_nodeattributes, emb_dim)
        self.depth_encoder = torch.nn.Embedding(self.max_depth + 1, emb_dim)

    def forward(self, x, depth):
        depth = depth.clamp(max=self.max_depth)
        return self.type_encoder(x[:, 0]) + self.attribute_encoder(x[:, 1]) + self.depth_encoder(depth)


class _SumEmbedding(torch.nn.Module):
    def _build(self, dims, emb_dim, list_name):
        embs = torch.nn.ModuleList()
        for dim in dims:
            emb = torch.nn.Embedding(dim, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            embs.append(emb)
        setattr(self, list_name, embs)
        return embs

    @staticmethod
    def _sum(embs, x):
        out = 0
        for i in range(x.shape[1]):
            out = out + embs[i](x[:, i])
        return out


class AtomEncoder(_SumEmbedding):
    def __init__(self, emb_dim):
        super().__init__()
        self._build(ATOM_FEATURE_DIMS, emb_dim, "atom_embedding_list")

    def forward(self, x):
        return self._sum(self.atom_embedding_list, x)


class BondEncoder(_SumEmbedding):
    def __init__(self, emb_dim):
        super().__init__()
        self._build(BOND_FEATURE_DIMS, emb_dim, "bond_embedding_list")

    def forward(self, edge_attr):
        return self._sum(self.bond_embedding_list, edge_attr)
