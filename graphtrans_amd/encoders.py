"""Input encoders on the hot path, same parameter names / state_dict keys
as the reference's: ASTNodeEncoder (dataset/utils.py:8-30), ogb AtomEncoder / BondEncoder
(ogb==1.2.6, used at dataset/mol.py:83-84; ogb is not installed here, so these are our own
modules with ogb's published layout: one nn.Embedding per categorical column, summed).
"""
import torch

from . import ops

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
        return ops.embed_sum([x[:, 0], x[:, 1], depth.reshape(-1)],
                             [self.type_encoder.weight, self.attribute_encoder.weight, self.depth_encoder.weight],
                             clamps=[None, None, self.max_depth])


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
        cols = x.shape[1]
        if cols > 16:
            raise ValueError("at most 16 categorical columns")
        return ops.embed_sum([x[:, i] for i in range(cols)], [embs[i].weight for i in range(cols)])


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
