"""PNANodeEmbedding (modules/pna/pna_module.py:16-78) and the PNAConv it instantiates.

The reference's conv is PyG 1.6.3's `PNAConv(towers=4, divide_input=True)` (third-party, not in the
tree); its math is restated from the in-tree copy modules/pna_layer.py:131-167 and
modules/pna/{aggregators,scalers}.py.  Parameter names / state_dict keys follow PyG:
`pre_nns.{t}.0.{weight (F,2F),bias}`, `post_nns.{t}.0.{weight (F_out,(A*S+1)F),bias}`, `lin.{weight,bias}`,
and `batch_norms.{i}.module.*` (PyG's BatchNorm wraps nn.BatchNorm1d as `.module`).

Data path: the per-edge pre-Linear is split into per-node terms (U = x A^T + b for the target role,
V = x B^T for the source role; two small batched GEMMs), the four aggregators run in ONE HIP pass
over the CSR (gt_pna_aggregate_fwd/bwd), the degree scalers are per-node scalars applied while the
post-Linear input is assembled.
"""
import math

import torch
import torch.nn as nn

from ... import ops
from ..gnn_module import batch_structure
from ..norm import BatchNorm1d

_AGG_SLOT = {"mean": 0, "max": 1, "min": 2, "std": 3}


class _Restack(torch.autograd.Function):
    """out[:, i] = src[:, map[i]] (map == src.shape[1] selects an appended zero); the map is injective on the used
    source columns, so the backward is a gather through the inverse map instead of torch's sort-based index_add."""

    @staticmethod
    def forward(ctx, src, fmap, inv):
        ctx.save_for_backward(inv)
        return torch.cat([src, src.new_zeros(src.shape[0], 1)], 1).index_select(1, fmap)

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return torch.cat([g, g.new_zeros(g.shape[0], 1)], 1).index_select(1, inv), None, None


class PNAConv(nn.Module):
    def __init__(self, in_channels, out_channels, aggregators, scalers, deg, edge_dim=None, towers=1, pre_layers=1,
                 post_layers=1, divide_input=False):
        super().__init__()
        if edge_dim is not None or pre_layers != 1 or post_layers != 1:
            raise NotImplementedError("the reference uses PNAConv without edge features and with single-layer pre/post nets")
        if divide_input:
            assert in_channels % towers == 0
        assert out_channels % towers == 0
        for a in aggregators:
            if a not in _AGG_SLOT:
                raise NotImplementedError(f"aggregator {a}")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aggregators, self.scalers = list(aggregators), list(scalers)
        self.towers, self.divide_input = towers, divide_input
        self.F_in = in_channels // towers if divide_input else in_channels
        self.F_out = out_channels // towers
        # the kernels move 16-byte chunks of a tower's row: per-tower widths that are not multiples of 4 (the reference's default
        # gnn_emb_dim 300 with 4 towers: F = 75, modules/pna/pna_module.py:43-51) run on zero-padded tower rows (forward())
        self.F_pad = (self.F_in + 3) // 4 * 4
        if self.F_in != self.F_out and (self.F_in % 4 or self.F_out % 4):
            raise ValueError("graphtrans_amd PNAConv: unequal per-tower widths must be multiples of 4; got "
                             f"F_in = {self.F_in}, F_out = {self.F_out}")
        # the post-Linear is evaluated per scaler block (see forward); the x_i columns and the bias are NOT scaled, so
        # they ride in a block of their own unless the first scaler is the identity
        self._blocks = list(self.scalers) if self.scalers and self.scalers[0] == "identity" else [None] + list(self.scalers)
        deg = torch.as_tensor(deg).to(torch.float)
        # avg_deg over the degree HISTOGRAM tensor, exactly as PyG 1.6.3 / modules/pna_layer.py:92-97 do
        self.avg_deg = {"lin": deg.mean().item(), "log": (deg + 1).log().mean().item(), "exp": deg.exp().mean().item()}
        self.pre_nns = nn.ModuleList()
        self.post_nns = nn.ModuleList()
        for _ in range(towers):
            self.pre_nns.append(nn.Sequential(nn.Linear(2 * self.F_in, self.F_in)))
            self.post_nns.append(nn.Sequential(nn.Linear((len(aggregators) * len(scalers) + 1) * self.F_in, self.F_out)))
        self.lin = nn.Linear(out_channels, out_channels)

    def _scales(self, deg):
        """per-node scaler factors (modules/pna/scalers.py:10-31), deg = in-degree (N,1,1)."""
        out = []
        for s in self._blocks:
            if s is None or s == "identity":
                out.append(None)
            elif s == "amplification":
                out.append(torch.log(deg + 1) / self.avg_deg["log"])
            elif s == "attenuation":
                sc = self.avg_deg["log"] / torch.log(deg + 1)
                out.append(torch.where(deg == 0, torch.ones_like(sc), sc))
            elif s == "linear":
                out.append(deg / self.avg_deg["lin"])
            elif s == "inverse_linear":
                sc = self.avg_deg["lin"] / deg
                out.append(torch.where(deg == 0, torch.ones_like(sc), sc))
            else:
                raise ValueError(s)
        return out

    def _post_maps(self, device):
        """Gather maps that restack post_nns[t].weight (F_out, (A*S+1)F) into (S*F_out, 5F): block s holds the
        columns of scaler s over the kernel's [mean|max|min|std] slots (zeros for unused aggregators) and, for s = 0
        only, the x_i columns.  Then  post(cat[x, agg*s_0, agg*s_1, ...]) = sum_s scale_s (x) Y[:, s]."""
        key = str(device)
        if getattr(self, "_maps_key", None) == key:
            return self._wmap, self._bmap
        Fi, Fo, A, S = self.F_in, self.F_out, len(self.aggregators), len(self._blocks)
        cols = (A * len(self.scalers) + 1) * Fi
        zero_w = Fo * cols   # index of the appended zero
        wmap = torch.full((S * Fo, 5 * Fi), zero_w, dtype=torch.int64)
        o = torch.arange(Fo).view(Fo, 1)
        f = torch.arange(Fi).view(1, Fi)
        first = S - len(self.scalers)   # 1 when block 0 only carries x_i and the bias
        for s in range(S):
            rows = slice(s * Fo, (s + 1) * Fo)
            if s == 0:
                wmap[rows, 0:Fi] = o * cols + f
            if s < first:
                continue
            for ai, a in enumerate(self.aggregators):
                slot = _AGG_SLOT[a]
                wmap[rows, Fi + slot * Fi:Fi + (slot + 1) * Fi] = o * cols + Fi + (s - first) * A * Fi + ai * Fi + f
        bmap = torch.full((S * Fo,), Fo, dtype=torch.int64)
        bmap[:Fo] = torch.arange(Fo)
        # inverse maps for the backward (every source element lands at most once): source j <- restacked position
        # inv[j], or the appended zero when j is not used
        wflat = wmap.reshape(-1)
        winv = torch.full((zero_w,), wflat.numel(), dtype=torch.int64)
        used = wflat < zero_w
        winv[wflat[used]] = torch.nonzero(used).reshape(-1)
        binv = torch.arange(Fo)
        self._wmap, self._bmap, self._maps_key = wflat.to(device), bmap.to(device), key
        self._winv, self._binv = winv.to(device), binv.to(device)
        return self._wmap, self._bmap

    def forward(self, x, edge_index, edge_attr=None, graph=None):
        if edge_attr is not None:
            raise NotImplementedError("PNAConv with edge features is not on the reference's path")
        gs = graph
        if gs is None:
            from ...graph import GraphStructure
            gs = GraphStructure.build(edge_index, torch.zeros(x.shape[0], dtype=torch.int64, device=x.device), num_graphs=1)
        N, T, Fi, Fo = x.shape[0], self.towers, self.F_in, self.F_out
        S = len(self._blocks)
        if Fi % 4:
            return self._forward_padded(x, gs)
        xt = (x.view(N, T, Fi) if self.divide_input else x.view(N, 1, Fi).expand(N, T, Fi)).contiguous()
        Wp = torch.stack([m[0].weight for m in self.pre_nns])  # (T, F, 2F): [A | B] on [x_i || x_j]
        bp = torch.stack([m[0].bias for m in self.pre_nns])    # (T, F)
        # the per-edge pre-Linear splits into per-node terms: U = x A^T + b (target role), V = x B^T (source role)
        U = ops.tower_linear(xt, Wp[:, :, :Fi].contiguous(), bp).view(N, T * Fi)
        V = ops.tower_linear(xt, Wp[:, :, Fi:].contiguous(), None).view(N, T * Fi)
        agg4 = ops.pna_aggregate(U, V, gs, T).view(N, T, 4 * Fi)         # [mean | max | min | std]
        # post-Linear without materialising the (A*S+1)F-wide scaled copies: one GEMM on [x | agg] per tower,
        # the degree scalers (per-node scalars) are applied to its S output blocks
        wmap, bmap = self._post_maps(x.device)
        Wq = torch.stack([m[0].weight for m in self.post_nns])           # (T, F_out, (A*S+1)F)
        bq = torch.stack([m[0].bias for m in self.post_nns])             # (T, F_out)
        Wst = _Restack.apply(Wq.reshape(T, -1), wmap, self._winv).view(T, S * Fo, 5 * Fi)
        bst = _Restack.apply(bq, bmap, self._binv)
        Y = ops.tower_linear(torch.cat([xt, agg4], dim=-1), Wst, bst).view(N, T, S, Fo)
        # the degree scalers depend on the batch's graph structure only: computed by the first layer, reused by the rest
        cache = getattr(gs, "_pna_scales", None)
        key = (tuple(self._blocks), self.avg_deg["log"], self.avg_deg["lin"])
        if cache is None or cache[0] != key:
            deg = (gs.in_ptr[1:] - gs.in_ptr[:-1]).to(torch.float32).view(-1, 1)
            cols = [torch.ones_like(deg) if sc is None else sc for sc in self._scales(deg)]
            cache = (key, torch.cat(cols, dim=1).contiguous())   # (N, S)
            try:
                gs._pna_scales = cache
            except AttributeError:
                pass
        out = ops.scale_combine(Y, cache[1])   # sum_s scale_s * Y[:, :, s]
        return ops.linear_module(self.lin, out.reshape(N, T * Fo))


    def _forward_padded(self, x, gs):
        """Per-tower width F not a multiple of 4 (F = 75 for the reference's default gnn_emb_dim 300 with 4 towers): every tower row is
        zero-padded to Fp = ceil4(F) -- inputs, the pre-Linear's weight blocks and bias, the post-Linear's operand columns and
        output rows.  Padded columns carry U = V = 0, so their aggregates are 0 / 0 / 0 / sqrt(1e-5) and meet zero weight columns in
        the post-Linear; padded outputs are cut off before `lin`.  Same kernels as the aligned path on (N, T, Fp) operands; torch's
        pad / slice ops and their autograd do the re-layout (this configuration is not on the fused path)."""
        import torch.nn.functional as Fn
        N, T, F = x.shape[0], self.towers, self.F_in
        Fp, pad = self.F_pad, self.F_pad - self.F_in
        A_, S_ = len(self.aggregators), len(self.scalers)
        xt = x.view(N, T, F) if self.divide_input else x.view(N, 1, F).expand(N, T, F)
        xt = Fn.pad(xt, (0, pad)).contiguous()                                       # (N, T, Fp)
        Wp = torch.stack([m[0].weight for m in self.pre_nns])                          # (T, F, 2F)
        bp = Fn.pad(torch.stack([m[0].bias for m in self.pre_nns]), (0, pad))          # (T, Fp)
        Wa = Fn.pad(Wp[:, :, :F], (0, pad, 0, pad)).contiguous()                       # (T, Fp, Fp)
        Wb = Fn.pad(Wp[:, :, F:], (0, pad, 0, pad)).contiguous()
        U = ops.tower_linear(xt, Wa, bp).view(N, T * Fp)
        V = ops.tower_linear(xt, Wb, None).view(N, T * Fp)
        agg = ops.pna_aggregate(U, V, gs, T).view(N, T, 4, Fp)                         # kernel slots [mean | max | min | std]
        sel = agg[:, :, [_AGG_SLOT[a] for a in self.aggregators], :]                   # (N, T, A, Fp) in the module's aggregator order
        deg = (gs.in_ptr[1:] - gs.in_ptr[:-1]).to(torch.float32).view(-1, 1, 1, 1)
        scs = []
        for sname in self.scalers:
            if sname == "identity":
                scs.append(torch.ones_like(deg))
            elif sname == "amplification":
                scs.append(torch.log(deg + 1) / self.avg_deg["log"])
            elif sname == "attenuation":
                sc = self.avg_deg["log"] / torch.log(deg + 1)
                scs.append(torch.where(deg == 0, torch.ones_like(sc), sc))
            elif sname == "linear":
                scs.append(deg / self.avg_deg["lin"])
            elif sname == "inverse_linear":
                sc = self.avg_deg["lin"] / deg
                scs.append(torch.where(deg == 0, torch.ones_like(sc), sc))
            else:
                raise ValueError(sname)
        scaled = torch.cat([sel * sc for sc in scs], dim=2)                            # (N, T, S*A, Fp): scaler-major like PyG's cat
        inp = torch.cat([xt.unsqueeze(2), scaled], dim=2).reshape(N, T, (S_ * A_ + 1) * Fp).contiguous()
        Wq = torch.stack([m[0].weight for m in self.post_nns]).view(T, self.F_out, S_ * A_ + 1, F)
        Wq = Fn.pad(Wq, (0, pad, 0, 0, 0, pad)).reshape(T, Fp, (S_ * A_ + 1) * Fp).contiguous()   # output rows and operand blocks padded
        bq = Fn.pad(torch.stack([m[0].bias for m in self.post_nns]), (0, pad))
        out = ops.tower_linear(inp, Wq, bq)[:, :, :self.F_out].reshape(N, T * self.F_out)
        return ops.linear_module(self.lin, out)


class BatchNorm(nn.Module):
    """PyG 1.6.3 `torch_geometric.nn.BatchNorm`: nn.BatchNorm1d held as `.module` (state_dict keys)."""

    def __init__(self, in_channels):
        super().__init__()
        self.module = BatchNorm1d(in_channels)

    def forward(self, x, relu=False):
        return self.module(x, relu=relu)


class PNANodeEmbedding(nn.Module):
    @staticmethod
    def add_args(parser):
        group = parser.add_argument_group("PNANet configs")
        group.add_argument("--aggregators", type=str, nargs="+", default=["mean", "max", "min", "std"])
        group.add_argument("--scalers", type=str, nargs="+", default=["identity", "amplification", "attenuation"])
        group.add_argument("--post_layers", type=int, default=1)
        group.add_argument("--add_edge", type=str, default="none")
        group.set_defaults(gnn_residual=True)
        group.set_defaults(gnn_dropout=0.3)
        group.set_defaults(gnn_emb_dim=70)
        group.set_defaults(gnn_num_layer=4)

    def __init__(self, node_encoder, args):
        super().__init__()
        self.num_layer = args.gnn_num_layer
        self.max_seq_len = args.max_seq_len
        self.aggregators = args.aggregators
        self.scalers = args.scalers
        self.residual = args.gnn_residual
        self.drop_ratio = args.gnn_dropout
        self.graph_pooling = args.graph_pooling
        self.node_encoder = node_encoder
        self.layers = nn.ModuleList([
            PNAConv(args.gnn_emb_dim, args.gnn_emb_dim, aggregators=self.aggregators, scalers=self.scalers, deg=args.deg,
                    towers=4, divide_input=True) for _ in range(self.num_layer)])
        self.batch_norms = nn.ModuleList([BatchNorm(args.gnn_emb_dim) for _ in range(self.num_layer)])

    def forward(self, batched_data, perturb=None):
        x, edge_index = batched_data.x, batched_data.edge_index
        node_depth = batched_data.node_depth if hasattr(batched_data, "node_depth") else None
        gs = batch_structure(batched_data)
        encoded_node = self.node_encoder(x) if node_depth is None else self.node_encoder(x, node_depth.view(-1))
        x = encoded_node + perturb if perturb is not None else encoded_node
        for conv, batch_norm in zip(self.layers, self.batch_norms):
            h = batch_norm(conv(x, edge_index, graph=gs), relu=True)  # F.relu(batch_norm(conv(x)))  (:73)
            if self.residual:
                x = h + x
            else:  # the reference leaves x unchanged without the residual (h is dropped, :73-76)
                x = x
            x = ops.dropout(x, self.drop_ratio, training=self.training)   # F.dropout (:78)
        return x
