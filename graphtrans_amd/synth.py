"""Synthetic, dataset-shaped batches: no datasets exist on the GPU box.

All generators use numpy `default_rng(seed)` only, so CPU-oracle and GPU runs see identical
inputs.  Shapes/layouts follow the reference's dataset adapters:
  code2_like   — dataset/code.py + dataset/utils.py:89-141 (augment_edge): x (N,2) i64,
                 node_depth (N,1) i64, edge_attr (E,2) f32 in {0,1}^2, edges per graph ordered
                 [ast, ast^-1, next-token, next-token^-1], y_arr (B,5) i64.
  molpcba_like — dataset/mol.py: x (N,9) i64, edge_attr (E,3) i64, y (B,128) f32 with NaN.
  nci1_like    — dataset/tud.py: x (N,37) one-hot f32, no edge features, y (B,) i64.
  er_stress    — BASELINE.json configs[4]: Erdos-Renyi G(n, p) graphs, Code2-style edge_attr.
"""
import numpy as np
import torch

from .data import Batch

ATOM_DIMS = [119, 4, 12, 12, 10, 6, 6, 2, 2]
BOND_DIMS = [5, 6, 2]


def _finish(xs, eis, eas, sizes, extra):
    offs = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    edge_index = np.concatenate([ei + o for ei, o in zip(eis, offs)], axis=1) if eis else np.zeros((2, 0), np.int64)
    batch = np.repeat(np.arange(len(sizes)), sizes)
    d = dict(x=torch.from_numpy(np.concatenate(xs, 0)), edge_index=torch.from_numpy(edge_index.astype(np.int64)),
             batch=torch.from_numpy(batch.astype(np.int64)))
    d["edge_attr"] = torch.from_numpy(np.concatenate(eas, 0)) if eas is not None else None
    d.update(extra)
    b = Batch(**d)
    b._num_graphs = len(sizes)
    return b


def _code2_graphs(B, seed, mean_nodes, sigma, min_nodes, max_nodes, num_nodetypes, num_nodeattributes, num_vocab,
                  max_seq_len):
    """Raw per-graph arrays as the OGB dataset stores them BEFORE the reference's per-sample transform
    (dataset/code.py:97-101): AST edges only, `node_is_attributed`, DFS-ordered nodes."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.round(rng.lognormal(np.log(mean_nodes), sigma, B)), min_nodes, max_nodes).astype(np.int64)
    graphs = []
    for n in sizes:
        n = int(n)
        # random rooted tree, parent < child, parent among the 8 most recent nodes (AST-like)
        lo = np.maximum(np.arange(1, n) - 8, 0)
        parent = lo + (rng.random(n - 1) * (np.arange(1, n) - lo)).astype(np.int64)
        child = np.arange(1, n)
        depth = np.zeros(n, np.int64)
        for c, p in zip(child, parent):
            depth[c] = depth[p] + 1
        ast = np.stack([parent, child])
        is_attr = rng.random(n) < 0.4
        x = np.stack([rng.integers(0, num_nodetypes, n), rng.integers(0, num_nodeattributes, n)], 1).astype(np.int64)
        graphs.append(dict(x=x, edge_index=ast.astype(np.int64), node_depth=depth.reshape(-1, 1),
                           node_is_attributed=is_attr.astype(np.int64).reshape(-1, 1)))
    y_arr = rng.integers(0, num_vocab, (B, max_seq_len)).astype(np.int64)
    for g, y in zip(graphs, y_arr):
        g["y_arr"] = y.reshape(1, -1)
    return sizes, graphs


def code2_raw(B=256, seed=0, mean_nodes=105.0, sigma=0.6, min_nodes=11, max_nodes=2000,
              num_nodetypes=98, num_nodeattributes=10030, num_vocab=5002, max_seq_len=5):
    """List of raw graphs (numpy dicts) for data.GraphStore; collating graphs 0..B-1 of it on the device
    reproduces code2_like(B, seed) exactly."""
    return _code2_graphs(B, seed, mean_nodes, sigma, min_nodes, max_nodes, num_nodetypes, num_nodeattributes,
                         num_vocab, max_seq_len)[1]


def code2_like(B=256, seed=0, mean_nodes=105.0, sigma=0.6, min_nodes=11, max_nodes=2000,
               num_nodetypes=98, num_nodeattributes=10030, num_vocab=5002, max_seq_len=5):
    sizes, graphs = _code2_graphs(B, seed, mean_nodes, sigma, min_nodes, max_nodes, num_nodetypes,
                                  num_nodeattributes, num_vocab, max_seq_len)
    xs, eis, eas, depths = [], [], [], []
    for g in graphs:
        ast = g["edge_index"]
        attributed = np.nonzero(g["node_is_attributed"].reshape(-1))[0]
        nt = np.stack([attributed[:-1], attributed[1:]]) if attributed.size > 1 else np.zeros((2, 0), np.int64)
        ei = np.concatenate([ast, ast[::-1], nt, nt[::-1]], axis=1)
        ea = np.concatenate([
            np.zeros((ast.shape[1], 2)), np.stack([np.zeros(ast.shape[1]), np.ones(ast.shape[1])], 1),
            np.stack([np.ones(nt.shape[1]), np.zeros(nt.shape[1])], 1), np.ones((nt.shape[1], 2))], 0).astype(np.float32)
        xs.append(g["x"]); eis.append(ei); eas.append(ea); depths.append(g["node_depth"].reshape(-1))
    y_arr = np.concatenate([g["y_arr"] for g in graphs], 0)
    return _finish(xs, eis, eas, sizes, dict(
        node_depth=torch.from_numpy(np.concatenate(depths).reshape(-1, 1)), y_arr=torch.from_numpy(y_arr)))


def _random_undirected(rng, n, m):
    """m distinct undirected edges (no self loops) on n nodes, returned in both directions."""
    if n < 2 or m <= 0:
        return np.zeros((2, 0), np.int64)
    m = int(min(m, n * (n - 1) // 2))
    got = set()
    while len(got) < m:
        a = rng.integers(0, n, 2 * (m - len(got)) + 4)
        b = rng.integers(0, n, a.size)
        for u, v in zip(a, b):
            if u != v:
                got.add((min(u, v), max(u, v)))
                if len(got) == m:
                    break
    e = np.array(sorted(got), np.int64).T
    perm = rng.permutation(e.shape[1])
    e = e[:, perm]
    return np.concatenate([e, e[::-1]], axis=1)


def _molpcba_graphs(B, seed, num_tasks):
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.round(rng.normal(26, 6, B)), 2, 332).astype(np.int64)
    graphs = []
    for n in sizes:
        n = int(n)
        ei = _random_undirected(rng, n, int(round(1.08 * n)))
        half = ei.shape[1] // 2
        ea_half = np.stack([rng.integers(0, d, half) for d in BOND_DIMS], 1).astype(np.int64)
        x = np.stack([rng.integers(0, d, n) for d in ATOM_DIMS], 1).astype(np.int64)
        graphs.append(dict(x=x, edge_index=ei, edge_attr=np.concatenate([ea_half, ea_half], 0)))
    y = rng.integers(0, 2, (B, num_tasks)).astype(np.float32)
    y[rng.random((B, num_tasks)) < 0.6] = np.nan
    for g, row in zip(graphs, y):
        g["y"] = row.reshape(1, -1)
    return sizes, graphs


def molpcba_raw(B=256, seed=0, num_tasks=128):
    """Raw graphs for data.GraphStore (no per-sample transform for Molpcba, dataset/mol.py)."""
    return _molpcba_graphs(B, seed, num_tasks)[1]


def molpcba_like(B=256, seed=0, num_tasks=128):
    sizes, graphs = _molpcba_graphs(B, seed, num_tasks)
    y = np.concatenate([g["y"] for g in graphs], 0)
    return _finish([g["x"] for g in graphs], [g["edge_index"] for g in graphs], [g["edge_attr"] for g in graphs],
                   sizes, dict(y=torch.from_numpy(y)))


def nci1_like(B=32, seed=0, num_features=37):
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.round(rng.normal(30, 13, B)), 3, 111).astype(np.int64)
    xs, eis = [], []
    for n in sizes:
        n = int(n)
        x = np.zeros((n, num_features), np.float32)
        x[np.arange(n), rng.integers(0, num_features, n)] = 1.0
        xs.append(x); eis.append(_random_undirected(rng, n, int(round(1.08 * n))))
    y = rng.integers(0, 2, B).astype(np.int64)
    return _finish(xs, eis, None, sizes, dict(y=torch.from_numpy(y)))


def er_stress(B=256, seed=0, n=512, avg_deg=8.0, feat_dim=256, num_classes=2):
    """BASELINE.json configs[4]: G(n, p=avg_deg/(n-1)), both directions stored; x (N,feat) f32
    through a Linear node encoder; Code2-style 2-column {0,1} edge_attr."""
    rng = np.random.default_rng(seed)
    p = avg_deg / (n - 1)
    xs, eis, eas = [], [], []
    iu = np.triu_indices(n, 1)
    for _ in range(B):
        sel = rng.random(iu[0].size) < p
        e = np.stack([iu[0][sel], iu[1][sel]]).astype(np.int64)
        ei = np.concatenate([e, e[::-1]], axis=1)
        eis.append(ei)
        eas.append(rng.integers(0, 2, (ei.shape[1], 2)).astype(np.float32))
        xs.append(rng.standard_normal((n, feat_dim), dtype=np.float32))
    y = rng.integers(0, num_classes, B).astype(np.int64)
    return _finish(xs, eis, eas, np.full(B, n, np.int64), dict(y=torch.from_numpy(y)))


def tiny_mixed(seed=0, sizes=(7, 1, 12, 5), feat="code2", num_nodetypes=11, num_nodeattributes=13,
               num_features=6, max_depth_gen=25):
    """Small hand-sized batches for golden vectors / unit tests: includes a 1-node graph, an
    isolated node, multi-edges and a hub node."""
    rng = np.random.default_rng(seed)
    xs, eis, eas, depths = [], [], [], []
    for n in sizes:
        n = int(n)
        m = 0 if n < 2 else int(rng.integers(n, 3 * n))
        src = rng.integers(0, max(n, 1), m)
        dst = rng.integers(0, max(n, 1), m)
        if n > 3:  # hub: many edges into node 0; node n-1 isolated
            dst[: m // 3] = 0
            keep = (src != n - 1) & (dst != n - 1)
            src, dst = src[keep], dst[keep]
        ei = np.stack([src, dst]).astype(np.int64).reshape(2, -1)
        eis.append(ei)
        m = ei.shape[1]
        if feat == "code2":
            xs.append(np.stack([rng.integers(0, num_nodetypes, n), rng.integers(0, num_nodeattributes, n)], 1).astype(np.int64))
            eas.append(rng.integers(0, 2, (m, 2)).astype(np.float32))
            depths.append(rng.integers(0, max_depth_gen, n).astype(np.int64))
        elif feat == "mol":
            xs.append(np.stack([rng.integers(0, d, n) for d in ATOM_DIMS], 1).astype(np.int64))
            eas.append(np.stack([rng.integers(0, d, m) for d in BOND_DIMS], 1).astype(np.int64).reshape(m, 3))
        elif feat == "tud":
            x = np.zeros((n, num_features), np.float32)
            x[np.arange(n), rng.integers(0, num_features, n)] = 1.0
            xs.append(x)
        elif feat == "dense":
            xs.append(rng.standard_normal((n, num_features)).astype(np.float32))
            eas.append(rng.standard_normal((m, 2)).astype(np.float32))
        else:
            raise ValueError(feat)
    extra = {}
    if feat == "code2":
        extra["node_depth"] = torch.from_numpy(np.concatenate(depths).reshape(-1, 1))
    return _finish(xs, eis, eas if feat != "tud" else None, np.array(sizes, np.int64), extra)
