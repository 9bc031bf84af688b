"""Mini-batch assembly: `augment_edge` (reference dataset/utils.py:89-141) + PyG
collation (main.py:149-152), on the device from an HBM-resident graph store (csrc/collate.hip).

CPU part: the numpy oracle (oracle/collate.py) against the G11 fixtures, whose augmented edges were
produced by the reference's own `augment_edge`.  GPU part: `GraphStore.collate` against the fixtures
and the oracle, bit-exact (all integer work; the float edge_attr takes only the values 0 and 1).
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from graphtrans_amd import synth
from oracle import collate as oc

NODE_KEYS = ("x", "node_depth", "node_is_attributed")
EDGE_KEYS = ("edge_attr",)
GRAPH_KEYS = ("y", "y_arr")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    store = {k[6:]: z[k] for k in z.files if k.startswith("store.")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out.")}
    return store, z["in.ids"], out


def unpack(store):
    """store layout (node_ptr/edge_ptr + concatenated arrays) -> list of raw graphs"""
    graphs = []
    npz, ep = store["node_ptr"], store["edge_ptr"]
    for g in range(npz.size - 1):
        d = {"edge_index": store["edge_index"][:, ep[g]:ep[g + 1]]}
        for k in NODE_KEYS:
            if k in store:
                d[k] = store[k][npz[g]:npz[g + 1]]
        for k in EDGE_KEYS:
            if k in store:
                d[k] = store[k][ep[g]:ep[g + 1]]
        for k in GRAPH_KEYS:
            if k in store:
                d[k] = store[k][g:g + 1]
        graphs.append(d)
    return graphs


def same(a, b):
    """bitwise equality incl. NaN labels"""
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def check(got, want):
    for k, v in want.items():
        if k == "ptr" and k not in got:
            continue
        assert same(got[k], v), k


@pytest.mark.parametrize("name", ["G11_collate_code2", "G11_collate_mol"])
def test_oracle_collate_matches_golden(name):
    store, ids, want = load(name)
    graphs = unpack(store)
    got = oc.collate([graphs[i] for i in ids], augment="node_is_attributed" in store)
    check(got, want)


def test_oracle_augment_edge_properties():
    """the four blocks of dataset/utils.py:101-136 on a hand-made graph"""
    ei = np.array([[0, 0, 1, 1], [1, 4, 2, 3]])
    flag = np.array([0, 1, 1, 0, 1])
    out, ea = oc.augment_edge(ei, flag)
    assert out.tolist() == [[0, 0, 1, 1, 1, 4, 2, 3, 1, 2, 2, 4], [1, 4, 2, 3, 0, 0, 1, 1, 2, 4, 1, 2]]
    assert ea.tolist() == [[0, 0]] * 4 + [[0, 1]] * 4 + [[1, 0]] * 2 + [[1, 1]] * 2
    out, ea = oc.augment_edge(np.zeros((2, 0), np.int64), np.array([1]))
    assert out.shape == (2, 0) and ea.shape == (0, 2)


def test_synth_raw_and_batch_agree():
    """code2_like(B, seed) is by construction collate(code2_raw(B, seed)): the raw store and the ready batch
    the benchmarks use describe the same graphs."""
    raw = synth.code2_raw(B=9, seed=3, mean_nodes=30.0)
    b = synth.code2_like(B=9, seed=3, mean_nodes=30.0)
    got = oc.collate(raw, augment=True)
    for k in ("x", "edge_index", "edge_attr", "batch", "node_depth", "y_arr"):
        assert same(got[k], getattr(b, k).numpy()), k
    raw = synth.molpcba_raw(B=7, seed=3)
    b = synth.molpcba_like(B=7, seed=3)
    got = oc.collate(raw)
    for k in ("x", "edge_index", "edge_attr", "batch", "y"):
        assert same(got[k], getattr(b, k).numpy()), k


# ---------------------------------------------------------------------------------------------- GPU
def gpu_collate(graphs, ids):
    from graphtrans_amd.data import GraphStore
    store = GraphStore(graphs)
    b = store.collate(ids)
    torch.cuda.synchronize()
    return store, b, {k: getattr(b, k).cpu().numpy() for k in b.keys() if getattr(b, k) is not None}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["G11_collate_code2", "G11_collate_mol"])
def test_hip_collate_matches_golden(name):
    store, ids, want = load(name)
    _, b, got = gpu_collate(unpack(store), ids)
    check(got, want)
    assert b.num_graphs == ids.size and np.array_equal(b._sizes, np.diff(want["ptr"]))


@pytest.mark.gpu
def test_hip_collate_full_batch_is_the_benchmark_batch():
    """BASELINE configs[2] size: 256 Code2-shaped graphs; the device result is the batch bench.py trains on."""
    raw = synth.code2_raw(B=256, seed=0)
    b = synth.code2_like(B=256, seed=0)
    _, _, got = gpu_collate(raw, np.arange(256))
    for k in ("x", "edge_index", "edge_attr", "batch", "node_depth", "y_arr"):
        assert same(got[k], getattr(b, k).numpy()), k


@pytest.mark.gpu
def test_hip_collate_shuffled_subsets_and_model_input():
    """random subsets in random order (a sampler's output), then straight into the graph-structure build"""
    from graphtrans_amd.graph import GraphStructure
    from oracle.graph_struct import graph_struct
    raw = synth.code2_raw(B=64, seed=5, mean_nodes=40.0)
    rng = np.random.default_rng(0)
    from graphtrans_amd.data import GraphStore
    store = GraphStore(raw)
    for B in (1, 2, 17, 64):
        ids = rng.permutation(64)[:B]
        want = oc.collate([raw[i] for i in ids], augment=True)
        b = store.collate(ids)
        for k in ("x", "edge_index", "edge_attr", "batch", "node_depth", "y_arr"):
            assert same(getattr(b, k).cpu().numpy(), want[k]), (B, k)
        gs = GraphStructure.build(b.edge_index, b.batch, num_graphs=b.num_graphs, sizes=b._sizes)
        gs.validate()
        ref = graph_struct(want["edge_index"], want["batch"], num_graphs=B)
        assert np.array_equal(gs.in_src.cpu().numpy()[:gs.E], ref["in_src"])
        assert np.array_equal(gs.graph_ptr.cpu().numpy(), ref["ptr"])


@pytest.mark.gpu
def test_hip_collate_edge_cases():
    from graphtrans_amd.data import GraphStore
    raw = synth.code2_raw(B=4, seed=2, mean_nodes=15.0)
    raw[1]["node_is_attributed"][:] = 0            # no next-token chain at all
    raw[2]["node_is_attributed"][:] = 1            # every node on the chain
    raw[3]["node_is_attributed"][:] = 0
    raw[3]["node_is_attributed"][0] = 1            # a single attributed node: still no edge
    raw.append(dict(x=np.array([[1, 2]]), edge_index=np.zeros((2, 0), np.int64), node_depth=np.zeros((1, 1), np.int64),
                    node_is_attributed=np.ones((1, 1), np.int64), y_arr=np.arange(5).reshape(1, 5)))
    store = GraphStore(raw)
    for ids in ([4], [4, 4, 0], [1, 2, 3, 4, 0], []):
        want = oc.collate([raw[i] for i in ids], augment=True) if ids else None
        b = store.collate(ids)
        torch.cuda.synchronize()
        if not ids:
            assert b.x.shape == (0, 2) and b.edge_index.shape == (2, 0) and b.num_graphs == 0
            continue
        for k in ("x", "edge_index", "edge_attr", "batch", "node_depth", "y_arr"):
            assert same(getattr(b, k).cpu().numpy(), want[k]), (ids, k)
    with pytest.raises(IndexError):
        store.collate([5])


@pytest.mark.gpu
def test_hip_attr_rank_long():
    """the one-off prefix scan over a store larger than one scan chunk (8192 nodes)"""
    from graphtrans_amd import _lib
    rng = np.random.default_rng(1)
    for n in (0, 1, 8191, 8192, 8193, 100003):
        flag = rng.integers(0, 3, n).astype(np.int64)   # values other than 1 do not count (utils.py:118-123)
        d = torch.from_numpy(flag).cuda()
        rank = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        _lib.launch("gt_attr_rank", d.data_ptr() if n else None, n, rank.data_ptr(), torch.cuda.current_stream().cuda_stream)
        want = np.concatenate([[0], np.cumsum(flag == 1)])
        assert np.array_equal(rank.cpu().numpy(), want), n
