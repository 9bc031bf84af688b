"""oracle/graph_struct.py — numpy restatement of the index structures the HIP path builds.

*** TEST INFRASTRUCTURE.  Not product code (see oracle/reference_math.py header). ***

The reference never materialises these: it scatters over the unsorted `edge_index`
(modules/conv.py:28,63 via PyG MessagePassing) and recomputes `degree(row)` every layer
(conv.py:57).  The HIP path sorts once per batch; "bit-exact edge_index scatter indices"
(BASELINE.json north_star) is pinned here as: CSR-by-destination / CSC-by-source built with a
STABLE sort of the edge list (ties keep original edge order), integer-exact.
"""
import numpy as np


def graph_struct(edge_index, batch, num_nodes=None, num_graphs=None):
    """Returns dict of int64/int32 arrays:
    ptr      (B+1)  graph offsets from the sorted `batch` vector (PyG collation)
    in_ptr   (N+1)  CSR by destination col = edge_index[1]
    in_eid   (E)    original edge ids, stable order within each destination
    in_src   (E)    edge_index[0][in_eid]
    out_ptr  (N+1)  CSC by source row = edge_index[0]
    out_eid  (E)    original edge ids, stable order within each source
    out_dst  (E)    edge_index[1][out_eid]
    deg_out  (N)    #{k : row_k == v}  (GCN's deg = deg_out + 1, conv.py:57)
    deg_in   (N)    #{k : col_k == v}
    """
    ei = np.asarray(edge_index)
    batch = np.asarray(batch)
    n = int(batch.shape[0]) if num_nodes is None else int(num_nodes)
    b = (int(batch[-1]) + 1 if batch.size else 0) if num_graphs is None else int(num_graphs)
    row, col = ei[0].astype(np.int64), ei[1].astype(np.int64)
    counts = np.bincount(batch, minlength=b).astype(np.int64)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    in_eid = np.argsort(col, kind="stable").astype(np.int64)
    out_eid = np.argsort(row, kind="stable").astype(np.int64)
    deg_in = np.bincount(col, minlength=n).astype(np.int64)
    deg_out = np.bincount(row, minlength=n).astype(np.int64)
    in_ptr = np.concatenate([[0], np.cumsum(deg_in)]).astype(np.int64)
    out_ptr = np.concatenate([[0], np.cumsum(deg_out)]).astype(np.int64)
    return dict(ptr=ptr, in_ptr=in_ptr, in_eid=in_eid, in_src=row[in_eid], out_ptr=out_ptr,
                out_eid=out_eid, out_dst=col[out_eid], deg_out=deg_out, deg_in=deg_in)


def pad_index(ptr, max_input_len):
    """Integer part of pad_batch (modules/utils.py:5-29): S, kept length per graph and the
    first kept node per graph (graphs keep their LAST min(n_b, S) nodes)."""
    ptr = np.asarray(ptr, np.int64)
    n = np.diff(ptr)
    S = int(min(int(n.max()), int(max_input_len)))
    kept = np.minimum(n, S)
    first = ptr[1:] - kept
    return S, kept, first
