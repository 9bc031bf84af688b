"""Data-parallel fused path with two real ranks: both processes share cuda:0 and reduce over gloo (a 1-GPU box cannot
host two RCCL ranks), which exercises the in-backward three-range all-reduce of the flat gradient buffer
(engine.py -> GradSync.reduce_flat) and finish(); the result must equal the average of the two shards' gradients
computed one after the other in a single process (BatchNorm statistics are per rank by design)."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _args():
    return SimpleNamespace(gnn_virtual_node=True, gnn_num_layer=3, gnn_emb_dim=64, gnn_JK="cat", gnn_dropout=0.0,
                           gnn_residual=False, gnn_type="gcn", pretrained_gnn=None, freeze_gnn=None, d_model=32, nhead=4,
                           dim_feedforward=64, transformer_dropout=0.0, transformer_activation="relu", num_encoder_layers=2,
                           max_input_len=1000, transformer_norm_input=True, graph_pooling="cls", num_encoder_layers_masked=0,
                           transformer_prenorm=False, pos_encoder=False, max_seq_len=3, compute_dtype=torch.float32,
                           token_layout="auto")


def _model(gnn_type="gcn"):
    from graphtrans_amd.encoders import ASTNodeEncoder
    from graphtrans_amd.models.gnn_transformer import GNNTransformer
    torch.manual_seed(0)
    a = _args()
    a.gnn_type = gnn_type
    m = GNNTransformer(50, ASTNodeEncoder(64, 98, 300, 20), lambda d: torch.nn.Linear(2, d), a).to("cuda:0")
    with torch.no_grad():
        m.gnn_node.virtualnode_embedding.weight.normal_(0, 0.3)
    return m.train()


def _shard(rank):
    from graphtrans_amd import synth
    b = synth.code2_like(B=6 + 3 * rank, seed=10 + rank, num_nodeattributes=300, mean_nodes=30.0).to("cuda:0")
    y = torch.randint(0, 50, (6 + 3 * rank, 5), generator=torch.Generator().manual_seed(rank)).to("cuda:0")
    return b, y


def _grads(model, rank, sync=None):
    from graphtrans_amd import losses
    b, y = _shard(rank)
    for p in model.parameters():
        p.grad = None
    losses.code2_loss(model(b), y).backward()
    if sync is not None:
        sync.finish()
    return [p.grad.detach().float().cpu().clone() for p in model.parameters()]


def _worker(rank, world, port, out):
    from graphtrans_amd import engine
    from graphtrans_amd.dist import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _model()
    sync = GradSync(model.parameters(), world_size=world).attach(model)
    b, _ = _shard(rank)
    assert engine.eligible(model, b, None)
    for _ in range(2):   # the second step reuses the persistent flat buffer
        g = _grads(model, rank, sync)
    out[rank] = g
    dist.destroy_process_group()


def test_two_rank_fused_backward_averages_the_shard_gradients():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    model = _model()
    ref = [(a + b) / 2 for a, b in zip(_grads(model, 0), _grads(model, 1))]
    for r in (0, 1):
        for got, want, (name, _) in zip(out[r], ref, model.named_parameters()):
            scale = max(1.0, float(want.abs().max()))
            assert torch.allclose(got / scale, want / scale, rtol=1e-5, atol=1e-6), (r, name, (got - want).abs().max())
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)   # both ranks hold the same reduced gradients


# ---- synchronised BatchNorm: two shards of ONE global batch reproduce the single-device gradients -------------------------
def _global_batch(ids):
    from graphtrans_amd import synth
    from graphtrans_amd.data import GraphStore
    raw = synth.code2_raw(B=12, seed=21, mean_nodes=30.0, max_nodes=80, num_nodeattributes=300, num_vocab=50)
    return GraphStore(raw).collate(ids)


def _sync_worker(rank, world, port, out, gnn_type, fused=True):
    import numpy as np

    from graphtrans_amd import engine, losses
    from graphtrans_amd.dist import GradSync
    from graphtrans_amd.modules.norm import convert_sync_batchnorm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = convert_sync_batchnorm(_model(gnn_type))
    sync = GradSync(model.parameters(), world_size=world).attach(model)
    b = _global_batch(np.arange(6 * rank, 6 * rank + 6))
    model.fused = fused
    # synchronised statistics run on the FUSED path too (round 3): the library's BatchNorm calls exchange them through
    # dist.BnSyncHook (gt_bn_sync_set), installed around the fused forward / backward
    assert engine.eligible(model, b, None) == fused
    for p in model.parameters():
        p.grad = None
    losses.code2_loss(model(b), b.y_arr).backward()
    sync.finish()
    if fused:
        hook = engine.state(model).get("bn_hook")
        assert hook is not None and hook.calls > 0, "the fused path did not exchange any BatchNorm statistics"
    out[rank] = ([p.grad.detach().float().cpu().clone() for p in model.parameters()],
                 {k: v.detach().float().cpu().clone() for k, v in model.named_buffers() if "running" in k})
    dist.destroy_process_group()


@pytest.mark.parametrize("fused", [True, False], ids=["engine", "modules"])
@pytest.mark.parametrize("gnn_type", ["gcn", "gin"])
def test_sync_batchnorm_shards_equal_the_single_device_batch(gnn_type, fused):
    """with graphs sharded over ranks, per-rank BatchNorm statistics are a different model from the
    reference's single-device batch; convert_sync_batchnorm restores it.  Two ranks x 6 graphs with synchronised
    statistics + gradient averaging == one process on the 12-graph batch (gradients AND running statistics)."""
    import numpy as np

    from graphtrans_amd import losses
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_sync_worker, args=(2, port, out, gnn_type, fused), nprocs=2, join=True)
    model = _model(gnn_type)
    b = _global_batch(np.arange(12))
    losses.code2_loss(model(b), b.y_arr).backward()
    ref = [p.grad.detach().float().cpu() for p in model.parameters()]
    ref_buf = {k: v.detach().float().cpu() for k, v in model.named_buffers() if "running" in k}
    for r in (0, 1):
        grads, bufs = out[r]
        for got, want, (name, _) in zip(grads, ref, model.named_parameters()):
            scale = max(1e-3, float(want.abs().max()))
            assert float((got - want).abs().max()) <= 2e-4 * scale, (r, name, float((got - want).abs().max()), scale)
        for k, want in ref_buf.items():
            assert torch.allclose(bufs[k], want, rtol=1e-4, atol=1e-5), (r, k)
