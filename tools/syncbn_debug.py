import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
from test_hip_dp import _model, _global_batch

def worker(rank, world, port, out, fused, nogs):
    from graphtrans_amd import engine, losses
    from graphtrans_amd.dist import GradSync
    from graphtrans_amd.modules.norm import convert_sync_batchnorm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = convert_sync_batchnorm(_model("gcn"))
    model.fused = fused
    sync = None
    if nogs == 0: sync = GradSync(model.parameters(), world_size=world).attach(model)
    if nogs == 2: _unused = GradSync(model.parameters(), world_size=world)          # constructed, not attached
    if nogs == 3: _pad = torch.zeros(9_000_000, device="cuda:0")                     # only shifts the allocator
    b = _global_batch(np.arange(6 * rank, 6 * rank + 6))
    for p in model.parameters(): p.grad = None
    o = model(b)
    loss = losses.code2_loss(o, b.y_arr)
    loss.backward()
    if sync is not None: sync.finish()
    torch.cuda.synchronize()
    hk = engine.state(model).get("bn_hook")
    out[rank] = (float(loss), [p.grad.detach().float().cpu().clone() for p in model.parameters()], {k: v.detach().float().cpu().clone() for k, v in model.named_buffers() if "running" in k})
    dist.destroy_process_group()

if __name__ == "__main__":
    from graphtrans_amd import losses
    model = _model("gcn")
    b = _global_batch(np.arange(12))
    l = losses.code2_loss(model(b), b.y_arr); l.backward()
    ref = [p.grad.detach().float().cpu() for p in model.parameters()]
    refb = {k: v.detach().float().cpu() for k, v in model.named_buffers() if "running" in k}
    names = [n for n, _ in model.named_parameters()]
    for fused, nogs in ((True, 1),) * 4 + ((True, 0),) * 4:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out = mp.Manager().dict()
        mp.spawn(worker, args=(2, port, out, fused, nogs), nprocs=2, join=True)
        print(f"== fused={fused} no_gradsync={nogs}: global loss {float(l):.6f}  rank losses {out[0][0]:.6f} {out[1][0]:.6f} avg {(out[0][0]+out[1][0])/2:.6f}")
        g = [(a + c) / 2 for a, c in zip(out[0][1], out[1][1])] if nogs else out[0][1]
        worst = sorted(((float((x - y).abs().max()) / max(1e-3, float(y.abs().max())), n) for x, y, n in zip(g, ref, names)), reverse=True)[:6]
        print("   worst grads:", [(n, f"{e:.2e}") for e, n in worst])
        wb = sorted(((float((out[0][2][k] - v).abs().max()), k) for k, v in refb.items()), reverse=True)[:3]
        print("   worst running stats:", [(k, f"{e:.2e}") for e, k in wb])
