"""BatchNorm1d with the torch parameter/buffer layout (weight, bias, running_mean, running_var,
num_batches_tracked — the reference's state_dict keys) computing through gt_batchnorm_fwd/bwd,
with an optional fused ReLU (modules/gnn_module.py:84-90,161-170,204-209; modules/conv.py:18-20)."""
import torch

from .. import ops


class BatchNorm1d(torch.nn.BatchNorm1d):
    sync = False        # set by convert_sync_batchnorm: statistics over the rows of every data-parallel rank
    sync_group = None

    def forward(self, x, relu=False, dropout_p=0.0, seed=0):
        if x.dim() != 2:
            raise ValueError("expected 2D input (got {}D input)".format(x.dim()))
        if not (self.affine and self.track_running_stats) or self.momentum is None:
            raise NotImplementedError("graphtrans_amd BatchNorm1d supports the reference's default configuration only")
        if self.sync and self.training and sync_active(self.sync_group):
            return ops.sync_batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked,
                                       self.momentum, self.eps, relu, dropout_p, seed, self.sync_group)
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var,
                              self.num_batches_tracked if self.training else None, self.momentum, self.eps, self.training,
                              relu, dropout_p, seed)


def sync_active(group=None):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def convert_sync_batchnorm(module, group=None):
    """Turn every BatchNorm1d of `module` into a synchronised one (the node-level BatchNorms after each conv, GINConv's
    inner one and the two of every virtual-node MLP: modules/gnn_module.py:84,204,164,167; modules/conv.py:19): with
    graphs sharded over ranks the statistics then cover the GLOBAL batch, i.e. the model is the single-device b256 model
    of the reference.  The composite / fused paths compute BatchNorm inside one C call, so a synchronised model runs
    module by module (engine.eligible and the layer composites decline it).  Returns `module`."""
    for m in module.modules():
        if isinstance(m, BatchNorm1d):
            m.sync, m.sync_group = True, group
    try:
        from .. import engine
        engine.invalidate(module)
    except Exception:
        pass
    return module


def any_sync(*bns):
    return any(getattr(b, "sync", False) and sync_active(getattr(b, "sync_group", None)) for b in bns)


def mlp_bn_relu(seq, x, dropout_p=0.0, seed=0):
    """Run the reference's nn.Sequential MLPs (Linear, BatchNorm1d, ReLU[, Linear, BatchNorm1d, ReLU])
    with BN+ReLU fused; module indices (and therefore state_dict keys) are unchanged.  dropout_p / seed:
    the F.dropout the caller applies to the MLP output, fused behind the LAST BatchNorm+ReLU."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, BatchNorm1d) and i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.ReLU):
            last = i + 2 >= len(mods)
            x = m(x, relu=True, dropout_p=dropout_p if last else 0.0, seed=seed)
            i += 2
        elif isinstance(m, torch.nn.Linear):
            x = ops.linear_module(m, x)
            i += 1
        else:
            x = m(x)
            i += 1
    return x
