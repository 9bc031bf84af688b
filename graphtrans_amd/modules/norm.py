"""BatchNorm1d with the torch parameter/buffer layout (weight, bias, running_mean, running_var,
num_batches_tracked — the reference's state_dict keys) computing through gt_batchnorm_fwd/bwd,
with an optional fused ReLU (modules/gnn_module.py:84-90,161-170,204-209; modules/conv.py:18-20)."""
import torch

from .. import ops


class BatchNorm1d(torch.nn.BatchNorm1d):
    def forward(self, x, relu=False, dropout_p=0.0, seed=0):
        if x.dim() != 2:
            raise ValueError("expected 2D input (got {}D input)".format(x.dim()))
        if not (self.affine and self.track_running_stats) or self.momentum is None:
            raise NotImplementedError("graphtrans_amd BatchNorm1d supports the reference's default configuration only")
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var,
                              self.num_batches_tracked if self.training else None, self.momentum, self.eps, self.training,
                              relu, dropout_p, seed)


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
