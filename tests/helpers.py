"""Shared helpers for the GPU parity tests: build graphtrans_amd modules for a golden fixture."""
import torch
import torch.nn as nn

from graphtrans_amd.encoders import ASTNodeEncoder, AtomEncoder, BondEncoder


def zero_edge_encoder_cls(_):
    def zero(_):
        return 0

    return zero


def edge_cls(kind):
    return {"linear": lambda d: nn.Linear(2, d), "bond": lambda d: BondEncoder(emb_dim=d),
            "none": zero_edge_encoder_cls}[kind]


def node_encoder(feat, D):
    if feat == "code2":
        return ASTNodeEncoder(D, num_nodetypes=11, num_nodeattributes=13, max_depth=20)
    if feat == "mol":
        return AtomEncoder(D)
    if feat == "tud":
        return nn.Linear(6, D)
    raise ValueError(feat)


def load_sd(module, sd):
    """strict load of a golden (reference) state_dict: the key sets must match exactly, except the
    deterministic `pos_encoder.pe` table which fixtures do not store."""
    own = module.state_dict()
    missing = [k for k in own if k not in sd and not k.endswith("pos_encoder.pe")]
    unexpected = [k for k in sd if k not in own]
    assert not missing and not unexpected, f"state_dict mismatch: missing {missing}, unexpected {unexpected}"
    module.load_state_dict({k: v for k, v in sd.items()}, strict=False)
    return module


def grads_of(module):
    return {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in module.named_parameters()}
