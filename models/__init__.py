"""Top-level alias of graphtrans_amd.models: `from models import get_model_and_parser` (main.py:21 of the reference) and
`from models.gnn_transformer import GNNTransformer` resolve to the MI355X-native implementation when this repository
root is on sys.path, with no edit to the reference's driver."""
import sys

import graphtrans_amd.models as _m
from graphtrans_amd.models import *  # noqa: F401,F403
from graphtrans_amd.models import MODELS, get_model_and_parser  # noqa: F401

for _name in ("base_model", "gnn_transformer", "pna_transformer"):
    _mod = __import__("graphtrans_amd.models." + _name, fromlist=["x"])
    sys.modules[__name__ + "." + _name] = _mod
    setattr(sys.modules[__name__], _name, _mod)
