"""Top-level alias of graphtrans_amd.modules: `from modules.gnn_module import GNNNodeEmbedding`,
`from modules.transformer_encoder import TransformerNodeEncoder`, `from modules.utils import pad_batch`, ...
(models/gnn_transformer.py:8-11 of the reference) resolve to the MI355X-native implementation."""
import sys

for _name in ("conv", "gnn_module", "transformer_encoder", "masked_transformer_encoder", "utils", "norm", "pna", "pna.pna_module"):
    _mod = __import__("graphtrans_amd.modules." + _name, fromlist=["x"])
    sys.modules[__name__ + "." + _name] = _mod
    if "." not in _name:
        setattr(sys.modules[__name__], _name, _mod)
