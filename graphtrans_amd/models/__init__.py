"""Model registry for the hot path (models/__init__.py:9-15): `gnn-transformer` and `pna-transformer`; the
reference's other registry entries are ablation/baseline models outside SURVEY.md §8."""
from .gnn_transformer import GNNTransformer
from .pna_transformer import PNATransformer


def get_model_and_parser(args, parser):
    model_cls = MODELS[args.model_type]
    model_cls.add_args(parser)
    return model_cls


MODELS = {"gnn-transformer": GNNTransformer, "pna-transformer": PNATransformer}
