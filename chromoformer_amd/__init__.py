"""chromoformer_amd -- the Chromoformer training hot path on MI355X (gfx950).

Same three public names as the reference package (chromoformer/__init__.py:1-2)."""
from .net import Chromoformer, ChromoformerClassifier, ChromoformerRegressor  # noqa: F401

try:  # the dataset needs pandas; keep the model importable without it
    from .data import ChromoformerDataset  # noqa: F401
except ImportError:  # pragma: no cover
    pass
