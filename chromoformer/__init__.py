"""Drop-in alias: `import chromoformer` / `python -m chromoformer.train` resolve to the MI355X
implementation in chromoformer_amd (same three public names as the reference package)."""
from chromoformer_amd import ChromoformerClassifier, ChromoformerRegressor, ChromoformerDataset  # noqa: F401
