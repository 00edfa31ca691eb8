"""`python -m chromoformer.train ...` -> chromoformer_amd.train (same CLI, config and .pt layout)."""
import sys

from chromoformer_amd.train import main

if __name__ == "__main__":
    sys.exit(main())
