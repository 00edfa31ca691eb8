"""`python -m chromoformer.predict` -> chromoformer_amd.predict (inference entrypoint of the drop-in package)."""
from chromoformer_amd.predict import main, modernise_keys, predict  # noqa: F401

if __name__ == "__main__":
    raise SystemExit(main())
