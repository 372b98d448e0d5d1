"""Importable alias of the `3d-beats_amd` package (whose directory name is not an identifier)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("3d-beats_amd")
