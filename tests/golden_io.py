"""Loaders for the fixtures under tests/golden/ (data produced by tests/golden/make_golden.py)."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def long_sequence(recipe_name):
    """Rebuilds the input of a minimizer_long.json entry from its recipe (seeded generator)."""
    from sequencealigner_amd import synth
    if recipe_name in ("uniform50k", "uniform50k_k7w8"):
        return bytes(synth.genome(50000, seed=21))
    if recipe_name == "tandem37x700":
        return bytes(synth.genome(37, seed=22)) * 700
    if recipe_name == "mixed_homopolymer":
        return (bytes(synth.genome(8000, seed=23)) + b"A" * 3000 + bytes(synth.genome(5000, seed=24))
                + b"CA" * 2000)
    raise KeyError(recipe_name)
