"""Structured synthetic probability maps for the oracle's tests and golden-vector generators: one implementation, shared with
bench.py's post-processing leg, lives in cerberus_amd/synth_maps.py (an input generator, not a checker)."""
from cerberus_amd.synth_maps import _sigmoid, blob_maps, gland_maps, nuclei_maps, softmax_nuclei_maps  # noqa: F401
