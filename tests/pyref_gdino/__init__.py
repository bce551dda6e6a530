"""Test infrastructure: the Python-sequenced GroundingDINO (generic ovm_g_* device ops driven from Python, round 1) used as an
independent cross-check of the C++ engine. Never imported by the product package."""
