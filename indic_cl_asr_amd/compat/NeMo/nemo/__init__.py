"""Alias package: names only (indic_cl_asr_amd/compat/__init__.py)."""
