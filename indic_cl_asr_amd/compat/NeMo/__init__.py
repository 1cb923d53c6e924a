"""Alias of the reference repo's vendored `NeMo/` directory name (see indic_cl_asr_amd/compat/__init__.py)."""
__ia_alias__ = True
