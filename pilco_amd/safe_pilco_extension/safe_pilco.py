"""safe_pilco_extension/safe_pilco.py of the reference: SafePILCO."""
from ..safe import SafePILCO  # noqa: F401

__all__ = ["SafePILCO"]
