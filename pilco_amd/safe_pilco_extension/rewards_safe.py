"""safe_pilco_extension/rewards_safe.py of the reference: the risk terms and the combined objective."""
from ..safe import ObjectiveFunction, RiskOfCollision, SingleConstraint  # noqa: F401

__all__ = ["RiskOfCollision", "SingleConstraint", "ObjectiveFunction"]
