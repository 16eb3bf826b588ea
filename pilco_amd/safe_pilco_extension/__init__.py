"""The module paths of the reference's Safe-PILCO extension (/root/reference/safe_pilco_extension/): a script written
against ``safe_pilco_extension.safe_pilco`` / ``safe_pilco_extension.rewards_safe`` only changes the package prefix.
The classes live in pilco_amd/safe.py."""
from . import rewards_safe, safe_pilco  # noqa: F401
