"""reference network/switchwhiten.py surface."""
from .sync_switchwhiten import SwitchWhiten2d  # noqa: F401
