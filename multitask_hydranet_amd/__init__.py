"""MI355X-native HydraNet forward/backward hot path (drop-in for FlowEternal/multitask-hydranet's model/model.py surface)."""
from .model import HydraNet  # noqa: F401
