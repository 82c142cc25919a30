"""mmearth-train_amd — MI355X-native MP-MAE pretraining hot path."""
from . import MODALITIES, config, synth  # noqa: F401

__all__ = ["MODALITIES", "config", "synth"]
