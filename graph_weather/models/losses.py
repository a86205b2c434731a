"""graph_weather/models/losses.py of the reference (the hot path's loss only)."""
from graph_weather_amd.losses import NormalizedMSELoss  # noqa: F401
