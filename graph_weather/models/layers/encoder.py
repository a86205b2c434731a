"""graph_weather/models/layers/encoder.py of the reference."""
from graph_weather_amd.layers import Encoder  # noqa: F401
