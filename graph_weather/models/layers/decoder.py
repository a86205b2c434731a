"""graph_weather/models/layers/decoder.py of the reference."""
from graph_weather_amd.layers import Decoder  # noqa: F401
