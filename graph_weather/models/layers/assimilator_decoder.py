"""graph_weather/models/layers/assimilator_decoder.py of the reference."""
from graph_weather_amd.layers import AssimilatorDecoder  # noqa: F401
