"""graph_weather/models/layers/processor.py of the reference."""
from graph_weather_amd.layers import Processor  # noqa: F401
