"""graph_weather/models/layers/assimilator_encoder.py of the reference."""
from graph_weather_amd.analysis import AssimilatorEncoder  # noqa: F401
