"""graph_weather/models/analysis.py of the reference."""
from graph_weather_amd.analysis import GraphWeatherAssimilator, GraphWeatherAssimilatorConfig  # noqa: F401
