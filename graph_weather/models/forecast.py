"""graph_weather/models/forecast.py of the reference."""
from graph_weather_amd.forecast import GraphWeatherForecaster, GraphWeatherForecasterConfig  # noqa: F401
