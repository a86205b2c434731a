"""graph_weather/models/regional_forecast.py of the reference."""
from graph_weather_amd.regional import BoundaryNudgingLayer, RegionalForecaster, RegionalForecasterConfig  # noqa: F401
