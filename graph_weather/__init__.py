"""``graph_weather`` import paths served by ``graph_weather_amd`` (drop-in alias package).

A script written against the reference (``from graph_weather import GraphWeatherForecaster``,
``from graph_weather.models.losses import NormalizedMSELoss`` ...) runs on the MI355X implementation when this repository
precedes the reference on ``sys.path`` - no edit of the import lines.  Only the hot-path surface of
``graph_weather/__init__.py:8-9`` and ``graph_weather/models/__init__.py:13-15`` exists here; the reference's data
loaders and other model families are out of scope and raise ``ImportError`` as absent modules do.
"""
from graph_weather_amd import GraphWeatherAssimilator, GraphWeatherForecaster  # noqa: F401

__all__ = ["GraphWeatherAssimilator", "GraphWeatherForecaster"]
