"""graph_weather/models/__init__.py:13-22 of the reference (hot-path classes)."""
from graph_weather_amd import (  # noqa: F401
    AssimilatorDecoder,
    AssimilatorEncoder,
    Decoder,
    Encoder,
    GraphCast,
    GraphCastConfig,
    Processor,
    RegionalForecaster,
    RegionalForecasterConfig,
)
