"""graph_weather/models/graphcast/model.py of the reference."""
from graph_weather_amd.graphcast import GraphCast, GraphCastConfig  # noqa: F401
