"""graph_weather/models/layers/dynamic_graph_builder.py of the reference."""
from graph_weather_amd.regional import DynamicGraphBuilder  # noqa: F401
