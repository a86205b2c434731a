"""graph_weather/models/layers/graph_net_block.py of the reference."""
from graph_weather_amd.layers import (  # noqa: F401
    MLP,
    EdgeProcessor,
    GraphProcessor,
    NodeProcessor,
    build_graph_processor_block,
)
