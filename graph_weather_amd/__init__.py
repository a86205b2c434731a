"""graph_weather_amd - MI355X (gfx950) native implementation of the GraphWeatherForecaster hot path.

Mirrors the import surface of the reference for that path (``graph_weather/__init__.py:9``,
``graph_weather/models/__init__.py:13-15``): ``GraphWeatherForecaster``, ``Encoder``, ``Processor``, ``Decoder``,
``GraphProcessor``, ``MLP``, ``NormalizedMSELoss``.
"""
from .analysis import AssimilatorEncoder, GraphWeatherAssimilator, GraphWeatherAssimilatorConfig  # noqa: F401
from .forecast import GraphWeatherForecaster, GraphWeatherForecasterConfig  # noqa: F401
from .layers import (  # noqa: F401
    MLP,
    AssimilatorDecoder,
    Decoder,
    EdgeProcessor,
    Encoder,
    GraphProcessor,
    NodeProcessor,
    Processor,
    build_graph_processor_block,
    set_compute_dtype,
    set_deterministic,
)
from .ops import BF16X3  # noqa: F401  (set_compute_dtype(model, BF16X3): split-operand products, csrc/gw_split.hip)
from .graphcast import GraphCast, GraphCastConfig  # noqa: F401
from .losses import NormalizedMSELoss  # noqa: F401
from .regional import (  # noqa: F401
    BoundaryNudgingLayer,
    DynamicGraphBuilder,
    RegionalForecaster,
    RegionalForecasterConfig,
)
from .graphed import ForwardGraph  # noqa: F401  (the inference forward as one HIP graph)
from .rollout import rollout  # noqa: F401
from .optim import AdamW  # noqa: F401

__version__ = "0.1.0"
