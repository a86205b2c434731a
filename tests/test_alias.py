"""The ``graph_weather`` alias package (the reference's import paths) and strict state_dict
exchange with the reference's own classes.

CPU: the alias exposes the hot-path names of ``graph_weather/__init__.py:8-9`` and ``graph_weather/models/__init__.py:13-15``;
the ``state_dict()`` of the reference's OWN ``GraphWeatherForecaster`` / ``GraphCast`` (executed from /root/reference through
``oracle/refload.py``, build container only) loads with ``strict=True`` and has exactly the key -> shape table the GPU test
below rebuilds without the reference.
GPU: a state_dict keyed and shaped like the reference's (built from that table, not from our module) loads ``strict=True``
into the alias' model and reproduces the golden output of the reference's own class.
"""
import contextlib
import os
import sys

import numpy as np
import pytest
import torch

from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons, seeded_features
from oracle.refload import reference_available

from .helpers import forecaster_param_shapes, make_params


@contextlib.contextmanager
def alias_modules():
    """``oracle/refload.py`` registers shells named ``graph_weather*`` in ``sys.modules`` when the live reference has been
    loaded in this process; put them aside so that ``import graph_weather`` resolves to the alias package of this repo."""
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "graph_weather" or k.startswith("graph_weather.")}
    try:
        yield
    finally:
        for k in [k for k in sys.modules if k == "graph_weather" or k.startswith("graph_weather.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_alias_exposes_the_reference_import_paths():
    import graph_weather_amd as gw

    with alias_modules():
        import graph_weather
        from graph_weather import GraphWeatherAssimilator, GraphWeatherForecaster
        from graph_weather.models import (AssimilatorDecoder, AssimilatorEncoder, Decoder, Encoder, GraphCast, GraphCastConfig,
                                          Processor, RegionalForecaster, RegionalForecasterConfig)
        from graph_weather.models.analysis import GraphWeatherAssimilator as A2
        from graph_weather.models.forecast import GraphWeatherForecaster as F2, GraphWeatherForecasterConfig
        from graph_weather.models.graphcast import GraphCast as G2
        from graph_weather.models.graphcast.model import GraphCastConfig as GC2
        from graph_weather.models.layers.assimilator_decoder import AssimilatorDecoder as AD2
        from graph_weather.models.layers.decoder import Decoder as D2
        from graph_weather.models.layers.dynamic_graph_builder import DynamicGraphBuilder
        from graph_weather.models.layers.encoder import Encoder as E2
        from graph_weather.models.layers.graph_net_block import MLP, EdgeProcessor, GraphProcessor, NodeProcessor
        from graph_weather.models.layers.processor import Processor as P2
        from graph_weather.models.losses import NormalizedMSELoss
        from graph_weather.models.regional_forecast import BoundaryNudgingLayer
        from graph_weather.utils import validate_lat_lons

        assert os.path.dirname(graph_weather.__file__).startswith(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert GraphWeatherForecaster is gw.GraphWeatherForecaster is F2 and GraphWeatherAssimilator is gw.GraphWeatherAssimilator is A2
    assert (Encoder, Processor, Decoder) == (gw.Encoder, gw.Processor, gw.Decoder) == (E2, P2, D2)
    assert (MLP, GraphProcessor, EdgeProcessor, NodeProcessor) == (gw.MLP, gw.GraphProcessor, gw.EdgeProcessor, gw.NodeProcessor)
    assert NormalizedMSELoss is gw.NormalizedMSELoss and GraphCast is gw.GraphCast is G2 and GraphCastConfig is GC2
    assert AssimilatorDecoder is AD2 and AssimilatorEncoder is gw.AssimilatorEncoder
    assert RegionalForecaster is gw.RegionalForecaster and RegionalForecasterConfig is gw.RegionalForecasterConfig
    assert DynamicGraphBuilder is gw.DynamicGraphBuilder and BoundaryNudgingLayer is gw.BoundaryNudgingLayer
    assert GraphWeatherForecasterConfig is gw.GraphWeatherForecasterConfig and callable(validate_lat_lons)


@pytest.mark.reference
@pytest.mark.skipif(not reference_available(), reason="/root/reference not present (GPU box)")
def test_reference_state_dicts_load_strict():
    """state_dict() of the reference's own classes -> load_state_dict(strict=True) of ours, and back."""
    import graph_weather_amd as gw
    from oracle.refload import load_reference

    ns = load_reference()
    lat_lons = regular_lat_lons(30.0)
    ref = ns.GraphWeatherForecaster(lat_lons)
    deterministic_fill_(ref, seed=5)
    sd = ref.state_dict()
    ours = gw.GraphWeatherForecaster(lat_lons)
    res = ours.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in ours.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # the key -> shape table the GPU test rebuilds without the reference is exactly the reference's
    table = forecaster_param_shapes(ours.encoder.num_h3)
    assert {k: tuple(v.shape) for k, v in sd.items()} == table
    ref.load_state_dict(ours.state_dict(), strict=True)  # and the other way round
    # GraphCast wrapper (graphcast/model.py): same sub-module names
    refc = ns.GraphCast(lat_lons)
    oursc = gw.GraphCast(lat_lons)
    res = oursc.load_state_dict(refc.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    refc.load_state_dict(oursc.state_dict(), strict=True)


def test_topology_hash_and_check():
    from graph_weather_amd.graphs import build_forecast_graphs, check_topology

    g1 = build_forecast_graphs(regular_lat_lons(30.0), 2)
    g2 = build_forecast_graphs(regular_lat_lons(30.0), 2)
    g3 = build_forecast_graphs(regular_lat_lons(20.0), 2)
    assert g1.provider == "builtin"  # h3 is not installed in this image
    assert g1.topology_hash() == g2.topology_hash() != g3.topology_hash()

    class M:  # stands for a model: .encoder.graphs
        pass

    m = M()
    m.encoder = M()
    m.encoder.graphs = g1
    assert check_topology(m, g1.topology_hash())
    with pytest.warns(UserWarning):
        assert not check_topology(m, g3.topology_hash())
    with pytest.raises(RuntimeError):
        check_topology(m, "0" * 16, strict=True)


def test_output_dim_must_equal_feature_dim():
    """decoder.py:93 adds features[..., :feature_dim] to a [.., output_dim] tensor - a shape error in the reference."""
    import graph_weather_amd as gw

    with pytest.raises(RuntimeError):
        gw.GraphWeatherForecaster(regular_lat_lons(30.0), feature_dim=78, output_dim=40)


@pytest.mark.gpu
def test_reference_keyed_state_dict_reproduces_reference_output(golden_dir):
    """A state_dict with the reference's keys and shapes (built from the SURVEY appendix-B table, values from the per-key seeded
    streams the golden generator used on the reference's own model) loads strict=True through the alias import path and
    gives the output the reference's own GraphWeatherForecaster produced (tests/golden/forecaster_10deg_b2.npz)."""
    with alias_modules():
        from graph_weather import GraphWeatherForecaster

        lat_lons = regular_lat_lons(10.0)
        model = GraphWeatherForecaster(lat_lons)
    sd = make_params(forecaster_param_shapes(model.encoder.num_h3), seed=0)
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to("cuda:0").eval()
    g = np.load(os.path.join(golden_dir, "forecaster_10deg_b2.npz"))
    feats = seeded_features(2, len(lat_lons), 102, seed=42)
    with torch.no_grad():
        y = model(feats.to("cuda:0")).cpu()
    ref = torch.from_numpy(g["out"])
    d_ref = ref - feats[..., :78]
    err = ((y - feats[..., :78]) - d_ref).abs().max().item()
    assert err <= 2e-4 * d_ref.abs().max().item()
    # inference_mode works too (cache keys do not read ._version of inference tensors)
    with torch.inference_mode():
        y2 = model(feats.to("cuda:0")).cpu()
        y3 = model(feats.to("cuda:0")).cpu()
    assert (y2 - y).abs().max().item() <= 1e-5 and torch.equal(y2, y3) or (y3 - y2).abs().max().item() <= 1e-5
