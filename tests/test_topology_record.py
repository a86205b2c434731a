"""CPU tests of round-5 host logic: the mesh-topology record in checkpoints."""
import io

import pytest
import torch

import graph_weather_amd as gw
from graph_weather_amd.utils import regular_lat_lons


def test_checkpoint_topology_guard_warns_on_load():
    """A state_dict carries the mesh topology it was trained on (in ``_metadata``: no extra key, ``strict=True`` loads and
    reference checkpoints unaffected); loading it into a model whose mesh provider numbers the cells differently warns loudly,
    loading a checkpoint without the record (a reference-trained one) on the built-in mesh warns that topology parity is
    unverified (encoder.py:76-104, assimilator_decoder.py:69-101 number the cells with h3)."""
    import warnings

    lat_lons = regular_lat_lons(30.0)
    a = gw.GraphWeatherForecaster(lat_lons)
    b = gw.GraphWeatherForecaster(lat_lons)
    sd = a.state_dict()
    assert getattr(sd, "_metadata", {}).get("", {}).get("gw_topology") == a.encoder.graphs.topology_hash()
    buf = io.BytesIO()
    torch.save(sd, buf)  # the record survives serialisation
    buf.seek(0)
    sd = torch.load(buf)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        b.load_state_dict(sd)  # same provider, same hash: silent
    # a checkpoint of another topology
    sd2 = a.state_dict()
    sd2._metadata[""]["gw_topology"] = "0" * 16
    with pytest.warns(UserWarning, match="mesh topology"):
        b.load_state_dict(sd2)
    # a reference-trained checkpoint: plain dict of tensors, no record
    from graph_weather_amd import graphs as G

    plain = {k: v.clone() for k, v in a.state_dict().items()}
    G._WARNED_NO_RECORD = False
    with pytest.warns(UserWarning, match="no mesh-topology record"):
        b.load_state_dict(plain)
    with warnings.catch_warnings():  # said once per process: the repo's own dict round trips do not repeat it
        warnings.simplefilter("error")
        b.load_state_dict(plain)
    # sub-modules and the GraphCast wrapper
    assert "gw_topology" not in a.encoder.state_dict()._metadata.get("", {})
    gc = gw.GraphCast(lat_lons)
    assert gc.state_dict()._metadata[""]["gw_topology"] == gc.encoder.graphs.topology_hash()


def test_topology_hash_memo_follows_the_index_tensors():
    """``ForecastGraphs`` is a mutable dataclass - a replaced or in-place edited edge list must not be answered with
    the digest memoised for the old one."""
    import copy

    g = copy.copy(gw.GraphWeatherForecaster(regular_lat_lons(30.0)).encoder.graphs)
    h0 = g.topology_hash()
    assert g.topology_hash() == h0
    g.lat_edge_index = g.lat_edge_index.clone()
    assert g.topology_hash() == h0  # same content at another address: recomputed, equal
    g.lat_edge_index[0, 0] += 1  # in-place edit: the version counter moves
    assert g.topology_hash() != h0
    g.lat_edge_index = g.lat_edge_index.flip(1).contiguous()
    assert g.topology_hash() != h0


def test_hub_round_trip_keeps_the_topology_record(tmp_path):
    """``save_pretrained`` / ``from_pretrained`` (PyTorchModelHubMixin, forecast.py:61) write safetensors, which drop
    ``_metadata``: the record travels as gw_topology.json and is checked after the load - silent on a match, loud on another
    topology, and a directory without the file (a reference checkpoint) gets the no-record warning once."""
    import json
    import os
    import warnings

    from graph_weather_amd import graphs as G

    if not hasattr(gw.GraphWeatherForecaster, "save_pretrained"):
        pytest.skip("huggingface_hub is not importable")
    lat_lons = regular_lat_lons(30.0)
    a = gw.GraphWeatherForecaster(lat_lons)
    a.save_pretrained(str(tmp_path))
    rec = json.load(open(os.path.join(tmp_path, G.TOPOLOGY_FILE)))
    assert rec["gw_topology"] == a.encoder.graphs.topology_hash() and rec["gw_provider"] == a.encoder.graphs.provider
    G._WARNED_NO_RECORD = False
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        b = gw.GraphWeatherForecaster.from_pretrained(str(tmp_path), lat_lons=lat_lons)
    assert all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())
    json.dump({"gw_topology": "0" * 16, "gw_provider": "h3"}, open(os.path.join(tmp_path, G.TOPOLOGY_FILE), "w"))
    with pytest.warns(UserWarning, match="mesh topology"):
        gw.GraphWeatherForecaster.from_pretrained(str(tmp_path), lat_lons=lat_lons)
    os.remove(os.path.join(tmp_path, G.TOPOLOGY_FILE))
    with pytest.warns(UserWarning, match="unverified"):
        gw.GraphWeatherForecaster.from_pretrained(str(tmp_path), lat_lons=lat_lons)
