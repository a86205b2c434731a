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
    plain = {k: v.clone() for k, v in a.state_dict().items()}
    with pytest.warns(UserWarning, match="no mesh-topology record"):
        b.load_state_dict(plain)
    # sub-modules and the GraphCast wrapper
    assert "gw_topology" not in a.encoder.state_dict()._metadata.get("", {})
    gc = gw.GraphCast(lat_lons)
    assert gc.state_dict()._metadata[""]["gw_topology"] == gc.encoder.graphs.topology_hash()
