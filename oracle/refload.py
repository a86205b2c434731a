"""TEST INFRASTRUCTURE ONLY - never imported by the product path.

Loads the *real* reference source files for the hot path from ``/root/reference`` (read only,
this container only) by file path, with the four third-party packages that are not installable
here replaced by minimal stand-ins of their *public, documented* behaviour:

* ``h3``               -> ``graph_weather_amd.mesh.H3Like`` (same call surface, own mesh; h3 4.3.1
                          is a C library outside the reference tree - topology parity unpinned)
* ``torch_geometric``  -> ``nn.MetaLayer`` (``row, col = edge_index; e = edge_model(x[row], x[col],
                          e, u, batch); x = node_model(x, edge_index, e, u, batch); return x, e, u``)
                          and ``data.Data`` (attribute bag with ``.to(device)``)
* ``torch_scatter``    -> ``scatter_sum(src, index, dim, dim_size)`` ==
                          ``zeros(dim_size, F).scatter_add_(0, index[:, None].expand_as(src), src)``
* ``torch_harmonics``  -> empty module (only imported at the top of ``models/losses.py:6``; the
                          NormalizedMSELoss class does not use it)

It is used by ``oracle/gen_golden.py`` to produce ``tests/golden/*.npz`` from the reference's own
code, and by ``tests/test_oracle_vs_reference.py`` when ``/root/reference`` exists.  Nothing from
the reference is copied into this repository.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("GW_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "graph_weather", "models", "forecast.py"))


def _stub_torch_scatter():
    mod = types.ModuleType("torch_scatter")

    def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
        assert dim == 0 and src.dim() == 2
        if dim_size is None:
            dim_size = int(index.max()) + 1
        res = torch.zeros((dim_size, src.shape[1]), dtype=src.dtype, device=src.device)
        return res.scatter_add_(0, index.view(-1, 1).expand_as(src), src)

    mod.scatter_sum = scatter_sum
    return mod


def _stub_torch_geometric():
    pkg = types.ModuleType("torch_geometric")
    nn_mod = types.ModuleType("torch_geometric.nn")
    data_mod = types.ModuleType("torch_geometric.data")

    class MetaLayer(torch.nn.Module):
        def __init__(self, edge_model=None, node_model=None, global_model=None):
            super().__init__()
            self.edge_model = edge_model
            self.node_model = node_model
            self.global_model = global_model

        def forward(self, x, edge_index, edge_attr=None, u=None, batch=None):
            row = edge_index[0]
            col = edge_index[1]
            if self.edge_model is not None:
                edge_attr = self.edge_model(x[row], x[col], edge_attr, u, batch if batch is None else batch[row])
            if self.node_model is not None:
                x = self.node_model(x, edge_index, edge_attr, u, batch)
            if self.global_model is not None:
                u = self.global_model(x, edge_index, edge_attr, u, batch)
            return x, edge_attr, u

    class Data:
        def __init__(self, **kwargs):
            for k, v in kwargs.items():
                setattr(self, k, v)

        def to(self, device):
            for k, v in list(self.__dict__.items()):
                if torch.is_tensor(v):
                    setattr(self, k, v.to(device))
            return self

    nn_mod.MetaLayer = MetaLayer
    data_mod.Data = Data
    pkg.nn = nn_mod
    pkg.data = data_mod
    return pkg, nn_mod, data_mod


def _load(name: str, relpath: str):
    path = os.path.join(REF_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_CACHE = {}


def load_reference():
    """Returns a namespace with the reference's own classes (executed from /root/reference)."""
    if "ns" in _CACHE:
        return _CACHE["ns"]
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    from graph_weather_amd.mesh import H3Like

    saved = {k: sys.modules.get(k) for k in ("h3", "torch_geometric", "torch_geometric.nn", "torch_geometric.data",
                                             "torch_scatter", "torch_harmonics", "graph_weather",
                                             "graph_weather.models", "graph_weather.models.layers")}
    h3mod = types.ModuleType("h3")
    prov = H3Like()
    for fn in ("get_num_cells", "get_res0_cells", "uncompact_cells", "latlng_to_cell", "cell_to_latlng", "grid_disk",
               "great_circle_distance"):
        setattr(h3mod, fn, getattr(prov, fn))
    sys.modules["h3"] = h3mod
    pkg, nn_mod, data_mod = _stub_torch_geometric()
    sys.modules["torch_geometric"] = pkg
    sys.modules["torch_geometric.nn"] = nn_mod
    sys.modules["torch_geometric.data"] = data_mod
    sys.modules["torch_scatter"] = _stub_torch_scatter()
    th = types.ModuleType("torch_harmonics")
    th.RealSHT = type("RealSHT", (), {})  # only named in a return annotation (losses.py:132)
    sys.modules["torch_harmonics"] = th
    # bare package shells so that the reference's absolute imports resolve without executing
    # graph_weather/__init__.py (which pulls the data loaders / other model families)
    for name in ("graph_weather", "graph_weather.models", "graph_weather.models.layers"):
        shell = types.ModuleType(name)
        shell.__path__ = []  # mark as package
        sys.modules[name] = shell
    base = "graph_weather/models/"
    gnb = _load("graph_weather.models.layers.graph_net_block", base + "layers/graph_net_block.py")
    _load("graph_weather.models.layers.thermalizer", base + "layers/thermalizer.py")
    enc = _load("graph_weather.models.layers.encoder", base + "layers/encoder.py")
    proc = _load("graph_weather.models.layers.processor", base + "layers/processor.py")
    adec = _load("graph_weather.models.layers.assimilator_decoder", base + "layers/assimilator_decoder.py")
    dec = _load("graph_weather.models.layers.decoder", base + "layers/decoder.py")
    _load("graph_weather.models.layers.constraint_layer", base + "layers/constraint_layer.py")
    models = sys.modules["graph_weather.models"]
    models.Encoder, models.Processor, models.Decoder = enc.Encoder, proc.Processor, dec.Decoder
    aenc = _load("graph_weather.models.layers.assimilator_encoder", base + "layers/assimilator_encoder.py")
    models.AssimilatorEncoder, models.AssimilatorDecoder = aenc.AssimilatorEncoder, adec.AssimilatorDecoder
    ana = _load("graph_weather.models.analysis", base + "analysis.py")
    gc_shell = types.ModuleType("graph_weather.models.graphcast")
    gc_shell.__path__ = []
    sys.modules["graph_weather.models.graphcast"] = gc_shell
    gcast = _load("graph_weather.models.graphcast.model", base + "graphcast/model.py")
    sys.modules["graph_weather.utils"] = _load("graph_weather.utils", "graph_weather/utils.py")
    dgb = _load("graph_weather.models.layers.dynamic_graph_builder", base + "layers/dynamic_graph_builder.py")
    reg = _load("graph_weather.models.regional_forecast", base + "regional_forecast.py")
    fc = _load("graph_weather.models.forecast", base + "forecast.py")
    losses = _load("graph_weather.models.losses", base + "losses.py")
    ns = types.SimpleNamespace(
        MLP=gnb.MLP, GraphProcessor=gnb.GraphProcessor, EdgeProcessor=gnb.EdgeProcessor,
        NodeProcessor=gnb.NodeProcessor, Encoder=enc.Encoder, Processor=proc.Processor, Decoder=dec.Decoder,
        AssimilatorDecoder=adec.AssimilatorDecoder, GraphWeatherForecaster=fc.GraphWeatherForecaster,
        GraphWeatherForecasterConfig=fc.GraphWeatherForecasterConfig, NormalizedMSELoss=losses.NormalizedMSELoss,
        AssimilatorEncoder=aenc.AssimilatorEncoder, GraphWeatherAssimilator=ana.GraphWeatherAssimilator,
        GraphCast=gcast.GraphCast, GraphCastConfig=getattr(gcast, "GraphCastConfig", None),
        DynamicGraphBuilder=dgb.DynamicGraphBuilder, RegionalForecaster=reg.RegionalForecaster,
        RegionalForecasterConfig=reg.RegionalForecasterConfig, BoundaryNudgingLayer=reg.BoundaryNudgingLayer,
    )
    # leave the stubs registered under their names only while the reference modules need them at call
    # time (h3 is used in __init__ of Encoder/Decoder) - they shadow nothing real in this image.
    _CACHE["ns"] = ns
    _CACHE["saved"] = saved
    return ns
