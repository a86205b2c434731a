"""Init-time construction of the three message-passing graphs and their dst-sorted plans.

What is built (same objects as the reference, reference edge order preserved in ``*_ref`` arrays):

* grid -> mesh bipartite graph, one edge per grid node            (encoder.py:75-104)
* latent mesh graph, disk-1 incl. self loop                        (encoder.py:244-268)
* mesh -> grid bipartite graph, disk-1 of the node's cell          (assimilator_decoder.py:68-103)

with edge attributes ``[sin d, cos d]``, ``d`` = great-circle distance in radians.  Mesh rows of the
encoder/decoder graphs use the *reversed* rank ``M-1-rank(cell)`` while the latent graph uses the
forward rank (encoder.py:80-84 vs :262-263) - reproduced verbatim, never "fixed".

The reference walks Python loops of h3 calls (O(7G) calls); here the built-in mesh is queried in
vectorised numpy so a 0.25 degree grid (1 M nodes) builds in seconds.  With the real ``h3`` package
the literal loops are used instead.

Each graph is also emitted as a **plan** for the HIP kernels: edges stably sorted by destination,
int32 ``src``/``dst``, plus ``perm`` (sorted position -> reference edge id) so that edge tensors can be
exposed in reference order at the API boundary.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import mesh as _mesh

__all__ = ["GraphPlan", "ForecastGraphs", "build_forecast_graphs", "plan_from_coo", "build_observation_graph", "build_latent_graph",
           "check_topology", "topology_hash", "provider_name"]


@dataclass
class GraphPlan:
    """Destination-sorted edge list of one (bipartite) graph.

    ``src`` indexes rows of the source node table (``n_src`` rows per sample), ``dst`` rows of the
    destination table (``n_dst`` rows per sample); both int32, sorted by ``dst`` (stable).
    ``perm[i]`` is the reference edge id stored at sorted position ``i``.
    """

    n_src: int
    n_dst: int
    src: torch.Tensor  # int32 [E]
    dst: torch.Tensor  # int32 [E]
    perm: torch.Tensor  # int64 [E]
    edge_attr: Optional[torch.Tensor]  # float32 [E, 2] in sorted order (None for user graphs)

    @property
    def num_edges(self) -> int:
        return int(self.src.shape[0])

    # ---- index structures of the backward pass (built lazily, on the device of the plan) ----
    def dst_ptr(self) -> torch.Tensor:
        """CSR row pointer over the destination-sorted edges: edges [ptr[n], ptr[n+1]) end in destination row n."""
        if getattr(self, "_dst_ptr", None) is None:
            counts = torch.bincount(self.dst.long(), minlength=self.n_dst)
            self._dst_ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
        return self._dst_ptr

    def src_sorted(self):
        """(perm, ptr): sorted positions grouped by source row - the dual of the x[row] gather (MetaLayer)."""
        if getattr(self, "_src_sorted", None) is None:
            perm = torch.argsort(self.src.long(), stable=True)
            counts = torch.bincount(self.src.long(), minlength=self.n_src)
            ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
            self._src_sorted = (perm.to(torch.int32).contiguous(), ptr)
        return self._src_sorted

    def identity_ptr(self) -> torch.Tensor:
        if getattr(self, "_ident_ptr", None) is None:
            self._ident_ptr = torch.arange(self.num_edges + 1, dtype=torch.int32, device=self.src.device)
        return self._ident_ptr

    def seg_tiles(self, split: bool = False):
        """Segment-aligned tiles of the destination-sorted edge list (include/gw_amd.h: GW_EDGE_SEGMENT_TILES): the edges are
        re-laid into tiles of 64 columns so that no destination's run of edges crosses a tile - runs are packed next-fit in
        order, the rest of a tile is padding (the decoder graph, assimilator_decoder.py:92-103: 7 or 6 edges per grid node, 9
        nodes = 63 columns per tile).  Returns ``SegTiles`` - padded int32 ``src`` / ``dst`` (padding: src 0, dst -1), ``pos[i]``
        = padded position of sorted edge i, ``complete`` = every destination row has an edge.  A destination with more than 64
        edges (the encoder's polar mesh cells, encoder.py:75-104) makes this None - unless ``split``: such a run then starts a
        fresh tile and continues over whole tiles (``SegTiles.split``: its pieces are partial sums that meet in atomics,
        GW_EDGE_SEGMENT_SPLIT)."""
        cache = getattr(self, "_seg_tiles", None)
        if cache is None:
            cache = self._seg_tiles = {}
        if split not in cache:
            dst = self.dst.cpu().numpy().astype(np.int64)
            E = int(dst.size)
            st = None
            if E > 0:
                starts = np.flatnonzero(np.concatenate([[True], dst[1:] != dst[:-1]]))
                ends = np.concatenate([starts[1:], [E]])
                n_runs = int(starts.size)
                longest = int((ends - starts).max())
                has_split = longest > 64
                if has_split and split:  # pieces of 64 (the last one shorter) stand for the long runs in the packing below
                    nparts = (ends - starts + 63) // 64
                    run_of = np.repeat(np.arange(n_runs), nparts)
                    part_ix = np.arange(run_of.size) - np.repeat(np.cumsum(nparts) - nparts, nparts)
                    starts = starts[run_of] + 64 * part_ix
                    ends = np.minimum(starts + 64, ends[run_of])
                if not has_split or split:
                    first, tile_of_seg = 0, np.empty(starts.size, dtype=np.int64)
                    tile_start_edge = []
                    t = 0
                    while first < starts.size:  # next-fit: as many whole runs (pieces) as fit into 64 columns
                        nxt = int(np.searchsorted(ends, starts[first] + 64, side="right"))
                        tile_of_seg[first:nxt] = t
                        tile_start_edge.append(int(starts[first]))
                        first, t = nxt, t + 1
                    tile_start_edge = np.asarray(tile_start_edge, dtype=np.int64)
                    tile_of_edge = np.repeat(tile_of_seg, ends - starts)
                    pos = tile_of_edge * 64 + (np.arange(E, dtype=np.int64) - tile_start_edge[tile_of_edge])
                    n_pad = 64 * t
                    src_p = np.zeros(n_pad, dtype=np.int32)
                    dst_p = np.full(n_pad, -1, dtype=np.int32)
                    src_p[pos] = self.src.cpu().numpy()
                    dst_p[pos] = dst
                    dev = self.src.device
                    st = SegTiles(torch.from_numpy(src_p).to(dev), torch.from_numpy(dst_p).to(dev), torch.from_numpy(pos).to(dev),
                                  n_pad, bool(n_runs == self.n_dst), int(np.bincount(tile_of_seg).max()), bool(has_split))
            cache[split] = st
        return cache[split]

    def to(self, device) -> "GraphPlan":
        return GraphPlan(self.n_src, self.n_dst, self.src.to(device), self.dst.to(device), self.perm.to(device),
                         None if self.edge_attr is None else self.edge_attr.to(device))


@dataclass
class SegTiles:
    src: torch.Tensor  # int32 [n_pad]: source row of each column (0 in padding columns)
    dst: torch.Tensor  # int32 [n_pad]: destination row, -1 in padding columns
    pos: torch.Tensor  # int64 [E]: padded position of destination-sorted edge i
    n_pad: int
    complete: bool  # every destination row has at least one edge
    max_slots: int = 64  # most destinations in one tile (the processor form of the kernels takes up to 16)
    split: bool = False  # some destination's run spans several tiles: its pieces are partial sums (atomics; zero-filled fp32 aggregate)

    def pad_rows(self, rows: torch.Tensor) -> torch.Tensor:
        """A per-edge table [E, w] in destination-sorted order -> [n_pad, w] in padded order (zero rows in padding columns)."""
        out = rows.new_zeros((self.n_pad,) + tuple(rows.shape[1:]))
        out[self.pos] = rows
        return out

    def pad_batched_rows(self, rows: torch.Tensor, batch: int) -> torch.Tensor:
        """Per-sample tables [batch * E, w] -> [batch * n_pad, w]."""
        e = rows.shape[0] // batch
        out = rows.new_zeros((batch, self.n_pad) + tuple(rows.shape[1:]))
        out[:, self.pos] = rows.reshape((batch, e) + tuple(rows.shape[1:]))
        return out.reshape((batch * self.n_pad,) + tuple(rows.shape[1:]))


def plan_from_coo(src: np.ndarray, dst: np.ndarray, n_src: int, n_dst: int,
                  edge_attr: Optional[np.ndarray] = None) -> GraphPlan:
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    if src.size and (src.min() < 0 or src.max() >= n_src or dst.min() < 0 or dst.max() >= n_dst):
        raise ValueError("edge index out of range for node tables (n_src=%d, n_dst=%d)" % (n_src, n_dst))
    if max(n_src, n_dst, src.size) >= 2**31 - 1:
        raise ValueError("graph too large for int32 plans")
    perm = np.argsort(dst, kind="stable")
    ea = None if edge_attr is None else torch.from_numpy(np.ascontiguousarray(edge_attr[perm], dtype=np.float32))
    return GraphPlan(
        int(n_src), int(n_dst),
        torch.from_numpy(src[perm].astype(np.int32)),
        torch.from_numpy(dst[perm].astype(np.int32)),
        torch.from_numpy(perm.astype(np.int64)),
        ea,
    )


@dataclass
class ForecastGraphs:
    num_grid: int
    num_mesh: int
    # reference-order arrays (node ids exactly as the reference numbers them)
    enc_edge_index: torch.Tensor  # int64 [2, G]     targets G + (M-1-rank)
    enc_edge_attr: torch.Tensor  # float32 [G, 2]
    lat_edge_index: torch.Tensor  # int64 [2, E_lat]
    lat_edge_attr: torch.Tensor  # float32 [E_lat, 2]
    dec_edge_index: torch.Tensor  # int64 [2, E_dec] sources M-1-rank, targets M + i
    dec_edge_attr: torch.Tensor  # float32 [E_dec, 2]
    # dst-sorted plans over per-table row ids
    enc_plan: GraphPlan  # src: grid rows,  dst: mesh rows (reversed rank)
    lat_plan: GraphPlan  # src/dst: mesh rows
    dec_plan: GraphPlan  # src: mesh rows (reversed rank), dst: grid rows
    provider: str = "builtin"  # which mesh provider numbered the cells: "h3" (real h3 importable) or "builtin"

    def topology_hash(self) -> str:
        """Digest of the three edge lists: equal hashes = same mesh numbering and topology.  A checkpoint is only meaningful on
        the topology it was trained on (real h3 for reference-trained weights); ``check_topology`` compares."""
        # memoised per content identity of the three index tensors (0.25 degree hashes 116 MB): the dataclass is mutable, so a
        # replaced or in-place edited edge list must not be answered with the digest of the old one
        ident = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (self.enc_edge_index, self.lat_edge_index, self.dec_edge_index))
        memo = self.__dict__.get("_topo_hash")
        if memo is None or memo[0] != ident:
            memo = (ident, topology_hash(self.enc_edge_index, self.lat_edge_index, self.dec_edge_index))
            object.__setattr__(self, "_topo_hash", memo)
        return memo[1]

    def as_oracle_dict(self) -> dict:
        return {
            "num_grid": self.num_grid, "num_mesh": self.num_mesh,
            "enc_edge_index": self.enc_edge_index, "enc_edge_attr": self.enc_edge_attr,
            "lat_edge_index": self.lat_edge_index, "lat_edge_attr": self.lat_edge_attr,
            "dec_edge_index": self.dec_edge_index, "dec_edge_attr": self.dec_edge_attr,
        }


def _sincos(d: np.ndarray) -> np.ndarray:
    return np.stack([np.sin(d), np.cos(d)], axis=1).astype(np.float32)


def _build_vectorised(lat_lons, resolution: int):
    m = _mesh.get_mesh(resolution)
    ll = np.asarray(lat_lons, dtype=np.float64).reshape(-1, 2)
    G, M = ll.shape[0], m.num
    cell = m.locate(ll[:, 0], ll[:, 1])  # rank of the containing cell
    # encoder graph
    d_enc = _mesh.haversine_rads(ll[:, 0], ll[:, 1], m.lat[cell], m.lon[cell])
    enc_src = np.arange(G, dtype=np.int64)
    enc_dst = (M - 1 - cell) + G
    # latent graph
    ptr, idx = m.disk1_csr()
    lat_src = np.repeat(np.arange(M, dtype=np.int64), np.diff(ptr))
    lat_dst = idx
    d_lat = _mesh.haversine_rads(m.lat[lat_src], m.lon[lat_src], m.lat[lat_dst], m.lon[lat_dst])
    # decoder graph
    deg = np.diff(ptr)[cell]
    dec_dst_node = np.repeat(np.arange(G, dtype=np.int64), deg)
    start = np.repeat(ptr[cell], deg)
    within = np.arange(dec_dst_node.size, dtype=np.int64) - np.repeat(np.cumsum(deg) - deg, deg)
    h = idx[start + within]
    d_dec = _mesh.haversine_rads(ll[dec_dst_node, 0], ll[dec_dst_node, 1], m.lat[h], m.lon[h])
    dec_src = M - 1 - h
    dec_dst = dec_dst_node + M
    return (G, M, enc_src, enc_dst, _sincos(d_enc), lat_src, lat_dst, _sincos(d_lat), dec_src, dec_dst, _sincos(d_dec))


def _build_with_h3(lat_lons, resolution: int, h3):  # pragma: no cover - h3 absent in this image
    """Literal loops over a real h3 module (same statements as the reference's constructors)."""
    G = len(lat_lons)
    base = sorted(list(h3.uncompact_cells(h3.get_res0_cells(), resolution)))
    rank = {c: i for i, c in enumerate(base)}
    M = len(base)
    cells = [h3.latlng_to_cell(lat, lon, resolution) for lat, lon in lat_lons]
    enc_src = np.arange(G, dtype=np.int64)
    enc_dst = np.array([M - 1 - rank[c] + G for c in cells], dtype=np.int64)
    d_enc = np.array([h3.great_circle_distance(lat_lons[i], h3.cell_to_latlng(c), unit="rads")
                      for i, c in enumerate(cells)])
    ls, ld, dl = [], [], []
    for c in base:
        for hcell in h3.grid_disk(c, 1):
            ls.append(rank[c]); ld.append(rank[hcell])
            dl.append(h3.great_circle_distance(h3.cell_to_latlng(c), h3.cell_to_latlng(hcell), unit="rads"))
    ds, dd, ddist = [], [], []
    for i, c in enumerate(cells):
        for hcell in h3.grid_disk(c, 1):
            ds.append(M - 1 - rank[hcell]); dd.append(i + M)
            ddist.append(h3.great_circle_distance(lat_lons[i], h3.cell_to_latlng(hcell), unit="rads"))
    return (G, M, enc_src, enc_dst, _sincos(d_enc), np.array(ls), np.array(ld), _sincos(np.array(dl)),
            np.array(ds), np.array(dd), _sincos(np.array(ddist)))


def provider_name(provider=None) -> str:
    provider = provider if provider is not None else _mesh.get_provider()
    return "builtin" if isinstance(provider, _mesh.H3Like) else "h3"


def topology_hash(*index_tensors) -> str:
    import hashlib

    h = hashlib.sha256()
    for t in index_tensors:
        h.update(np.ascontiguousarray(t.cpu().numpy() if torch.is_tensor(t) else t, dtype=np.int64).tobytes())
    return h.hexdigest()[:16]


def check_topology(module, expected_hash: str, strict: bool = False) -> bool:
    """Compare the mesh topology a model runs on with the one a checkpoint was made on (``expected_hash`` = the value of
    ``model.encoder.graphs.topology_hash()`` when it was trained).  Mesh numbering depends on the provider: real h3 when
    importable, the built-in geodesic mesh otherwise (mesh.get_provider) - weights trained on one are meaningless on the
    other although the state_dict loads (the tensors have equal shapes).  Warns (or raises with ``strict``) on mismatch."""
    import warnings

    graphs = getattr(getattr(module, "encoder", module), "graphs", None)
    have = graphs.topology_hash() if graphs is not None else None
    if have == expected_hash:
        return True
    msg = ("graph_weather_amd: mesh topology %s (provider %r) differs from the checkpoint's %s - weights trained on another "
           "mesh numbering (e.g. real h3 vs the built-in mesh) do not transfer" % (have, getattr(graphs, "provider", "?"), expected_hash))
    if strict:
        raise RuntimeError(msg)
    warnings.warn(msg)
    return False


class TopologyRecord:
    """Mixin of the models built on ``ForecastGraphs`` (``self.encoder.graphs``): ``state_dict()`` records which mesh topology
    the weights belong to (``_metadata[""]["gw_topology"]`` / ``["gw_provider"]`` - metadata survives ``torch.save`` and adds no
    key, so reference checkpoints and ``strict=True`` loads are unaffected) and ``load_state_dict()`` compares it with the
    running one.  The reference numbers mesh cells with h3 (encoder.py:76-104, assimilator_decoder.py:69-101); this image has
    no h3 wheel and builds the mesh with ``mesh.H3Like`` - same counts, possibly another numbering - so a checkpoint trained
    elsewhere loads (equal shapes) but may mean something else: that is said out loud here instead of staying silent."""

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        md = getattr(sd, "_metadata", None)
        graphs = getattr(getattr(self, "encoder", None), "graphs", None)
        top_level = not (kwargs.get("prefix") or (len(args) > 1 and args[1]))
        if md is not None and graphs is not None and top_level:
            md.setdefault("", {})
            md[""]["gw_topology"] = graphs.topology_hash()
            md[""]["gw_provider"] = graphs.provider
        return sd

    def load_state_dict(self, state_dict, *args, **kwargs):
        import warnings

        global _WARNED_NO_RECORD
        res = super().load_state_dict(state_dict, *args, **kwargs)
        graphs = getattr(getattr(self, "encoder", None), "graphs", None)
        if graphs is None:
            return res
        md = getattr(state_dict, "_metadata", None) or {}
        expected = md.get("", {}).get("gw_topology") if isinstance(md.get("", {}), dict) else None
        if expected is not None:
            check_topology(self, expected)
        elif graphs.provider != "h3" and not _HUB_LOADING and not _WARNED_NO_RECORD:
            # (once per process: plain {name: tensor} dicts - own round trips through dict comprehensions - carry no record either)
            _WARNED_NO_RECORD = True
            warnings.warn("graph_weather_amd: this checkpoint carries no mesh-topology record (e.g. it was written by the reference, "
                          "whose mesh cells are numbered by h3) and this model runs on the built-in mesh provider (%r, topology %s): "
                          "the tensors load, but parity with the weights' original mesh numbering is unverified - see "
                          "graphs.check_topology (said once per process)" % (graphs.provider, graphs.topology_hash()))
        return res

    # Hub round trip (PyTorchModelHubMixin.save_pretrained / from_pretrained, forecast.py:61): safetensors carries no
    # ``_metadata``, so the record travels as gw_topology.json beside model.safetensors and is checked after the load.
    def _save_pretrained(self, save_directory) -> None:
        import json
        import os

        super()._save_pretrained(save_directory)
        graphs = getattr(getattr(self, "encoder", None), "graphs", None)
        if graphs is not None:
            with open(os.path.join(str(save_directory), TOPOLOGY_FILE), "w") as f:
                json.dump({"gw_topology": graphs.topology_hash(), "gw_provider": graphs.provider}, f)

    @classmethod
    def _from_pretrained(cls, *, model_id, **kwargs):
        import json
        import os
        import warnings

        global _HUB_LOADING, _WARNED_NO_RECORD
        _HUB_LOADING = True  # (the safetensors loader hands load_state_dict a plain dict: the record is checked below instead)
        try:
            model = super()._from_pretrained(model_id=model_id, **kwargs)
        finally:
            _HUB_LOADING = False
        graphs = getattr(getattr(model, "encoder", None), "graphs", None)
        path = os.path.join(str(model_id), TOPOLOGY_FILE)
        if graphs is not None and os.path.isfile(path):
            with open(path) as f:
                check_topology(model, json.load(f).get("gw_topology"))
        elif graphs is not None and graphs.provider != "h3" and not _WARNED_NO_RECORD:
            _WARNED_NO_RECORD = True
            warnings.warn("graph_weather_amd: %s holds no %s (a checkpoint written by the reference, or fetched from the hub without "
                          "it): the tensors load, parity with the weights' original mesh numbering is unverified (provider %r, "
                          "topology %s; said once per process)" % (model_id, TOPOLOGY_FILE, graphs.provider, graphs.topology_hash()))
        return model


TOPOLOGY_FILE = "gw_topology.json"
_HUB_LOADING = False
_WARNED_NO_RECORD = False

_BUILT: list = []  # the last few (key, ForecastGraphs, provider): see build_forecast_graphs


def build_forecast_graphs(lat_lons, resolution: int = 2, provider=None) -> ForecastGraphs:
    """The three graphs of a forecaster.  The last two builds of the process are kept (keyed by a digest of the coordinates,
    the resolution and the provider; the index arrays are read-only by convention): models built again on the same grid - the
    bench's configurations, ranks forked by ``bench.py --gpus N`` after the launcher built them - share one set."""
    import hashlib

    provider = provider if provider is not None else _mesh.get_provider()
    ll = np.ascontiguousarray(np.asarray(lat_lons, dtype=np.float64).reshape(-1, 2))
    key = (hashlib.sha256(ll.tobytes()).hexdigest(), int(resolution), id(provider) if not isinstance(provider, _mesh.H3Like) else "builtin")
    for k, g, _ in _BUILT:
        if k == key:
            return g
    g = _build_forecast_graphs(lat_lons, resolution, provider)
    # (the entry holds the provider object: an id() used as key cannot be recycled while the entry lives.  The plans and their
    # lazily attached tile maps are shared by every model built on the grid and are immutable by convention.)
    _BUILT.append((key, g, provider))
    del _BUILT[:-2]
    return g


def _build_forecast_graphs(lat_lons, resolution: int, provider) -> ForecastGraphs:
    if isinstance(provider, _mesh.H3Like):
        parts = _build_vectorised(lat_lons, resolution)
    else:  # pragma: no cover
        parts = _build_with_h3(lat_lons, resolution, provider)
    G, M, es, ed, ea, ls, ld, la, ds, dd, da = parts
    t = torch.from_numpy
    return ForecastGraphs(
        num_grid=G, num_mesh=M,
        enc_edge_index=t(np.stack([es, ed]).astype(np.int64)), enc_edge_attr=t(ea),
        lat_edge_index=t(np.stack([ls, ld]).astype(np.int64)), lat_edge_attr=t(la),
        dec_edge_index=t(np.stack([ds, dd]).astype(np.int64)), dec_edge_attr=t(da),
        enc_plan=plan_from_coo(es, ed - G, G, M, ea),
        lat_plan=plan_from_coo(ls, ld, M, M, la),
        dec_plan=plan_from_coo(ds, dd - M, M, G, da),
        provider=provider_name(provider),
    )



def _h3_base(h3, resolution: int):  # pragma: no cover - h3 absent in this image
    base = sorted(list(h3.uncompact_cells(h3.get_res0_cells(), resolution)))
    return base, {c: i for i, c in enumerate(base)}


def build_latent_graph(resolution: int = 2, provider=None):
    """Latent mesh graph alone (assimilator_encoder.py:221-242 = encoder.py:244-268): (edge_index [2, E] int64 in reference
    order, edge_attr [E, 2], dst-sorted plan).  Same mesh provider as ``build_forecast_graphs`` (real h3 when importable),
    so that every graph of one model numbers the cells the same way."""
    provider = provider if provider is not None else _mesh.get_provider()
    if not isinstance(provider, _mesh.H3Like):  # pragma: no cover - literal loops of encoder.py:255-263 over real h3
        base, rank = _h3_base(provider, resolution)
        ls, ld, dl = [], [], []
        for c in base:
            for hcell in provider.grid_disk(c, 1):
                ls.append(rank[c]); ld.append(rank[hcell])
                dl.append(provider.great_circle_distance(provider.cell_to_latlng(c), provider.cell_to_latlng(hcell), unit="rads"))
        src, dst, attr = np.array(ls, dtype=np.int64), np.array(ld, dtype=np.int64), _sincos(np.array(dl))
        return (torch.from_numpy(np.stack([src, dst])), torch.from_numpy(attr), plan_from_coo(src, dst, len(base), len(base), attr))
    m = _mesh.get_mesh(resolution)
    M = m.num
    ptr, idx = m.disk1_csr()
    src = np.repeat(np.arange(M, dtype=np.int64), np.diff(ptr))
    dst = idx
    d = _mesh.haversine_rads(m.lat[src], m.lon[src], m.lat[dst], m.lon[dst])
    attr = _sincos(d)
    return (torch.from_numpy(np.stack([src, dst]).astype(np.int64)), torch.from_numpy(attr), plan_from_coo(src, dst, M, M, attr))


def build_observation_graph(lat_lon_heights, resolution: int = 2, provider=None):
    """Bipartite observation -> mesh graph of the assimilation encoder (assimilator_encoder.py:166-219): observation i is
    connected to the cell that contains it, at row ``N + (M - 1 - rank(cell))`` of the [observations ; mesh] table;
    edge attributes [sin d, cos d, height].  Returns (edge_index [2, N] int64 in reference order, edge_attr [N, 3],
    dst-sorted plan over (observation rows, mesh rows))."""
    llh = np.asarray(lat_lon_heights, dtype=np.float64).reshape(-1, 3)
    provider = provider if provider is not None else _mesh.get_provider()
    if not isinstance(provider, _mesh.H3Like):  # pragma: no cover - literal loops of assimilator_encoder.py:166-219 over real h3
        base, rank = _h3_base(provider, resolution)
        N, M = llh.shape[0], len(base)
        cells = [provider.latlng_to_cell(float(a), float(b), resolution) for a, b, _ in llh]
        d = np.array([provider.great_circle_distance((float(llh[i, 0]), float(llh[i, 1])), provider.cell_to_latlng(c), unit="rads")
                      for i, c in enumerate(cells)])
        attr = np.stack([np.sin(d), np.cos(d), llh[:, 2]], axis=1).astype(np.float32)
        src = np.arange(N, dtype=np.int64)
        dst_row = np.array([M - 1 - rank[c] for c in cells], dtype=np.int64)
        return (torch.from_numpy(np.stack([src, dst_row + N])), torch.from_numpy(attr), plan_from_coo(src, dst_row, N, M, attr))
    m = _mesh.get_mesh(resolution)
    N, M = llh.shape[0], m.num
    cell = m.locate(llh[:, 0], llh[:, 1])
    d = _mesh.haversine_rads(llh[:, 0], llh[:, 1], m.lat[cell], m.lon[cell])
    attr = np.stack([np.sin(d), np.cos(d), llh[:, 2]], axis=1).astype(np.float32)
    src = np.arange(N, dtype=np.int64)
    dst_row = (M - 1 - cell).astype(np.int64)
    edge_index = torch.from_numpy(np.stack([src, dst_row + N]).astype(np.int64))
    return edge_index, torch.from_numpy(attr), plan_from_coo(src, dst_row, N, M, attr)
