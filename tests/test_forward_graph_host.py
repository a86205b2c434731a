"""Host logic of graph_weather_amd.ForwardGraph that needs no GPU: what makes a captured graph stale (tests/test_gpu_round5.py
replays real captures on the device)."""
import pytest
import torch

import graph_weather_amd as gw
from graph_weather_amd.utils import deterministic_fill_, regular_lat_lons


def _model():
    model = gw.GraphWeatherForecaster(regular_lat_lons(30.0))
    deterministic_fill_(model, seed=0)
    return model.eval()


def test_capture_key_follows_weights_dtype_flags_and_input():
    model = _model()
    fg = gw.ForwardGraph(model)
    shape = (2, 72, 102)
    k0 = fg._state_key(shape, "cuda:0", torch.float32)
    assert k0 == fg._state_key(shape, "cuda:0", torch.float32)  # nothing changed: the same graph is replayed
    assert k0 != fg._state_key((1, 72, 102), "cuda:0", torch.float32)  # another batch size
    assert k0 != fg._state_key(shape, "cuda:1", torch.float32)
    assert k0 != fg._state_key(shape, "cuda:0", torch.float64)
    with torch.no_grad():
        next(model.parameters()).add_(0)  # what an optimizer step does to the version counter
    k1 = fg._state_key(shape, "cuda:0", torch.float32)
    assert k1 != k0
    model.set_compute_dtype("bf16x3")  # the packed weights the launches point at are others
    k2 = fg._state_key(shape, "cuda:0", torch.float32)
    assert k2 != k1
    model.set_deterministic(True)  # other kernels' arguments (carry records instead of atomics)
    assert fg._state_key(shape, "cuda:0", torch.float32) != k2


def test_load_state_dict_invalidates_the_capture_key():
    a, b = _model(), _model()
    fg = gw.ForwardGraph(a)
    k0 = fg._state_key((1, 72, 102), "cuda:0", torch.float32)
    a.load_state_dict(b.state_dict())
    assert fg._state_key((1, 72, 102), "cuda:0", torch.float32) != k0


def test_refuses_cpu_tensors_training_mode_and_a_first_call_without_input():
    model = _model()
    fg = model.graphed()
    assert isinstance(fg, gw.ForwardGraph) and fg.captures == 0
    with pytest.raises(RuntimeError, match="first call"):
        fg()
    with pytest.raises(RuntimeError, match="no CPU path"):
        fg(torch.zeros(1, 72, 102))
    model.train()
    with pytest.raises(RuntimeError, match="inference"):
        fg(torch.zeros(1, 72, 102))


def test_capture_key_follows_launch_shaping_attributes_and_invalidate_rewalks():
    """state that changes the launches but not the weights - streams of the mesh stack, checkpoint segments, efficient
    batching - is part of the key; modules swapped in after the first call are picked up by ``invalidate()``."""
    model = _model()
    fg = gw.ForwardGraph(model)
    shape = (2, 72, 102)
    k0 = fg._state_key(shape, "cuda:0", torch.float32)
    gp = model.processor.graph_processor
    keep = gp.streams
    gp.streams = 1 if keep != 1 else 2
    k1 = fg._state_key(shape, "cuda:0", torch.float32)
    assert k1 != k0
    gp.streams = keep
    assert fg._state_key(shape, "cuda:0", torch.float32) == k0
    model.processor.set_checkpoint_segments(3)
    assert fg._state_key(shape, "cuda:0", torch.float32) != k0
    model.processor.set_checkpoint_segments(0)
    # a parameter swapped in behind the cached module walk is invisible until invalidate()
    blk = model.processor.graph_processor.blocks[0].node_model.node_mlp
    lin = next(m for m in blk.modules() if isinstance(m, torch.nn.Linear))
    lin.weight = torch.nn.Parameter(lin.weight.detach().clone() * 2)
    assert fg._state_key(shape, "cuda:0", torch.float32) == k0
    fg.invalidate()
    assert fg._state_key(shape, "cuda:0", torch.float32) != k0


def test_auto_graph_policy_on_the_host():
    """graphed.AutoGraph: which calls may be replayed (eval + no_grad + small CUDA input, nothing that forbids a capture) - the
    decisions that need no GPU; the replays themselves are tests/test_gpu_round6.py."""
    from graph_weather_amd.graphed import AutoGraph

    model = _model()
    auto = AutoGraph(model)
    x = torch.zeros(1, 72, 102)
    with torch.no_grad():
        assert not auto.usable(x)  # a CPU tensor: the ordinary path raises "no CPU path"
    assert model.auto_graph is True
    import copy
    import pickle

    model.__dict__["_auto"] = auto  # (what forward() stores on first use): not copied, not pickled
    assert "_auto" not in copy.deepcopy(model).__dict__
    assert "_auto" not in pickle.loads(pickle.dumps(model)).__dict__
    with pytest.raises(RuntimeError, match="no CPU path"):
        with torch.no_grad():
            model(x)
    # a shallow copy (what nn.DataParallel's replicas are) does not inherit the original's policy object
    replica = model._replicate_for_data_parallel()
    assert replica.__dict__.get("_auto") is auto
    with torch.no_grad():
        assert replica._auto_graph_step(x) is None  # (CPU tensor: not usable - but the step has re-bound the policy to the replica)
    assert replica.__dict__["_auto"] is not auto and replica.__dict__["_auto"]._model_ref() is replica


def test_a_model_with_an_automatic_graph_is_freed_by_reference_count():
    """The model owns its AutoGraph (and through it the ForwardGraph and the captured HIP graph); neither may own the model back:
    in a cycle the graph would be destroyed by the cyclic collector at an arbitrary moment - inside another capture that aborts
    the process (round 6, `~CUDAGraph`: "operation not permitted when stream is capturing")."""
    import gc
    import weakref

    from graph_weather_amd.graphed import AutoGraph

    gc.collect()
    gc.disable()
    try:
        model = _model()
        auto = AutoGraph(model)
        auto._fg = gw.ForwardGraph(model, warmup=1, weak=True)
        auto._fg._state_key((1, 72, 102), "cuda:0", torch.float32)  # (the cached module walk holds sub-modules, not the model)
        model.__dict__["_auto"] = auto
        probe, fg_probe = weakref.ref(model), weakref.ref(auto._fg)
        del auto
        del model
        assert probe() is None and fg_probe() is None  # gone without a collector pass
    finally:
        gc.enable()
    fg = gw.ForwardGraph(_model(), weak=True)
    with pytest.raises(RuntimeError, match="no longer exists"):
        fg.model
