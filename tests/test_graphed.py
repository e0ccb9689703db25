"""CPU: the host logic of mmfs_amd/graphed.py that needs no device -- on CPU tensors a call takes the caller's plain path; what
tells two calls apart (the key); a module that is copied or pickled starts without recorded calls.  The recording and the
replays themselves: tests/test_graphed_gpu.py."""
import copy
import pickle

import pytest
import torch

from mmfs_amd import graphed
from mmfs_amd.blocks.sd_mmfs import MMFSBlock


def block():
    return MMFSBlock(attn_dim=64, query_dim=32, feat_dim=64, num_heads=4, n_points=2, n_levels=2, spatial_shapes=[8, 4],
                     base_spatial_shape=8, gradient_checkpointing=True, grid_size=8)


def test_cpu_tensors_take_the_plain_path():
    blk = block()
    x = torch.randn(2, 32, 8, 8)
    assert graphed.graphed_call(blk, lambda *a: "recorded", (x, None, [(8, 8)]), True, plain=lambda: "plain") == "plain"
    assert "_graphed" not in blk.__dict__                  # (not even counted: nothing to record on this device)


def test_what_tells_two_calls_apart():
    blk = block()
    x, y = torch.randn(2, 32, 8, 8), torch.randn(2, 32, 8, 8)
    k = graphed._key(blk, (x, None, [(8, 8), (4, 4)], x), True, True)
    assert k == graphed._key(blk, (x, None, [(8, 8), (4, 4)], x), True, True) and hash(k) is not None
    assert k == graphed._key(blk, (y, None, [(8, 8), (4, 4)], y), True, True)             # other tensors of the same kind
    assert k != graphed._key(blk, (x, None, [(8, 8), (4, 4)], y), True, True)             # the SAME tensor twice is part of it
    assert k != graphed._key(blk, (x, None, [(8, 8), (2, 2)], x), True, True)             # an argument by value
    assert k != graphed._key(blk, (x[:1], None, [(8, 8), (4, 4)], x[:1]), True, True)     # a shape
    assert k != graphed._key(blk, (x, None, [(8, 8), (4, 4)], x), True, False)            # grad mode
    assert k != graphed._key(blk, (x.requires_grad_(True), None, [(8, 8), (4, 4)], x), True, True)
    x.requires_grad_(False)
    blk.mmfs.stack_heads_in_training = False                                              # a path switch of a sub-module
    assert k != graphed._key(blk, (x, None, [(8, 8), (4, 4)], x), True, True)
    blk.mmfs.stack_heads_in_training = True
    blk.conv.weight.requires_grad_(False)                                                 # a parameter that is frozen
    assert k != graphed._key(blk, (x, None, [(8, 8), (4, 4)], x), True, True)
    with pytest.raises(TypeError):
        graphed._key(blk, (x, object()), True, True)                                      # (the caller then takes the plain path)


def test_a_copied_or_pickled_module_starts_without_recorded_calls():
    blk = block()
    table = blk.__dict__["_graphed"] = graphed._Table()
    table["key"] = graphed._Entry("key")
    table.epoch = 7
    for twin in (copy.deepcopy(blk), pickle.loads(pickle.dumps(blk))):
        assert isinstance(twin.__dict__["_graphed"], graphed._Table) and len(twin.__dict__["_graphed"]) == 0
        assert twin.__dict__["_graphed"].epoch is None
    assert len(table) == 1


def test_memory_is_plentiful_asks_the_driver_at_most_once_a_second(monkeypatch):
    """``MMFSBlock.graph_keeps_activations = "auto"`` goes by ``memory_is_plentiful``: at least ``plentiful_fraction`` of the
    device's memory free, asked of the driver once and remembered for a second."""
    calls = []

    def mem_get_info(dev):
        calls.append(dev)
        return (60 << 30, 100 << 30) if len(calls) == 1 else (10 << 30, 100 << 30)
    monkeypatch.setattr(torch.cuda, "mem_get_info", mem_get_info)
    graphed._plenty.clear()
    dev = torch.device("cuda", 0)
    assert graphed.memory_is_plentiful(dev) and graphed.memory_is_plentiful(dev) and len(calls) == 1
    graphed._plenty[0] = (graphed._plenty[0][0] - 2.0, True)          # (a second later)
    assert not graphed.memory_is_plentiful(dev) and len(calls) == 2
    graphed._plenty.clear()


def test_a_trainable_parameters_version_is_not_in_the_key():
    """``optimizer.step()`` bumps a trainable parameter's version; the recorded kernels read it at its address,
    so neither the with-gradients nor the no-grad key of a call may change with it (a frozen parameter's version may have
    been baked in: it stays in the key)."""
    from mmfs_amd import graphed
    m = torch.nn.Linear(4, 4)
    m.bias.requires_grad_(False)
    x = torch.zeros(2, 4)
    keys = {g: graphed._key(m, (x,), False, g) for g in (True, False)}
    with torch.no_grad():
        m.weight.add_(1.0)                                  # what an optimizer step does
    assert all(graphed._key(m, (x,), False, g) == keys[g] for g in (True, False))
    with torch.no_grad():
        m.bias.add_(1.0)                                    # a frozen parameter moved: another key
    assert all(graphed._key(m, (x,), False, g) != keys[g] for g in (True, False))


def test_shared_stages_are_held_weakly():
    """the bank-sized static copies shared between recorded calls live as long as an entry that reads them."""
    import weakref
    from mmfs_amd import graphed
    assert isinstance(graphed._stages, weakref.WeakValueDictionary)
    s = graphed._Static(torch.zeros(3))
    graphed._stages[("probe",)] = s
    assert ("probe",) in graphed._stages
    del s
    import gc
    gc.collect()
    assert ("probe",) not in graphed._stages

