"""Level tables (``spatial_shapes`` / ``level_start_index``) for the MMFS callers.

The reference rebuilds both tensors from Python lists on every forward
(modeling_llama_mmfs.py:298-308, sd_mmfs.py:31-41): a host->device copy per call.  Here
they are built once per (shapes, n_images, device) and cached; the tensors carry the
"canonical packing" mark the op shim looks for
(``MultiScaleDeformableAttention.levels_are_canonical``), so the backward never has to
copy them back to the host to choose its algorithm.
"""
import numpy as np
import torch

_cache = {}


def make_level_tables(shapes_per_image, n_images, device):
    """shapes_per_image: [(H, W), ...] of one image; the op sees them repeated n_images
    times ("multiple images are extra levels").
    Returns (spatial_shapes [n*L, 2] int64, level_start_index [n*L] int64, S)."""
    key = (tuple((int(h), int(w)) for h, w in shapes_per_image), int(n_images), str(torch.device(device)))
    hit = _cache.get(key)
    if hit is not None:
        return hit
    # The tensors outlive the call that made them.  Made under torch.inference_mode() (an evaluation pass
    # before training) they would be inference tensors, and every later training step would fail at
    # save_for_backward ("Inference tensors cannot be saved for backward"): build them outside any mode.
    with torch.inference_mode(False), torch.no_grad():
        host = torch.tensor(list(key[0]) * int(n_images), dtype=torch.long).reshape(-1, 2)
        px = host[:, 0] * host[:, 1]
        start_host = px.cumsum(0) - px
        S = int(px.sum())
        shapes = host.to(device)
        start = start_host.to(device)
    # pre-seed the shim's cache: (versions, start ptr, S) -> (canonical, host shapes, host start)
    shapes._mmfs_canonical = ((shapes._version, start._version, start.data_ptr(), S), True,
                              np.ascontiguousarray(host.numpy(), dtype=np.int64),
                              np.ascontiguousarray(start_host.numpy(), dtype=np.int64))
    shapes._mmfs_host = host
    out = (shapes, start, S)
    _cache[key] = out
    return out


def host_shapes(spatial_shapes):
    """Host copy of a level table if it was built here, else None (no sync is forced)."""
    return getattr(spatial_shapes, "_mmfs_host", None)


def tensor_version(t):
    """The in-place version counter of ``t`` for identity caches (kept projections of a feature bank), or -1 for a tensor
    made inside ``torch.inference_mode()``: those have no counter (reading it raises).  Outside inference mode they
    cannot be modified at all; inside it an in-place change goes unseen -- a caller who does that clears the cache
    (``MMFSNet.clear_feature_cache`` / ``LlamaMMFSSchedule.clear_cache``)."""
    return -1 if t.is_inference() else t._version



# ---- kept no-grad artefacts (folded weights, parameter-only tables, projected banks) ------------------------------
# They are keyed on (data pointer, version counter) of the parameters they were made from -- and an in-place write
# through ``param.data`` (DeepSpeed's bit16 update into its flat buffer, ``EMAModel.copy_to``, a checkpoint loaded
# with ``param.data.copy_``) moves neither.  So every cache signature also carries this EPOCH, which
# moves whenever one of this package's modules changes mode (``train()`` / ``eval()``: DeepSpeed's periodic
# evaluation does), loads a state dict, or is told to (``clear_caches()`` on the modules, ``invalidate_caches()``
# here); and nothing is kept while a module is in training mode.  What is left to the caller: parameters written
# through ``.data`` while the modules STAY in eval mode -- call ``mmfs_amd.invalidate_caches()`` after such a write
# (INTEGRATION.md 3).
_epoch = 0


def cache_epoch():
    return _epoch


def invalidate_caches():
    """Forget every kept fold / table / projected bank of every mmfs_amd module (they are rebuilt on the next no-grad
    call).  Cheap: one counter."""
    global _epoch
    _epoch += 1


class CacheInvalidation:
    """Mixin for the modules that keep something: mode changes and state-dict loads move the epoch."""

    def train(self, mode=True):
        if bool(mode) != self.training:     # (a generation loop that says model.eval() before every step keeps its folds)
            invalidate_caches()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        invalidate_caches()
        return super()._load_from_state_dict(*args, **kwargs)

    def clear_caches(self):
        """Public invalidation (also ``mmfs_amd.invalidate_caches()``): call after writing parameters through ``.data``
        in eval mode."""
        invalidate_caches()


def hook_free(layer):
    """No hook of any kind could observe a call of ``layer``: only then may a no-grad path evaluate the layer's
    mathematics without calling the module (folded weights, the small-token Linear kernel).  Forward AND backward
    hooks, the layer's own and the process-wide ones (profilers, activation capture) -- ADVICE r3."""
    from torch.nn.modules import module as _m
    if (layer._forward_hooks or layer._forward_pre_hooks or layer._backward_hooks
            or getattr(layer, "_backward_pre_hooks", None)):
        return False
    for name in ("_global_forward_hooks", "_global_forward_pre_hooks", "_global_backward_hooks",
                 "_global_backward_pre_hooks"):
        if getattr(_m, name, None):
            return False
    return True
