"""HIP graphs behind a block's ``forward`` -- the eager training step stops being bound by the host (round 5).

The reference's trainer calls the blocks eagerly and under gradient checkpointing (sd_mmfs.py:138-141: every ``MMFSBlock``;
modeling_llama_mmfs.py:700-717: the decoder layer around every ``LlamaMMFSAttention``).  A block is 40-120 small launches
each way; at BASELINE config 4 the 13 blocks' step is 4 075 launches that the host issues in 34-40 ms while their kernels
take 23 (profiles/r05zz_module_bench.jsonl).  ``GraphedTrainingStep`` (mmfs_amd/graphs.py) removes that by recording the
WHOLE step, which needs the caller to restructure its loop.  This module does it behind the unchanged call:

  * ``graphed_call(owner, fn, args, recompute)`` is what ``MMFSBlock.forward`` / ``LlamaMMFSAttention.forward`` run their
    body through.  Calls are told apart by a KEY -- shapes, strides, dtypes and ``requires_grad`` of the tensor arguments,
    the other arguments by value, the module's parameters by address (and by version where a result could be baked in),
    grad mode, training mode, the package's cache epoch.  The first ``capture_after`` calls with a key run eagerly (they
    are also the warm-up a capture needs); the next one records HIP graphs over static buffers; from then on a call is:
    copy the arguments into the buffers, one graph launch, clone the result.
  * with gradients it is an ``autograd.Function`` whose inputs are the tensor arguments AND the module's parameters, so
    parameter gradients reach ``p.grad`` (and any hook on it: DDP, DeepSpeed) the ordinary way.  ``recompute=True``
    (the checkpointed blocks): the forward graph runs without autograd and keeps nothing, the backward graph is
    "forward with autograd + backward" -- what ``torch.utils.checkpoint`` does, as one launch (and like its forward, the
    recorded one runs the with-gradients formulation of the modules: a replayed call returns the eager call's bits).  ``recompute=False``:
    the forward graph is recorded WITH autograd, its saved activations live in the graph's private pool, the backward
    graph is the backward alone (``torch.cuda.make_graphed_callables``' scheme, per call and lazily).
  * nothing is assumed about the caller's order of calls.  Every static buffer counts its writes, and the Function saves
    the caller's own tensors (they are alive anyway).  A backward whose input buffers were written since its forward
    (two forwards before the first backward; a shared bank buffer another module used) copies the caller's tensors in
    again before it replays; one whose saved ACTIVATIONS are gone (``recompute=False`` and a later forward of the same
    key) does not replay at all: it recomputes eagerly from the caller's tensors -- slower, never wrong.
  * a capture that fails (an op that synchronises, an allocation refused) marks the key as refused: eager from then on.
  * a call that is bound by the DEVICE is not replayed.  A replay saves host time and costs device time (the copies
    into and out of the static buffers); where the kernels of a call take longer than the host needs to issue them it
    only costs (the LLM layers at 2048 tokens: 16.0 -> 17.0 ms when replayed, r05zzd).  ``graphed_call`` runs the
    caller's plain path itself and knows its host time; after recording it replays the forward graph once on an idle
    device; a key whose graph takes longer than ``device_bound_ratio`` x that host time gives its graphs back and stays
    on the plain path.

Not used (the caller's plain path runs): CPU tensors, inside another capture, under saved-tensor hooks (a non-reentrant
checkpoint AROUND the caller, activation offloading), under autocast / inference mode / an active
``TorchDispatchMode`` or the shim's launch-event log (someone is counting ops: let them see the ops),
``mmfs_amd.graphed.enabled = False``.
"""
import gc
import threading
import time
import weakref

import torch

try:                                                   # (the mechanism of torch.func.functional_call; without it: no graphs)
    from torch.nn.utils.stateless import _reparametrize_module as _reparametrize
except Exception:                                      # pragma: no cover
    _reparametrize = None

import MultiScaleDeformableAttention as MSDA

from .levels import cache_epoch, hook_free, tensor_version

enabled = True
capture_after = 2            # eager calls with a key before it is recorded
max_entries = 4              # recorded keys per module (least recently used goes first)
stage_bytes = 4 << 20        # read-only tensor arguments at least this large share ONE static copy across modules
device_bound_ratio = 1.0     # a recorded forward that runs longer on the device than this x its plain path's host time is dropped
plentiful_fraction = 0.5     # "memory is plentiful" = this much of the device's memory is free (memory_is_plentiful)
share_pool = True            # the recompute-mode graphs of all modules record into one memory pool (their transients overlap)
capture_error_mode = "thread_local"
trace = None                 # a callable(str): debugging aid

_pools = {}                  # device index -> (graph memory pool shared by the recompute-mode graphs, the entries recorded into it)
# (argument place, shape, stride, dtype, device) -> _Static shared by every entry that reads such an argument.  Held WEAKLY: the
# entries that read a stage keep it alive (``_Entry.static``); when the last of them goes -- LRU eviction, a moved cache epoch,
# a refused capture, the module itself -- so does the bank-sized buffer (a plain dict kept one device buffer per
# distinct bank shape ever captured, for the life of the process)
_stages = weakref.WeakValueDictionary()
stats = {"captures": 0, "replays": 0, "eager_backward": 0, "refused": 0, "device_bound": 0}
_recording = threading.Lock()        # one capture at a time in the process (a caller's threads; the autograd engine's thread)


class _Static:
    """A buffer recorded graphs read.  ``gen`` counts its writes."""
    __slots__ = ("t", "gen", "src", "ver", "__weakref__")

    def __init__(self, like, requires_grad=False):
        self.t = torch.empty_strided(like.shape, like.stride(), dtype=like.dtype, device=like.device)
        if requires_grad:
            self.t.requires_grad_(True)
        self.gen, self.src, self.ver = 0, None, None

    def load(self, x, shared):
        if shared:
            # the same tensor object, unmodified: the 13 blocks of a net are handed one bank
            if self.src is not None and self.src() is x and self.ver == tensor_version(x):
                return self.gen
            self.src, self.ver = weakref.ref(x), tensor_version(x)
        with torch.no_grad():
            self.t.copy_(x)
        self.gen += 1
        return self.gen


def _t(msg):
    if trace is not None:
        trace(msg)


def _pool(dev):
    """The shared pool and the set of entries whose graphs keep it alive.  A handle whose graphs have all been destroyed
    (the model was deleted) names a pool the allocator has released: recording into it again fails inside
    ``capture_begin`` -- after the generator has been put into capture mode, which then stays on (r05g6-8) -- so a pool
    nobody holds any more is replaced by a fresh one."""
    p = _pools.get(dev.index)
    if p is None or len(p[1]) == 0:
        p = _pools[dev.index] = (torch.cuda.graph_pool_handle(), weakref.WeakSet())
    return p


def _reset_generator(dev):
    """A capture that failed before it began leaves the default generator in capture mode (every later random number
    raises "Offset increment outside graph capture"); the next successful capture takes it out again: make one."""
    try:
        x = torch.zeros(8, device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=capture_error_mode):
            x.add_(1.0)
    except Exception:
        pass


def _hashable(a):
    if a is None or isinstance(a, (bool, int, float, str, torch.dtype)):
        return a
    if isinstance(a, (tuple, list)):
        return tuple(_hashable(v) for v in a)
    raise TypeError(type(a))


class _Table(dict):
    """A module's recorded calls.  Lives in the module's ``__dict__``; a copy or a pickle of the module starts empty."""
    epoch = None

    def __deepcopy__(self, memo):
        return _Table()

    def __reduce__(self):
        return (_Table, ())


class _Entry:
    def __init__(self, key):
        self.key, self.seen, self.state, self.tick = key, 0, 0, 0       # state: 0 counting, 1 recorded, -1 refused, -2 device-bound
        self.host_s = float("inf")                                      # host time of the plain path's forward (least seen)

    def drop(self, state):
        """Gives the graphs and their buffers back (refused, device-bound); the shared pool no longer counts this entry
        among its holders."""
        self.state = state
        holders = self.__dict__.pop("holders", None)
        if holders is not None:
            holders.discard(self)
        for name in ("fwd", "bwd", "out", "gout", "gin", "static"):
            self.__dict__.pop(name, None)

    def time_forward(self, dev):
        """Device time of the recorded forward, on an idle device (seconds)."""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        best = float("inf")
        for _ in range(2):
            ev0.record()
            self.fwd.replay()
            ev1.record()
            ev1.synchronize()
            best = min(best, ev0.elapsed_time(ev1) * 1e-3)
        return best

    # ---- what a call looks like: positions of the tensor arguments in ``args`` (an argument passed twice is one input)
    def bind(self, owner, fn, args, recompute, need_grad):
        self.owner, self.fn, self.recompute, self.need_grad = owner, fn, recompute, need_grad
        self.template = [None if isinstance(a, torch.Tensor) else a for a in args]     # (the call's tensors are not kept)
        self.slots, self.dyn_pos, seen = [], [], {}
        for i, a in enumerate(args):
            if isinstance(a, torch.Tensor):
                j = seen.get(id(a))
                if j is None:
                    j = seen[id(a)] = len(self.dyn_pos)
                    self.dyn_pos.append(i)
                self.slots.append((i, j))
        named = [(k, p) for k, p in owner.named_parameters() if p.requires_grad] if need_grad else []
        self.names, self.params = [k for k, _ in named], [p for _, p in named]

    def call_args(self, dyn):
        out = list(self.template)
        for i, j in self.slots:
            out[i] = dyn[j]
        return out

    def dyn_of(self, args):
        return [args[i] for i in self.dyn_pos]

    # ---- recording
    def capture(self, args):
        dyn = self.dyn_of(args)
        dev = dyn[0].device
        self.req = [bool(x.requires_grad) and self.need_grad for x in dyn]
        self.shared = [(not x.requires_grad) and x.numel() * x.element_size() >= stage_bytes for x in dyn]
        self.static = []
        for j, (x, rq, sh) in enumerate(zip(dyn, self.req, self.shared)):
            if sh:
                # (keyed by the argument's place too: a call's bank and its normalised bank have one shape)
                k = (j, tuple(x.shape), tuple(x.stride()), x.dtype, x.device)
                s = _stages.get(k)
                if s is None:
                    s = _stages[k] = _Static(x)
            else:
                s = _Static(x, rq and not self.recompute)
            self.static.append(s)
        self.load(dyn)
        ins = [s.t for s in self.static]
        holders = None
        self.pooled = self.recompute and share_pool
        if self.pooled:
            pool, holders = _pool(dev)
        else:
            pool = torch.cuda.graph_pool_handle()
        self.run_gen = 0
        # one run of everything that will be recorded, on a side stream (library handles, workspaces, kept tables)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if self.need_grad:
                outs, lv, pal = self._with_grad(ins)
                gout = [torch.zeros_like(o) for o in outs]
                self._grads(outs, lv, pal, gout)
                del outs, lv, pal
            else:
                with torch.no_grad():
                    outs = self._outs(self.fn(*self.call_args(ins)))
                del outs
        torch.cuda.current_stream(dev).wait_stream(side)
        _t("warm-up done")
        self.fwd = torch.cuda.CUDAGraph()
        if not self.need_grad:
            with torch.cuda.graph(self.fwd, pool=pool, capture_error_mode=capture_error_mode):
                with torch.no_grad():
                    outs = self._outs(self.fn(*self.call_args(ins)))
            self.out = [o.detach() for o in outs]
            del outs
        elif self.recompute:
            # the forward of a non-reentrant checkpoint runs WITH autograd (its saved tensors are dropped, not its
            # formulation changed): the same kernels here, so that a replayed call returns the eager call's bits
            # (the modules' no-grad formulations round at other points: one 16-bit step apart, r05g9)
            with torch.cuda.graph(self.fwd, pool=pool, capture_error_mode=capture_error_mode):
                outs, lv, pal = self._with_grad(ins)
            self.out = [o.detach() for o in outs]
            del outs, lv, pal
        _t("forward recorded")
        if self.need_grad:
            self.bwd = torch.cuda.CUDAGraph()
            if self.recompute:
                self.gout = [torch.zeros_like(o) for o in self.out]
                with torch.cuda.graph(self.bwd, pool=pool, capture_error_mode=capture_error_mode):
                    _t("backward: recording")
                    outs, lv, pal = self._with_grad(ins)
                    _t("backward: forward part issued")
                    try:
                        grads = self._grads(outs, lv, pal, self.gout)
                    except BaseException as ex:
                        import traceback
                        _t("backward: FAILED " + "".join(traceback.format_exception(type(ex), ex, ex.__traceback__))[-3000:])
                        raise
                    _t("backward: issued")
            else:
                with torch.cuda.graph(self.fwd, pool=pool, capture_error_mode=capture_error_mode):
                    outs, lv, pal = self._with_grad(ins, leaves=ins)
                self.out = [o.detach() for o in outs]
                self.gout = [torch.zeros_like(o) for o in self.out]
                with torch.cuda.graph(self.bwd, pool=pool, capture_error_mode=capture_error_mode):
                    grads = self._grads(outs, lv, pal, self.gout)
            del outs, lv, pal
            self.gin = list(grads)                 # aligned with [inputs that want a gradient] + params
            _t("backward recorded")
        if holders is not None:
            holders.add(self)
            self.holders = holders
        self.state = 1
        stats["captures"] += 1

    def _outs(self, res):
        self.single = isinstance(res, torch.Tensor)
        return [res] if self.single else list(res)

    def _with_grad(self, ins, leaves=None):
        """``fn`` with autograd on fresh leaves: detached views of the inputs, and ALIASES of the module's parameters
        (swapped in for the call, ``torch.func.functional_call``'s mechanism).  Not the parameters themselves: a
        parameter's gradient-accumulation node lives as long as any graph that used it -- a trainer's ``loss`` of the
        previous step is still alive when the next forward runs -- and belongs to the stream it was made on, normally the
        legacy default stream; a backward pass RECORDED on a capturing stream that ends at such a node fails
        ("operation would make the legacy stream depend on a capturing blocking stream"), and ending that capture
        takes the process down inside the HIP runtime (r05g3-5)."""
        with torch.enable_grad():
            lv = leaves if leaves is not None else [x.detach().requires_grad_(rq) for x, rq in zip(ins, self.req)]
            pal = [p.detach().requires_grad_(True) for p in self.params]
            with _reparametrize(self.owner, dict(zip(self.names, pal))):
                outs = self._outs(self.fn(*self.call_args(lv)))
        return outs, lv, pal

    def _grads(self, outs, lv, pal, gout):
        wrt = [x for x, rq in zip(lv, self.req) if rq] + pal
        pick = [(o, g) for o, g in zip(outs, gout) if o.requires_grad]
        if not pick or not wrt:
            return [None] * len(wrt)
        return torch.autograd.grad([o for o, _ in pick], wrt, [g for _, g in pick], allow_unused=True)

    # ---- a call
    def load(self, dyn):
        return tuple(s.load(x, sh) for s, x, sh in zip(self.static, dyn, self.shared))

    def forward(self, dyn):
        gens = self.load(dyn)
        self.run_gen += 1
        self.fwd.replay()
        stats["replays"] += 1
        return gens, [o.clone() for o in self.out]

    def result(self, outs):
        return outs[0] if self.single else tuple(outs)

    def replayable(self, run_gen):
        """Whether the backward graph can serve the call whose forward was replay ``run_gen``: always when it recomputes
        (it needs the inputs only); else only while no later forward has overwritten the activations it saved."""
        return self.state == 1 and (self.recompute or run_gen == self.run_gen)

    def backward(self, dyn, gens, gouts):
        for s, x, sh, g in zip(self.static, dyn, self.shared, gens):
            if s.gen != g:                         # (written since this call's forward: the caller's tensor again)
                s.load(x, sh)
        with torch.no_grad():
            for buf, g in zip(self.gout, gouts):
                if g is None:
                    buf.zero_()
                else:
                    buf.copy_(g)
        self.bwd.replay()
        stats["replays"] += 1
        # copies: the graphs of all modules share one pool, and what is an OUTPUT here may be a transient of a graph that
        # was recorded earlier -- whose next replay (the previous block's backward) comes before autograd has consumed
        # this gradient (a projected bank's: at the very end of the backward pass)
        # (and in keep mode too: the static gradient buffers are overwritten by this entry's next backward replay -- a
        # gradient someone keeps past that point, torch.autograd.grad's result or a hook's, must not change under them:
        # ADVICE r5)
        return [None if g is None else g.clone() for g in self.gin]

    def eager_backward(self, dyn, gouts):
        """The way back without the graphs (their buffers moved on since this call's forward): recompute from the
        caller's tensors, as ``torch.utils.checkpoint`` would."""
        stats["eager_backward"] += 1
        outs, lv, pal = self._with_grad(list(dyn))
        gout = [torch.zeros_like(o) if g is None else g for o, g in zip(outs, gouts)]
        return list(self._grads(outs, lv, pal, gout))


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, entry, n_dyn, *flat):
        dyn = flat[:n_dyn]
        gens, outs = entry.forward(dyn)
        ctx.entry, ctx.gens, ctx.run_gen = entry, gens, entry.run_gen
        ctx.save_for_backward(*dyn)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        e = ctx.entry
        if e.replayable(ctx.run_gen):
            grads = e.backward(ctx.saved_tensors, ctx.gens, gouts)
        else:
            grads = e.eager_backward(ctx.saved_tensors, gouts)
        it = iter(grads)
        gin = [next(it) if rq else None for rq in e.req]
        return (None, None) + tuple(gin) + tuple(it)


def _eligible(args):
    if not enabled or _reparametrize is None or not torch.cuda.is_available():
        return None
    first = None
    for a in args:
        if isinstance(a, torch.Tensor):
            if not a.is_cuda or (first is not None and a.device != first.device):
                return None
            if first is None:
                first = a
    if first is None or torch.is_autocast_enabled() or torch.is_inference_mode_enabled():
        return None
    if torch.compiler.is_compiling():                  # (a tracing compiler wants the operations, not a replay)
        return None
    if first.device.index != torch.cuda.current_device():       # (a capture records on the current device's stream)
        return None
    if MSDA._event_log is not None:                    # (someone brackets every launch with events: bench.py's kernel pass)
        return None
    # someone intercepts what autograd saves (a NON-reentrant checkpoint around the caller, activation offloading): the
    # plain path, in the forward and in the recomputation alike -- they count and match the saved tensors of the two
    hooks = getattr(torch._C._autograd, "_top_saved_tensors_default_hooks", None)
    if hooks is not None and hooks(True) is not None:
        return None
    depth = getattr(torch._C, "_len_torch_dispatch_stack", None)
    if (depth is not None and depth() > 0) or torch.cuda.is_current_stream_capturing():
        return None
    return first


def _key(owner, args, recompute, grad):
    parts = [recompute, grad, owner.training, cache_epoch()]
    for a in args:
        if isinstance(a, torch.Tensor):
            parts.append((tuple(a.shape), tuple(a.stride()), a.dtype, a.requires_grad and grad))
        else:
            parts.append(_hashable(a))
    ident = {}
    for i, a in enumerate(args):                       # which arguments are the same tensor
        if isinstance(a, torch.Tensor):
            parts.append(ident.setdefault(id(a), i))
    for p in owner.parameters():
        # a trainable parameter is read at its address every replay, with or without gradients (graphed_call only runs in
        # training mode, where no module keeps anything derived from a parameter: nothing is baked in) -- its VERSION must
        # not be in the key: every optimizer.step() bumps it, and the no-grad forward of a reentrant-checkpointed decoder
        # layer would re-record every step.  A frozen one may have been baked into something the recorded
        # kernels read: its version counts.
        parts.append((p.data_ptr(), p.dtype, p.requires_grad, -2 if p.requires_grad else tensor_version(p)))
    for b in owner.buffers():
        parts.append((b.data_ptr(), b.dtype, tensor_version(b)))
    for m in owner.modules():                          # the switches that choose a module's path (``_behaviour_flags``)
        for f in getattr(type(m), "_behaviour_flags", ()):
            parts.append(_hashable(getattr(m, f, None)))
    return tuple(parts)


_plenty = {}                 # device index -> (when asked, answer)


def memory_is_plentiful(dev):
    """Whether at least ``plentiful_fraction`` of the device's memory is free (asked of the driver at most once a second):
    what a checkpointed block's ``graph_keeps_activations = "auto"`` goes by."""
    now = time.monotonic()
    hit = _plenty.get(dev.index)
    if hit is None or now - hit[0] > 1.0:
        free, total = torch.cuda.mem_get_info(dev)
        hit = _plenty[dev.index] = (now, free >= plentiful_fraction * total)
    return hit[1]


def _hooks_free(owner):
    """No hook on the owner or on anything inside it (a replay runs no Python: hooks would fire at recording time only --
    activation capture, profilers, parameter-gathering pre-hooks).  Asked once per (module, hook state): the
    hook dictionaries' sizes are the state."""
    sizes = tuple(len(m._forward_hooks) + len(m._forward_pre_hooks) + len(m._backward_hooks)
                  + len(getattr(m, "_backward_pre_hooks", ())) for m in owner.modules())
    hit = owner.__dict__.get("_graphed_hooks")
    if hit is None or hit[0] != sizes:
        hit = owner.__dict__["_graphed_hooks"] = (sizes, all(hook_free(m) for m in owner.modules()))
    return hit[1] and hook_free(owner)             # (the process-wide hook lists are looked at every time)


def graphed_call(owner, fn, args, recompute, plain):
    """``fn(*args)`` through recorded HIP graphs when this call's key has them, else ``plain()`` -- the caller's own
    statement of the same call (``fn(*args)``, or it under ``torch.utils.checkpoint``).  ``owner``: the module whose
    parameters ``fn`` reads."""
    if _eligible(args) is None or not _hooks_free(owner):
        return plain()
    grad = torch.is_grad_enabled()
    try:
        key = _key(owner, args, recompute, grad)
    except TypeError:
        return plain()
    table = owner.__dict__.get("_graphed")
    if table is None:
        table = owner.__dict__["_graphed"] = _Table()
    if table.epoch != key[3]:
        # the package's cache epoch moved (a mode change, a state-dict load): no key of before can come back -- drop
        # what was recorded under it with its buffers
        table.clear()
        table.epoch = key[3]
    e = table.get(key)
    if e is None:
        e = table[key] = _Entry(key)
        if len(table) > 4 * max_entries:               # (keys that never came back)
            for k in [k for k, v in table.items() if v.state != 1 and v is not e][:len(table) // 2]:
                del table[k]
    e.seen += 1
    owner.__dict__["_graphed_tick"] = e.tick = owner.__dict__.get("_graphed_tick", 0) + 1
    if e.state == 0 and e.seen > capture_after and _recording.acquire(blocking=False):
        need = grad and (any(isinstance(a, torch.Tensor) and a.requires_grad for a in args)
                         or any(p.requires_grad for p in owner.parameters()))
        # (no garbage collection while a stream records: a dead module's graphs destroyed in the middle of a capture
        # are driver calls a capture does not allow)
        collecting = gc.isenabled()
        gc.disable()
        try:
            e.bind(owner, fn, args, recompute, need)
            e.capture(args)
            if device_bound_ratio is not None and e.host_s < float("inf"):
                dev_s = e.time_forward(args[e.dyn_pos[0]].device)
                _t("forward: %.3f ms on the device, %.3f ms of host time on the plain path" % (dev_s * 1e3, e.host_s * 1e3))
                if dev_s > device_bound_ratio * e.host_s:
                    e.drop(-2)
                    stats["device_bound"] += 1
                    stats["captures"] -= 1
            live = [v for v in table.values() if v.state == 1]
            if len(live) > max_entries:
                old = min((v for v in live if v is not e), key=lambda v: v.tick)
                del table[old.key]
        except Exception as ex:
            e.drop(-1)
            e.error = repr(ex)
            stats["refused"] += 1
            _t("refused: " + e.error[:500])
            _reset_generator(args[e.dyn_pos[0]].device if getattr(e, "dyn_pos", None) else torch.device("cuda"))
        finally:
            _recording.release()
            if collecting:
                gc.enable()
    if e.state != 1:
        t0 = time.perf_counter()
        out = plain()
        e.host_s = min(e.host_s, time.perf_counter() - t0)      # (the launches' host time: the least of the sightings)
        return out
    dyn = e.dyn_of(args)
    if not e.need_grad:
        return e.result(e.forward(dyn)[1])
    return e.result(list(_GraphedFn.apply(e, len(dyn), *dyn, *e.params)))
