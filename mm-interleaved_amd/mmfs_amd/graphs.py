"""HIP-graph replay of the sampling-time ``MMFSNet`` schedule.

A denoising loop calls ``MMFSNet.forward`` 30 x (sd_pipeline_monkey_patch.py:181-200) with the same
shapes, the same feature tensors and the same mask; only the UNet's residuals change.  The 13 blocks
are ~450 small launches, and at B=8 the host needs longer to issue them (5.8 ms) than the GPU to run
them (5.1 ms).  ``GraphedMMFSNet`` records the schedule once into a HIP graph (``torch.cuda.CUDAGraph``
is hipGraph on ROCm) over static input buffers -- the projected feature bank is computed before the
capture and baked in -- and replays it per step: one launch, no Python between kernels.

The kernels underneath need nothing special for this: the C ABI takes its stream as an argument,
allocates nothing, and never synchronises (tests/test_modules_gpu.py checks the last).

``GraphedTrainingStep`` (round 4) does the same for a TRAINING step -- forward + backward recorded as one graph: the 13-block
net's step is ~5 000 launches that the host issues in 37-39 ms while their kernels take under 30.
"""
import torch

from .blocks.llama_mmfs import LlamaMMFSSchedule, ProjectedBank
from .blocks.sd_mmfs import MMFSNet, ProjectedFeatures


class GraphedMMFSNet:
    """``g = GraphedMMFSNet(net, sample, residuals, mmfs_features, mmfs_mask)`` captures;
    ``g(sample, residuals)`` replays for new residuals of the same shapes and returns
    ``(sample', residuals')`` exactly like ``net(sample, residuals, mmfs_features, mmfs_mask)``.
    The results live in buffers owned by the graph and are overwritten by the next replay.
    Inference only (no autograd through a replay)."""

    def __init__(self, net, sample, down_block_res_samples, mmfs_features, mmfs_mask, warmup=2):
        assert isinstance(net, MMFSNet) and sample.is_cuda
        assert not net.training, "GraphedMMFSNet is inference only: net.eval() first (nothing is folded or kept in training mode)"
        self.net = net
        with torch.no_grad():
            proj = mmfs_features if isinstance(mmfs_features, ProjectedFeatures) \
                else net.project_features(mmfs_features)
            self._proj, self._mask = proj, mmfs_mask.clone()
            self._sample = sample.clone()
            self._res = [r.clone() for r in down_block_res_samples]
            side = torch.cuda.Stream(device=sample.device)
            side.wait_stream(torch.cuda.current_stream(sample.device))
            with torch.cuda.stream(side):          # first calls fill the caches (level tables, position tables)
                for _ in range(warmup):
                    net(self._sample, self._res, proj, self._mask)
            torch.cuda.current_stream(sample.device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._out = net(self._sample, self._res, proj, self._mask)
            # The recorded kernels read the blocks' kept artefacts (folded weights, parameter-only tables, position
            # tables) at the addresses they had during the capture: hold them, so that an invalidation of the modules'
            # caches (a mode change, a state-dict load, ``invalidate_caches()``) cannot free what the graph replays.
            # A graph bakes the parameters of its capture in: re-capture after they change.
            self._pinned = [(b.mmfs._tables, b._conv_fold._kept, b.__dict__.get("_pos_cache"), b.mmfs.__dict__.get("_ratios_f32"))
                            for b in net._blocks()]

    @torch.no_grad()
    def __call__(self, sample, down_block_res_samples):
        assert len(down_block_res_samples) == len(self._res)
        self._sample.copy_(sample)
        torch._foreach_copy_(self._res, list(down_block_res_samples))
        self.graph.replay()
        return self._out


class GraphedLlamaMMFSStack:
    """HIP-graph replay of a decoder's MMFS layers for ONE shape of the token stream -- the decode step of a
    generation loop (one new token per sequence, the bank and the mask fixed: mm_interleaved.py:598-664), where the
    8-10 layers are ~850 small launches that the host issues several times more slowly than the GPU runs them.

    ``g = GraphedLlamaMMFSStack(layers, hidden, features, mask)`` projects the bank once (LlamaMMFSSchedule), records
    ``for k: h = layers[k](h, features, mask, value=bank.values[k], image_ranks=ranks, residual=h)`` (= h + layer(h)) over a static input buffer and
    ``g(hidden)`` replays it.  The dense LLaMA layers that sit between the MMFS layers in the real decoder are out
    of scope here (a caller that graphs its whole decode step captures these layers with the rest: the op and the
    modules are capture-safe as they are -- no device->host copy, current stream, allocator workspaces).
    Inference only; the result lives in a buffer owned by the graph."""

    def __init__(self, layers, hidden, vision_hidden_states, cross_attention_mask, warmup=2):
        assert hidden.is_cuda
        self.layers = list(layers)
        assert not any(l.training for l in self.layers), \
            "GraphedLlamaMMFSStack is inference only: layer.eval() first (nothing is folded or kept in training mode)"
        with torch.no_grad():
            bank = vision_hidden_states if isinstance(vision_hidden_states, ProjectedBank) \
                else LlamaMMFSSchedule(self.layers).project(vision_hidden_states)
            self._bank, self._mask = bank, cross_attention_mask.clone()
            self._hidden = hidden.clone()
            # (the mask is fixed for the life of the graph: the images' ranks are made here, once, not in it)
            self._ranks = self.layers[0].attn._image_relpos(self._mask, hidden.shape[1])
            side = torch.cuda.Stream(device=hidden.device)
            side.wait_stream(torch.cuda.current_stream(hidden.device))
            with torch.cuda.stream(side):          # first calls fill the caches (level tables)
                for _ in range(warmup):
                    self._run()
            torch.cuda.current_stream(hidden.device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._out = self._run()
            # (as GraphedMMFSNet: what the recorded kernels read stays alive with the graph)
            self._pinned = [(l.attn._tables, l._gate_fold._kept, getattr(l, "_gate_tanh", None), l.attn.__dict__.get("_ratios_f32"))
                            for l in self.layers]

    def _run(self):
        h = self._hidden
        for k, layer in enumerate(self.layers):
            h = layer(h, self._bank.bank, self._mask, value=self._bank.values[k], image_ranks=self._ranks, residual=h)
        return h

    @torch.no_grad()
    def __call__(self, hidden):
        self._hidden.copy_(hidden)
        self.graph.replay()
        return self._out


class GraphedTrainingStep:
    """Forward + backward of ``fn`` recorded ONCE into a HIP graph and replayed per step ("whole-network capture" restricted
    to the part of the network this package owns).

    ``fn(*inputs)`` returns a tensor or a tuple of tensors; ``inputs`` / ``grad_outputs`` are example tensors of the step's
    shapes (inputs that require grad get one); ``params``: the parameters whose ``.grad`` the step produces.
    ``outs, input_grads = step(inputs, grad_outputs)`` copies the arguments into the graph's static buffers, replays, and
    returns the graph-owned results (overwritten by the next replay); parameter gradients are WRITTEN (not accumulated)
    into ``p.grad`` -- tensors the graph owns (re-attached after every replay, so ``zero_grad(set_to_none=True)`` between
    steps is harmless): an optimiser reads them as usual; gradient ACCUMULATION over several replays is the caller's to do.

    What it needs from ``fn``: fixed shapes, no host synchronisation (the ops and modules here have none), no random
    numbers.  Gradient checkpointing inside ``fn`` works (the blocks' checkpoints do not save RNG state) but is better
    switched off: a replayed step is bound by its kernels, and the recompute pass is kernels."""

    def __init__(self, fn, inputs, grad_outputs, params, warmup=3):
        self.fn = fn
        self.params = [p for p in params if p.requires_grad]
        dev = inputs[0].device
        self._in = [x.detach().clone().requires_grad_(x.requires_grad) for x in inputs]
        self._go = [g.detach().clone() for g in grad_outputs]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._run()
                self._clear()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._out = self._run()
        self._in_grads = [x.grad for x in self._in]
        self._param_grads = [p.grad for p in self.params]      # (the buffers the recorded backward writes)

    def _run(self):
        outs = self.fn(*self._in)
        tup = outs if isinstance(outs, (tuple, list)) else (outs,)
        torch.autograd.backward(list(tup), self._go)
        return outs

    def _clear(self):
        for x in self._in:
            x.grad = None
        for p in self.params:
            p.grad = None

    def __call__(self, inputs, grad_outputs):
        with torch.no_grad():
            torch._foreach_copy_(self._in, [x.detach() for x in inputs])
            torch._foreach_copy_(self._go, list(grad_outputs))
        self.graph.replay()
        for p, g in zip(self.params, self._param_grads):       # (a caller may have detached them: zero_grad(set_to_none=True))
            p.grad = g
        return self._out, self._in_grads
