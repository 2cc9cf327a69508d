"""hipGraph capture of the train step's forward + backward (launch-bound inner loop -> one graph replay).

The production step issues ~5 400 kernels; enqueueing them from Python costs about as much wall time as the GPU needs to
run them, so the step is host-bound in eager mode.  ``GraphedTrainStep`` captures loss forward + backward (including the
bucketed landing of the gradients in the optimizer's flat buffer) once into a ``torch.cuda.CUDAGraph`` and replays it; the
optimizer update (global-norm reduction + fused clip/Adam, 2 launches) and, under data parallelism, the bucketed all-reduce
stay eager after the replay.  Everything captured runs on hand-written HIP kernels / hipBLASLt exactly as in eager mode --
the graph only removes launch overhead.

What the capture needs and how it gets it:
  * static inputs: ``video`` / ``mask`` are copied into fixed buffers before each replay;
  * stochastic ops: ``Rngs.draw`` is pointed at fixed noise buffers (its injection hook) that are refilled by the default
    CUDA generator before each replay, so every step still sees fresh noise of the right distribution;
  * no collective inside the graph: with a ``GradReducer`` attached the all-reduce of the flat gradient buffer is issued
    after the replay (bucketed, async), i.e. graph mode trades the backward/all-reduce overlap of eager mode for zero
    launch overhead; eager mode (``loss.train_step``) keeps the overlap.
"""
import gc

import torch

from . import loss as L
from . import ops


class GraphedTrainStep:
    def __init__(self, model, optimizer, video, mask, hparams, hw, rngs, warmup=3):
        self.model, self.opt, self.hparams, self.hw, self.rngs = model, optimizer, hparams, hw, rngs
        self.video = video.clone()
        self.mask = mask.clone()
        self.rl = L._is_rl(model)
        self.noise = {}
        self.graph = None
        self._capture(warmup)

    def _fwd_bwd(self):
        emask = L.expand_mask(self.mask, self.hw)
        if self.rl:
            loss, aux = L.loss_fn(self.model, self.video, emask, self.mask, self.rngs, self.hparams)
        else:
            loss, aux = L.loss_fn_plain(self.model, self.video, emask, self.mask, self.rngs, self.hparams)
        # torch.autograd.grad instead of .backward(): no AccumulateGrad nodes take part, so nothing created on another stream
        # (e.g. by the optimizer's hooks at construction time) can leak a cross-stream dependency into the capture
        self.opt.external = set()
        self.opt.hooks_active = False                    # gradients arrive through land_all below, not through the hooks
        with ops.deferred_wgrad(self.opt):               # dense weight gradients: parked, then grouped launches into the flat buffer
            grads = torch.autograd.grad(loss, self.opt.params, allow_unused=True)
        self.opt.land_all(grads)
        return loss.detach(), {k: v.detach() for k, v in aux.items() if k != "reconstruction"}

    def _refill(self):
        for name, (kind, buf) in self.noise.items():
            buf.normal_() if kind == "normal" else buf.uniform_()

    def _capture(self, warmup):
        opt = self.opt
        # Discovery, warm-up and capture all run on ONE dedicated stream.  Autograd pins each parameter's AccumulateGrad node
        # to the stream that was current when the node was created; a node left over from an earlier pass on another stream
        # makes the engine insert cross-stream syncs, which corrupts a capture -- so no pass before the capture may run on a
        # different stream, and no autograd graph from an earlier pass may still be alive (gc below).
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        gc.collect()
        with torch.cuda.stream(self.stream):
            # 1. discover the stochastic draws of one step and pin them to static buffers
            self.rngs.recording = {}
            opt.zero_grad()
            self._fwd_bwd()
            opt.update()
            for name, (kind, shape, dtype) in self.rngs.recording.items():
                buf = torch.empty(shape, dtype=dtype, device=self.video.device)
                self.noise[name] = (kind, buf)
                self.rngs.inject(name, buf)
            self.rngs.recording = None
            # 2. warm up (allocator, hipBLASLt heuristics, one-time kernel attributes)
            opt.defer_reduce = True
            for _ in range(warmup):
                self._refill()
                opt.zero_grad()
                self._fwd_bwd()
                opt.update()
            self._refill()
            opt.zero_grad()
        torch.cuda.synchronize()
        gc.collect()
        # 3. capture forward + backward + gradient landing on the same stream
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            self.loss, self.aux = self._fwd_bwd()
        if not all(opt.landed):
            raise RuntimeError("a gradient bucket did not land inside the captured backward (parameter without gradient)")
        self.graph = g
        opt.update()                      # the captured pass produced real gradients: apply them

    def __call__(self, video=None, mask=None):
        if video is not None:
            self.video.copy_(video)
        if mask is not None:
            self.mask.copy_(mask)
        self._refill()
        self.graph.replay()
        self.opt.update()
        return self.loss, self.aux
