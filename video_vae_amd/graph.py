"""hipGraph capture of the train step's forward + backward (launch-bound inner loop -> one graph replay).

The production step issues ~800 kernels (796 graph nodes at the end of round 3; eager mode adds the framework's own small launches);
enqueueing them from Python costs more wall time than the GPU needs to run them, so the step is host-bound in eager mode.  ``GraphedTrainStep`` captures loss forward + backward (including the
bucketed landing of the gradients in the optimizer's flat buffer) once into a ``torch.cuda.CUDAGraph`` and replays it; the
optimizer update (global-norm reduction + fused clip/Adam, 2 launches) and, under data parallelism, the bucketed all-reduce
stay eager after the replay.  Everything captured runs on hand-written HIP kernels / hipBLASLt exactly as in eager mode --
the graph only removes launch overhead.

What the capture needs and how it gets it:
  * static inputs: ``video`` / ``mask`` are copied into fixed buffers before each replay;
  * stochastic ops: ``Rngs.draw`` is pointed at fixed noise buffers (its injection hook) that are refilled before each replay
    from a device generator seeded from ``rngs.seed`` (per-rank under data parallelism: the driver builds ``Rngs(3 + rank)``),
    so every step still sees fresh noise of the right distribution and ranks draw different noise, as in eager mode;
  * no collective inside a graph: with a ``GradReducer`` attached the step is captured as 1 + ``enc_segments`` graphs: (forward +
    everything downstream of the encoder's last block) and the encoder's backward in runs of blocks.  The buckets a stage completed
    are handed to the reducer before the next replay, so their all-reduce (RCCL's own stream) runs under the stages still to come;
    only the last segment's buckets (a third of the encoder: ~95 of the 683 MB) wait for the end.  Without a reducer one graph
    holds the whole pass.
  * streams: capture happens on a private stream (autograd's stream bookkeeping needs one that nothing else has used), but the
    graphs are REPLAYED on the caller's current stream, on purpose: the input copies, the noise refill, the eager clip+Adam and --
    the part that matters next to RCCL -- the collectives all order themselves against the current stream (ProcessGroupNCCL makes
    its internal stream wait on an event recorded on the current stream when ``all_reduce`` is called, and ``work.wait()`` makes
    the current stream wait for the collective).  A replay on the capture stream would need hand-written event edges to and
    from RCCL's stream for every bucket; on the current stream the framework's own edges are the correct ones.
  * discovery, warm-up and capture run real optimizer updates (they exercise the eager half of the step); the parameters, Adam
    moments and update count are snapshotted before and restored after, so constructing a GraphedTrainStep does not advance
    training: step 0 of the run starts from the weights the caller built (or resumed) and schedule(count) is unchanged.
"""
import ctypes
import gc

import torch

from . import loss as L
from . import ops


_NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}
_HIP = [None]


def graph_node_census(graph):
    """{node type: count} of a captured ``torch.cuda.CUDAGraph(keep_graph=True)`` (hipGraphGetNodes / hipGraphNodeGetType), or None when
    this build cannot hand out the raw graph.  Exists for ONE check: a captured step must hold no MEMSET node -- on ROCm 7.2 a replayed
    memset node fills its buffer with garbage (the node's element count, the low word of an address: tools/memset_node_probe.py), which is
    how the framework's multi-block reductions (semaphores zeroed by hipMemsetAsync) returned stale results inside the replayed step."""
    raw = getattr(graph, "raw_cuda_graph", None)
    if raw is None:
        return None
    try:
        handle = raw()
        if _HIP[0] is None:
            _HIP[0] = ctypes.CDLL("libamdhip64.so")
        hip = _HIP[0]
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(ctypes.c_void_p(handle), None, ctypes.byref(n)) != 0:
            return None
        nodes = (ctypes.c_void_p * max(1, n.value))()
        if hip.hipGraphGetNodes(ctypes.c_void_p(handle), nodes, ctypes.byref(n)) != 0:
            return None
        census = {}
        for i in range(n.value):
            t = ctypes.c_int(-1)
            if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) != 0:
                return None
            name = _NODE_TYPES.get(t.value, f"type{t.value}")
            census[name] = census.get(name, 0) + 1
        return census
    except Exception:
        return None


class GraphedTrainStep:
    def __init__(self, model, optimizer, video, mask, hparams, hw, rngs, warmup=3, split=None, enc_segments=3, debug_dot=None,
                 perceptual_loss_fn=None, vgg_params=None, stream=None):
        self.model, self.opt, self.hparams, self.hw, self.rngs = model, optimizer, hparams, hw, rngs
        self.ploss, self.vgg_params = perceptual_loss_fn, vgg_params      # rl flavour only (rl_nonadversarial.py:125)
        self.capture_stream = stream        # capture on this (non-default) stream instead of a fresh one: see _capture
        self.split = (optimizer.reducer is not None) if split is None else bool(split)
        enc = model.encoder
        if not (hasattr(enc, "layers") and hasattr(enc, "patch_embedding") and len(enc.layers) > 0):
            self.split = False
        self.graphs = []
        if self.split:
            # The cuts: the output of the encoder's last FactoredAttention block (ONE tensor; the mean / variance / selection heads
            # behind it depend on each other and stay with stage 0), then the outputs of earlier blocks so that the encoder's
            # backward is ``enc_segments`` graphs of about equal depth.  Stage 0 = forward + everything downstream of the last block;
            # stage j >= 1 owns a run of encoder blocks (the last one also the patch embedding) -- runs of the flat buffer, which is
            # laid out in reverse registration order, so the buckets complete stage by stage.
            n = len(enc.layers)
            k = max(1, min(int(enc_segments), n))
            bounds = [round(n * j / k) for j in range(k + 1)]            # segment j (from the top) = layers [bounds[k-j-1], bounds[k-j])
            self.cut_modules = [enc.layers[bounds[k - j] - 1] for j in range(k)]      # cut j feeds stage j + 1; cut 0 = last block
            stage_of = {}
            for j in range(k):
                for li in range(bounds[k - j - 1], bounds[k - j]):
                    for prm in enc.layers[li].parameters():
                        stage_of[id(prm)] = j + 1
            for prm in enc.patch_embedding.parameters():
                stage_of[id(prm)] = k
            self.nstages = k + 1
            self.stage_idx = [[i for i, prm in enumerate(optimizer.params) if stage_of.get(id(prm), 0) == st] for st in range(k + 1)]
            pstage = [stage_of.get(id(prm), 0) for prm in optimizer.params]
            self.bucket_stage = [max(pstage[i] for i in members) for members in optimizer.bucket_params]
        self.video = video.clone()
        self.mask = mask.clone()
        if video.is_cuda:
            ops.unit_grad(video)                 # the constant root gradient exists before anything is captured (never allocated from a graph's pool)
        self.rl = L._is_rl(model)
        self.noise = {}
        self.graph = None
        self.debug_dot = debug_dot           # path: write the captured graph (hipGraphDebugDotPrint) there, for tools/
        self._capture(warmup)

    @staticmethod
    def _unit(loss):
        """The root gradient as a constant tensor (ops.unit_grad) where the loss is an fp32 scalar: no ones_like fill inside the graph."""
        return ops.unit_grad(loss) if (loss.dtype == torch.float32 and loss.dim() == 0) else None

    def _loss(self):
        emask = L.compact_mask(self.mask)                # a view: the (b*hw, 1, 1, t) expansion would be two launches of every replay
        if self.rl:
            return L.loss_fn(self.model, self.video, emask, self.mask, self.rngs, self.hparams, self.ploss, self.vgg_params)
        return L.loss_fn_plain(self.model, self.video, emask, self.mask, self.rngs, self.hparams)

    # ---- staged form (data parallel): forward + decoder backward | encoder backward in segments -------------------------
    @staticmethod
    def _tensors(out):
        """The tensors a block handed on: one tensor, or the (sum, None) / (skip, branch) pair of the pending protocol."""
        outs = out if isinstance(out, (tuple, list)) else (out,)
        return [o for o in outs if isinstance(o, torch.Tensor) and o.requires_grad]

    def _stage0(self):
        """Forward + backward of everything downstream of the encoder's last block: those gradients land, the gradients of the cut
        tensors are kept for the next stage."""
        grabbed = {}
        hooks = [m.register_forward_hook(lambda mod, inp, out, j=j: grabbed.setdefault(j, []).append(out)) for j, m in enumerate(self.cut_modules)]
        try:
            loss, aux = self._loss()
        finally:
            for h in hooks:
                h.remove()
        self._cuts = []
        for j in range(len(self.cut_modules)):
            if len(grabbed.get(j, [])) != 1 or not self._tensors(grabbed[j][0]):
                raise RuntimeError("split capture expects every cut block of the encoder to run once and hand on differentiable tensors")
            self._cuts.append(self._tensors(grabbed[j][0]))
        opt = self.opt
        opt.external = set()
        opt.hooks_active = False
        params = [opt.params[i] for i in self.stage_idx[0]]
        with ops.deferred_wgrad(opt):
            grads = torch.autograd.grad(loss, params + self._cuts[0], grad_outputs=self._unit(loss), allow_unused=True)
        self._gcut = list(grads[len(params):])
        opt.land_subset(self.stage_idx[0], grads[:len(params)])
        return loss.detach(), {k: v.detach() for k, v in aux.items() if k != "reconstruction"}

    def _stage(self, st):
        """Backward of encoder segment ``st`` (>= 1) from the gradients kept at cut st - 1; its slice of the flat buffer lands."""
        opt = self.opt
        pairs = [(c, g) for c, g in zip(self._cuts[st - 1], self._gcut) if g is not None]
        params = [opt.params[i] for i in self.stage_idx[st]]
        nxt = self._cuts[st] if st < len(self._cuts) else []
        with ops.deferred_wgrad(opt):
            grads = torch.autograd.grad([c for c, _ in pairs], params + nxt, grad_outputs=[g for _, g in pairs], allow_unused=True)
        self._gcut = list(grads[len(params):])
        opt.land_subset(self.stage_idx[st], grads[:len(params)])
        if st == self.nstages - 1:
            opt.landed = [True] * len(opt.buckets)
            self._cuts = self._gcut = None

    def _prelaunch(self, st):
        """Hand the buckets that stage ``st`` completed to the reducer: their all-reduce overlaps the stages still to run."""
        opt = self.opt
        if opt.reducer is not None and opt.defer_reduce:
            if st == 0:
                opt.reducer.reset()
            for b, bs in enumerate(self.bucket_stage):
                if bs == st:
                    opt.reducer.launch(b)
                    opt.prelaunched.add(b)

    def _pass(self):
        if not self.split:
            return self._fwd_bwd()
        out = self._stage0()
        for st in range(1, self.nstages):
            self._prelaunch(st - 1)
            self._stage(st)
        return out

    def _fwd_bwd(self):
        loss, aux = self._loss()
        # torch.autograd.grad instead of .backward(): no AccumulateGrad nodes take part, so nothing created on another stream
        # (e.g. by the optimizer's hooks at construction time) can leak a cross-stream dependency into the capture
        self.opt.external = set()
        self.opt.hooks_active = False                    # gradients arrive through land_all below, not through the hooks
        with ops.deferred_wgrad(self.opt):               # dense weight gradients: parked, then grouped launches into the flat buffer
            grads = torch.autograd.grad(loss, self.opt.params, grad_outputs=self._unit(loss), allow_unused=True)
        self.opt.land_all(grads)
        return loss.detach(), {k: v.detach() for k, v in aux.items() if k != "reconstruction"}

    def _check_one_stream(self):
        """Refuse a capture the autograd engine would crash in.  An eager backward leaves every parameter's AccumulateGrad node pinned to
        the stream it ran on for as long as anything keeps that pass's graph alive (a kept ``loss`` is enough); Optimizer's hooks wrote
        that stream onto the node.  Capturing on another stream makes the engine insert a cross-stream sync inside the capture: a
        segmentation fault on ROCm 7.2 (round 3: tools/r03_second.sh, `python -m video_vae_amd.train`), not an exception -- so raise one."""
        from .optim import STREAM_TAG, accumulate_grad_node
        from ._lib import VvaeError
        if not self.video.is_cuda:
            return
        want = self.stream.cuda_stream
        with torch.cuda.stream(self.stream):            # nodes that do not exist yet are born on the capture stream, where they belong
            for name, prm in zip(self.opt.names, self.opt.params):
                node = accumulate_grad_node(prm)
                have = node.metadata.get(STREAM_TAG) if node is not None else None
                if have is not None and have != want:
                    raise VvaeError(
                        f"GraphedTrainStep: parameter {name!r} has already taken part in an eager backward on stream {have:#x} and that pass's "
                        f"autograd graph is still alive, but the capture would run on stream {want:#x}.  One stream for the whole run: build the "
                        "model, run the eager steps and capture on the same non-default stream (pass it as stream=..., as train.py does), or drop "
                        "every tensor of the eager passes (loss, aux) before capturing.")

    def _refill(self):
        for name, (kind, buf) in self.noise.items():
            buf.normal_(generator=self.gen) if kind == "normal" else buf.uniform_(generator=self.gen)

    def _capture(self, warmup):
        opt = self.opt
        # Discovery, warm-up and capture all run on ONE dedicated stream.  Autograd pins each parameter's AccumulateGrad node
        # to the stream that was current when the node was created; a node left over from an earlier pass on another stream
        # makes the engine insert cross-stream syncs, which corrupts a capture -- so no pass before the capture may run on a
        # different stream, and no autograd graph from an earlier pass may still be alive (gc below).
        # A driver that has ALREADY run eager steps (train.py: a shape is met once before it is captured) hands over the stream those steps
        # ran on: the AccumulateGrad nodes they created are pinned to it, and a capture on any other stream would segfault in the engine's
        # cross-stream sync.  Such a driver runs everything -- model construction, eager steps, captures, replays -- on that one stream.
        self.stream = self.capture_stream if self.capture_stream is not None else torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        if getattr(self.model, "_kl", None) is not None:
            self.model._kl = None           # the last forward's (mean, log_variance, kl): tensors that keep that pass's autograd graph alive
        gc.collect()
        self._check_one_stream()
        # per-rank noise stream for the static buffers (eager mode draws from Rngs(seed) keys; same distributions here)
        self.gen = torch.Generator(device=self.video.device)
        self.gen.manual_seed((0x9E3779B97F4A7C15 * (self.rngs.seed + 1)) & 0x7FFFFFFFFFFFFFFF)
        # the passes below apply real updates: keep the training state they start from and put it back afterwards
        snap = (opt.p.clone(), opt.m.clone(), opt.v.clone(), opt.count)
        with torch.cuda.stream(self.stream):
            # 1. discover the stochastic draws of one step and pin them to static buffers
            opt.defer_reduce = True           # collectives are issued by _prelaunch / Optimizer.update, never by the landing
            self.rngs.recording = {}
            opt.zero_grad()
            self._pass()
            opt.update()
            for name, (kind, shape, dtype) in self.rngs.recording.items():
                buf = torch.empty(shape, dtype=dtype, device=self.video.device)
                self.noise[name] = (kind, buf)
                self.rngs.inject(name, buf)
            self.rngs.recording = None
            # 2. warm up (allocator, hipBLASLt heuristics, one-time kernel attributes)
            for _ in range(warmup):
                self._refill()
                opt.zero_grad()
                self._pass()
                opt.update()
            self._refill()
            opt.zero_grad()
        torch.cuda.synchronize()
        gc.collect()
        # 3. capture forward + backward + gradient landing on the same stream
        try:
            g = torch.cuda.CUDAGraph(keep_graph=True)        # keeps the hipGraph_t: its nodes are counted below
        except TypeError:
            g = torch.cuda.CUDAGraph()
        if self.debug_dot:
            g.enable_debug_mode()
        # With a process group alive its watchdog thread polls events while we capture: only the capturing thread's own calls
        # may invalidate the capture then ("thread_local"); the default mode would let that poll abort it.
        mode = "thread_local" if opt.reducer is not None else "global"
        if self.split:
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode=mode):
                self.loss, self.aux = self._stage0()
            for st in range(1, self.nstages):
                try:
                    gs = torch.cuda.CUDAGraph(keep_graph=True)
                except TypeError:
                    gs = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gs, stream=self.stream, pool=g.pool(), capture_error_mode=mode):
                    self._stage(st)
                self.graphs.append(gs)
        else:
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode=mode):
                self.loss, self.aux = self._fwd_bwd()
        if not all(opt.landed):
            raise RuntimeError("a gradient bucket did not land inside the captured backward (parameter without gradient)")
        # what was captured: node types per graph.  A memset node must not be there (see graph_node_census): fail loudly, not staly.
        self.census = [graph_node_census(x) for x in [g] + self.graphs]
        for c in self.census:
            if c and c.get("memset", 0):
                raise RuntimeError(f"the captured train step holds {c['memset']} hipGraph memset node(s) ({c}): on ROCm 7.2 a replayed memset node writes "
                                   "garbage, and whatever it was meant to zero (e.g. the semaphore of a multi-block framework reduction) misbehaves on "
                                   "replay.  Replace the op that issued hipMemsetAsync (DESIGN.md section 3, tools/memset_node_probe.py).")
        self.graph = g
        if self.debug_dot:
            g.debug_dump(self.debug_dot)
        opt.update()                      # the captured pass produced real gradients and left buckets to reduce: run the eager half once
        with torch.no_grad():             # ... then put the training state back where the caller left it
            opt.p.copy_(snap[0]); opt.m.copy_(snap[1]); opt.v.copy_(snap[2])
        opt.count = snap[3]
        opt.refresh_shadow()

    def __call__(self, video=None, mask=None, noise=None):
        """One train step.  ``noise``: {draw name: tensor} to use for this step's stochastic draws instead of fresh ones from the
        step's generator (the counterpart of ``Rngs.inject`` for the replayed graph: parity tests hand the oracle's noise over)."""
        if video is not None:
            self.video.copy_(video)
        if mask is not None:
            self.mask.copy_(mask)
        if noise is None:
            self._refill()
        else:
            if set(noise) != set(self.noise):
                raise KeyError(f"noise for {sorted(noise)} given, the captured step draws {sorted(self.noise)}")
            for name, t in noise.items():
                self.noise[name][1].copy_(t)
        self.graph.replay()
        for st, gs in enumerate(self.graphs, start=1):
            self._prelaunch(st - 1)                      # buckets the previous stage completed: reduced under this stage
            gs.replay()
        self.opt.update()
        return self.loss, self.aux
