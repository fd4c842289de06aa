"""HIP-graph replay of the hot path for fixed shapes.

One pass of pixel decoder + decoder + post-processing is ~140 launches of 5-200 us each; issued eagerly from Python the
host cannot keep a single MI355X busy (launch-bound below batch ~4).  The path has no data-dependent host control flow,
so it is captured once per input geometry into a HIP graph and replayed: inputs are copied into the graph's static
buffers, outputs are the graph's static tensors.  (The reference relies on eager PyTorch; a tracing compiler is
deliberately not used -- explicit kernels + explicit graphs.)
"""
import os
import warnings

import torch

# attributes under which the modules keep derived tensors (packed weights, folded constants, position codes, broadcast
# queries) that the kernels of a forward read by address
_CACHE_ATTRS = ("_q0", "_kv_cache", "_fold_cache", "_tails_cache", "_pos_cache", "_cache", "_packed", "_front", "_w3_cache", "_wl_cache", "_iota", "_heads0_cache",
                "_packed_mf", "_bf16_cache", "_folded_cache", "_conv_fold_cache",
                "_folded", "_lp", "_plan_cache")        # the backbones' folded / 16-bit weight copies (ucn_backbone.py, resnet_backbone.py)


from ._plan import PLAN_ATTRS as _PLAN_ATTRS, TensorList, plan_epoch


def cache_refs(model):
    """Strong references to every derived-tensor cache entry the model holds right now.  A captured HIP graph bakes the
    device addresses of these tensors into its nodes; the modules may later replace or evict the entries (another batch
    size, more input geometries than a cache keeps, a parameter update), so a graph keeps what it was captured with alive
    for as long as it can be replayed."""
    refs = []
    for m in model.modules():
        for a in _CACHE_ATTRS:
            v = m.__dict__.get(a)
            if v is not None:
                refs.append(dict(v) if isinstance(v, dict) else v)      # a dict is copied: eviction edits it in place
    return refs


def param_signature(model):
    """The exhaustive signature (every tensor's address and version, every plan attribute of every module): ~1 ms of
    Python for the head.  Used at capture time and by ``strict=True`` replays; the per-replay check is StaleCheck."""
    sig = tuple((t.data_ptr(), t._version) for t in list(model.parameters()) + list(model.buffers()))
    plan = tuple(m.__dict__.get(a) for m in model.modules() for a in sorted(_PLAN_ATTRS) if a in m.__dict__)
    return sig + plan


class StaleCheck:
    """Cheap per-replay staleness test of a captured graph (the exhaustive ``param_signature`` cost 0.75-1.3 ms per call,
    a third of the 2.2 ms step it guards).  The signature is

        (plan epoch, number of tensors, sum of the tensors' version counters, sum of the tensors' addresses)

    * the plan epoch (``_plan.py``) moves when any plan attribute of a module is assigned a new value or a library option
      is set: one integer compare instead of modules x attributes dictionary probes;
    * in-place updates (``load_state_dict``, an optimizer step, ``.copy_``) bump ``_version`` of the tensor they touch;
    * ``.to(device)`` / ``.half()`` / ``p.data = new`` move tensors: the address sum changes;
    * a Parameter OBJECT replaced by hand (``m.weight = nn.Parameter(...)``, ``load_state_dict(assign=True)``, parametrize)
      goes through ``Module.register_parameter``: the parameter epoch moves and the tensor list is rebuilt (``_plan.TensorList``).
    ~60 us for the head's 298 tensors."""

    def __init__(self, model, strict=False):
        self.model = model
        self.strict = strict
        self._tensors = TensorList.of(model, buffers=True)
        self._manual = 0

    def invalidate(self):
        self._manual += 1
        self._tensors.clear()

    def __call__(self):
        if self.strict:
            return param_signature(self.model)
        ts = self._tensors()
        return (plan_epoch(), len(ts), sum([t._version for t in ts]), sum([t.data_ptr() for t in ts]), self._manual)


class GraphedInference:
    """``GraphedInference(model)(features, image_size)`` == ``model.inference(features, image_size)`` (meta_arch.py),
    replayed from a HIP graph.  ``features``: dict of device tensors.  The returned tensors are owned by the graph and
    are overwritten by the next call with the same geometry: ``.clone()`` what must outlive it."""

    def __init__(self, model, warmup=2, strict=False, entry="inference"):
        self.model = model
        self.entry = entry             # the model method that is captured: "inference" (features in) or "inference_images" (backbone included)
        self.warmup = max(1, int(warmup))
        self._graphs = {}
        self._stream = None
        self._sig = StaleCheck(model, strict)

    def invalidate(self):
        """Force a re-capture on the next call (after replacing Parameter objects by hand; see StaleCheck)."""
        self._sig.invalidate()

    def _key(self, features, image_size, padded_size):
        return (tuple((k, tuple(v.shape), v.dtype, v.device) for k, v in sorted(features.items())), tuple(image_size),
                tuple(padded_size or image_size))

    @torch.no_grad()
    def __call__(self, features, image_size, padded_size=None):
        for v in features.values():
            if not v.is_cuda:
                raise RuntimeError("GraphedInference needs device tensors (there is no CPU path)")
        key = self._key(features, image_size, padded_size)
        sig = self._sig()
        entry = self._graphs.get(key)
        if entry is not None and entry[3] != sig:          # parameters changed since the capture
            entry = None
        if entry is None:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=next(iter(features.values())).device)
            static_in = {k: v.clone() for k, v in features.items()}
            cur = torch.cuda.current_stream()
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                run = getattr(self.model, self.entry)
                for _ in range(self.warmup):                       # builds every weight cache outside the capture
                    run(static_in, image_size, padded_size)
                self._stream.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=self._stream):
                    static_out = run(static_in, image_size, padded_size)
            cur.wait_stream(self._stream)
            entry = (graph, static_in, static_out, sig, cache_refs(self.model))
            self._graphs[key] = entry
        graph, static_in, static_out = entry[:3]
        for k, v in features.items():
            static_in[k].copy_(v)
        graph.replay()
        return static_out


# Slot i of every PipelinedInference on a device replays on the SAME stream: the HIP runtime hands hardware queues to streams in
# creation order, so a process that builds one pipeline after another (another precision, another batch shape) would otherwise put
# two slots of the later pipeline on one queue, where they serialise (measured: 2.15 -> 2.35 ms per batch at depth 4 after ten
# streams had been created).
_SLOT_STREAMS = {}


def _slot_stream(device, i):
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), int(i))
    if key not in _SLOT_STREAMS:
        _SLOT_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SLOT_STREAMS[key]


class PipelinedInference:
    """Throughput mode: ``depth`` batches in flight, each replayed from its own HIP graph on its own stream.

    One pass of the hot path is a chain of ~140 dependent launches, and a fifth of its time goes to kernels that cannot fill
    the chip on their own -- the decoder's row-local chains own 50 16-row tiles (50 of 256 CUs), the small-level attention
    and the top-k a few dozen workgroups.  Nothing orders the passes of DIFFERENT batches, so a second and third batch on
    their own streams fill those gaps (640x480, batch 8: 2.88 ms per batch alone, 2.45 with two in flight, 2.29 with four;
    bench.py --inflight N).  Every slot owns its graph, its input buffers and its outputs; weights and the derived
    weight caches are shared and read-only.

        pipe = model.pipelined(depth=3)
        h = pipe.submit(features, image_size)      # copies `features` into the slot's input buffers, replays its graph
        ...                                        # submit more batches; up to `depth` overlap on the GPU
        out = pipe.result(h)                       # waits for that batch; the tensors are the slot's own: consume (or
                                                   # .clone()) them before the slot comes round again (depth submits later)

    ``submit(None, image_size, slot_inputs=True)`` re-runs a slot on whatever its input buffers hold -- a producer (the
    backbone) can write its outputs straight into ``pipe.inputs(slot)`` instead of paying the copy (after ``result(slot)``
    of the slot's previous batch: its graph reads those buffers until then).
    """

    def __init__(self, model, depth=2, warmup=2, strict=False, entry="inference"):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.model = model
        self.entry = entry             # "inference" (features in) or "inference_images" (backbone in the slot's graph, as GraphedInference)
        self.depth = int(depth)
        self.warmup = max(1, int(warmup))
        self._sig = StaleCheck(model, strict)
        # the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; streams that share a
        # queue serialise.  The variable is read when the runtime initialises, so it can only be checked here.
        queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        if self.depth + 1 > queues:
            warnings.warn(f"PipelinedInference(depth={self.depth}): GPU_MAX_HW_QUEUES={queues} hardware queues; export "
                          f"GPU_MAX_HW_QUEUES>={self.depth + 2} before the first HIP call or slots will share a queue and "
                          "serialise", RuntimeWarning)
        self._slots = [None] * self.depth      # (key, stream, graph, static_in, static_out, done_event)
        self._next = 0

    @staticmethod
    def _key(features, image_size, padded_size):
        return (tuple((k, tuple(v.shape), v.dtype, v.device) for k, v in sorted(features.items())), tuple(image_size),
                tuple(padded_size or image_size))

    def _build(self, i, features, image_size, padded_size):
        old = self._slots[i]
        stream = old[1] if old is not None else _slot_stream(next(iter(features.values())).device, i)
        cur = torch.cuda.current_stream()
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            static_in = {k: v.clone() for k, v in features.items()}
            run = getattr(self.model, self.entry)
            for _ in range(self.warmup):                           # builds every weight cache outside the capture
                run(static_in, image_size, padded_size)
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                static_out = run(static_in, image_size, padded_size)
        self._slots[i] = (self._key(features, image_size, padded_size), stream, graph, static_in, static_out,
                          torch.cuda.Event(), self._sig(), cache_refs(self.model))
        return self._slots[i]

    def invalidate(self):
        """Force every slot to re-capture on its next submit (see StaleCheck)."""
        self._sig.invalidate()

    def inputs(self, slot):
        """The input buffers of a slot (dict of device tensors) once it has been built by a first ``submit``."""
        if self._slots[slot] is None:
            raise RuntimeError("slot %d has not been used yet" % slot)
        return self._slots[slot][3]

    @torch.no_grad()
    def submit(self, features, image_size, padded_size=None, slot_inputs=False):
        """Queue one batch; returns the slot handle for ``result``.  Slots are taken round-robin."""
        i = self._next
        entry = self._slots[i]
        if slot_inputs:
            if entry is None:
                raise RuntimeError("slot_inputs=True needs a slot that was built by an earlier submit")
        else:
            for v in features.values():
                if not v.is_cuda:
                    raise RuntimeError("PipelinedInference needs device tensors (there is no CPU path)")
            if entry is None or entry[0] != self._key(features, image_size, padded_size) or entry[6] != self._sig():
                entry = self._build(i, features, image_size, padded_size)
        self._next = (i + 1) % self.depth
        stream, graph, static_in, done = entry[1], entry[2], entry[3], entry[5]
        stream.wait_stream(torch.cuda.current_stream())            # the producer of the inputs runs on the caller's stream
        with torch.cuda.stream(stream):
            if not slot_inputs:
                for k, v in features.items():
                    v.record_stream(stream)         # the caller may free `features` right after submit(): the allocator must
                    static_in[k].copy_(v, non_blocking=True)        # not reuse that memory before this side-stream copy ran
            graph.replay()
            done.record(stream)
        return i

    def result(self, slot, wait="stream"):
        """Outputs of the batch last submitted to ``slot``.  wait="stream": the caller's current stream waits for the batch
        (no host block; what follows on that stream sees the results); wait="host": the host blocks until it is done."""
        entry = self._slots[slot]
        if entry is None:
            raise RuntimeError("slot %d has not been used yet" % slot)
        if wait == "host":
            entry[5].synchronize()
        else:
            torch.cuda.current_stream().wait_event(entry[5])
        return entry[4]

    def drain(self):
        """Host-block until every slot is idle."""
        for e in self._slots:
            if e is not None:
                e[5].synchronize()
