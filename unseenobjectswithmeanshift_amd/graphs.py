"""HIP-graph replay of the hot path for fixed shapes.

One pass of pixel decoder + decoder + post-processing is ~140 launches of 5-200 us each; issued eagerly from Python the
host cannot keep a single MI355X busy (launch-bound below batch ~4).  The path has no data-dependent host control flow,
so it is captured once per input geometry into a HIP graph and replayed: inputs are copied into the graph's static
buffers, outputs are the graph's static tensors.  (The reference relies on eager PyTorch; a tracing compiler is
deliberately not used -- explicit kernels + explicit graphs.)
"""
import torch


class GraphedInference:
    """``GraphedInference(model)(features, image_size)`` == ``model.inference(features, image_size)`` (meta_arch.py),
    replayed from a HIP graph.  ``features``: dict of device tensors.  The returned tensors are owned by the graph and
    are overwritten by the next call with the same geometry: ``.clone()`` what must outlive it."""

    def __init__(self, model, warmup=2):
        self.model = model
        self.warmup = max(1, int(warmup))
        self._graphs = {}
        self._stream = None

    def _key(self, features, image_size, padded_size):
        return (tuple((k, tuple(v.shape), v.dtype, v.device) for k, v in sorted(features.items())), tuple(image_size),
                tuple(padded_size or image_size))

    @torch.no_grad()
    def __call__(self, features, image_size, padded_size=None):
        for v in features.values():
            if not v.is_cuda:
                raise RuntimeError("GraphedInference needs device tensors (there is no CPU path)")
        key = self._key(features, image_size, padded_size)
        entry = self._graphs.get(key)
        if entry is None:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=next(iter(features.values())).device)
            static_in = {k: v.clone() for k, v in features.items()}
            cur = torch.cuda.current_stream()
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                for _ in range(self.warmup):                       # builds every weight cache outside the capture
                    self.model.inference(static_in, image_size, padded_size)
                self._stream.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=self._stream):
                    static_out = self.model.inference(static_in, image_size, padded_size)
            cur.wait_stream(self._stream)
            entry = (graph, static_in, static_out)
            self._graphs[key] = entry
        graph, static_in, static_out = entry
        for k, v in features.items():
            static_in[k].copy_(v)
        graph.replay()
        return static_out
