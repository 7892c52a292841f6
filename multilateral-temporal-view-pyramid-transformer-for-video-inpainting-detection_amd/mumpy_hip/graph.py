"""hipGraph capture of the whole forward (encoder + decoder): ~700 kernel launches become one graph launch, which
removes the Python/ctypes launch overhead that would otherwise bound small-batch throughput.  Every kernel in
libmumpy_hip.so is capture-safe by construction (no allocation, no sync, explicit stream)."""
import torch


class GraphedForward:
    def __init__(self, encoder, decoder, example: torch.Tensor, warmup: int = 2, with_mask: bool = False):
        """with_mask=True captures Decoder.predict_mask: outputs are (logits, uint8 mask, feats), the thresholded mask
        of test.py:100-108 coming out of the same last kernel."""
        self.encoder, self.decoder, self.with_mask = encoder, decoder, with_mask
        self.static_x = example.clone()
        self._warmup = warmup
        # load_state_dict rewrites parameters through torch (tensor versions move, the epoch does not): count it as a
        # weights change too, so that a graph captured before it is re-captured
        from .state import bump_weights_epoch
        for m in (encoder, decoder):
            m.register_load_state_dict_post_hook(lambda module, incompatible: bump_weights_epoch())
        self._capture()

    def _capture(self):
        """The graph bakes in the addresses of the weights AND of the tensors derived from them (transposed tokenizer
        weights, padded relative-position bias, concatenated k|v weights, KRSC convolution images -- models.modules.layers
        .Derived).  An optimizer step or load_state_dict makes the eager path rebuild those; replaying an old graph would
        then read freed or recycled memory.  So the capture remembers the weights epoch and __call__ re-captures when it
        has moved (validation between training epochs keeps working; steady-state inference never pays for it)."""
        from . import state
        # warm-up and capture run on the SAME side stream: the per-stream kept workspaces of the GEMMs (ops._kept_workspace)
        # and the fork/join side streams (keyed by their parent) that the warm-up created are then the ones the captured
        # launches use -- captured on another stream, every GEMM would record a zero-fill of a fresh 17 MB workspace
        # (0.86 ms per forward of fill kernels in the first round-2 profile)
        if getattr(self, "_side", None) is None:
            from .streams import new_distinct_stream       # never a handle that a fork/join side stream already wraps
            self._side = new_distinct_stream(self.static_x.device, (torch.cuda.current_stream().cuda_stream,))
        side = self._side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():          # warm the derived-table caches off the graph
            for _ in range(self._warmup):
                self._fwd()
        torch.cuda.current_stream().wait_stream(side)
        from .streams import quiesce_collectives
        quiesce_collectives()                                # device idle; RCCL's watchdog has nothing left to poll (streams.py)
        self.graph = torch.cuda.CUDAGraph()
        # capture_error_mode="thread_local": with a process group alive (one rank per GPU) RCCL's watchdog thread polls events
        # while this thread captures; in the default "global" mode such a call from ANOTHER thread invalidates the capture
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.static_out = self._fwd()
        self.weights_epoch = state.weights_epoch[0]

    def _fwd(self):
        from .pipeline import fused_forward
        return fused_forward(self.encoder, self.decoder, self.static_x, with_mask=self.with_mask)

    def __call__(self, x: torch.Tensor):
        """Returns the static (logits, feats) buffers; contents are overwritten by the next call."""
        from . import state
        if state.weights_epoch[0] != self.weights_epoch:           # weights (hence derived tensors) changed since capture
            self._capture()
        self.static_x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out
