"""HIP-graph replay of a fixed-shape step (MI355X: a launch-bound inner loop belongs in a hipGraph, not behind a tracing compiler).

The train step of the disparity stage issues ~2,800 kernel launches for 8 Config-B crops -- about 20 us of Python + ctypes per launch
against 20 us of GPU work per kernel, so the eager step is bounded by the host.  ``GraphedStep`` runs the step eagerly a few
times (workspaces, plans and kernel attributes get created), captures ONE more run into a HIP graph on PyTorch's capture stream
(every launch of libdisprcnn_hip.so goes to ``torch.cuda.current_stream()``, so the C-ABI kernels are captured like PyTorch's own),
and replays it.  Conditions, as for any captured step: static input tensors (``copy_`` new data into them), fixed shapes (one graph
per ROI-count bucket -- the reference pads/truncates to MAX_ROI_FOR_TRAINING the same way), no host synchronisation inside the step.
Reference for what the step contains: tools/train_net.py + engine/trainer.py (forward, loss, backward, optimizer step).
"""
import torch


class GraphedStep:
    def __init__(self, fn, warmup=3):
        """fn: () -> tensor or tuple of tensors (e.g. the loss); it must read its inputs from tensors that stay alive."""
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs the GPU (HIP graphs); there is no CPU fallback")
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
