"""HIP-graph replay of the whole inference step (forward + decode + NMS + masks).

The launch plan is ~220 short kernels; issued one by one from Python the host becomes visible between
them.  `GraphedInference` captures one complete step -- `model.infer_and_detect` = `model(x, "infer")` +
`postprocess.detect_and_segment`, with decode/NMS forked under the Segment/Proto launches -- into a HIP graph
on a private stream and replays it with a single call.  Inputs and outputs are static buffers: copy the next batch into `.x` (or construct with your own
resident buffer), call `replay()`, read `.out` (overwritten by the next replay).
"""
import torch

from . import postprocess as pp


class GraphedInference:
    def __init__(self, model, x: torch.Tensor, img_size: int, conf_th=pp.CONF_TH, iou_th=pp.NMS_IOU, top_k=pp.TOP_K, masks=True,
                 warmup: int = 2):
        if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("GraphedInference needs a contiguous fp32 CUDA/HIP batch [B,3,S,S] (it is read in place)")
        self.model, self.x = model, x
        self.args = (img_size, conf_th, iou_th, top_k, masks)
        self.stream = torch.cuda.Stream(device=x.device)
        self.side = torch.cuda.Stream(device=x.device)   # decode + NMS fork (see _Base.infer_and_detect)
        self.graph = torch.cuda.CUDAGraph()
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(warmup):           # first calls compile the plan and set one-time kernel attributes
                self._step()
            torch.cuda.synchronize(x.device)
            # thread_local: with a process group alive, RCCL's watchdog thread may touch the HIP runtime during the capture
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                self.fwd, self.out = self._step()
        torch.cuda.current_stream(x.device).wait_stream(self.stream)

    def _step(self):
        img_size, conf_th, iou_th, top_k, masks = self.args
        return self.model.infer_and_detect(self.x, img_size, conf_th, iou_th, top_k, masks, side_stream=self.side,
                                           own_outputs=False)   # static outputs: a replay overwrites them anyway

    def replay(self):
        self.graph.replay()
        return self.out
