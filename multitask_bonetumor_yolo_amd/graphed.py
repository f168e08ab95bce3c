"""HIP-graph replay of the whole inference step (forward + decode + NMS + masks).

The launch plan is ~220 short kernels; issued one by one from Python the host becomes visible between
them.  `GraphedInference` captures one complete step -- `model.infer_and_detect` = `model(x, "infer")` +
`postprocess.detect_and_segment`, with decode/NMS forked under the Segment/Proto launches -- into a HIP graph
on a private stream and replays it with a single call.  Inputs and outputs are static buffers: copy the next batch into `.x` (or construct with your own
resident buffer), call `replay()`, read `.out` (overwritten by the next replay).
"""
import torch

from . import postprocess as pp


# Schedule candidates of `autotune`: (plan option, values).  Coordinate search from the defaults: one knob at a time, keep what is faster.
AUTOTUNE_KNOBS = (("NODE_FUSED", ("1",)), ("SEG_GATE", ("1", "2")), ("LANE_WIDE_US", ("200",)), ("LANES", ("3",)), ("ADAPTOR_EARLY", ("1",)),
                  ("HEADS_EARLY", ("1",)))


class GraphedInference:
    def __init__(self, model, x: torch.Tensor, img_size: int, conf_th=pp.CONF_TH, iou_th=pp.NMS_IOU, top_k=pp.TOP_K, masks=True,
                 warmup: int = 2, autotune: bool = False, log=None):
        """`autotune`: before the final capture, time a few launch SCHEDULES of the same plan (lane count, which branches are issued early
        or gated, fused / two-launch BiFPN nodes: `model.plan_option`) as captured graphs -- a dozen replays each -- and keep the fastest
        in `model.plan_options`.  Why: the step is ~190 launches on four streams; which independent chains the GPU happens to run side by
        side moves the replay time by +-4 % (a fused kernel that is 15 % faster alone made the step 0.24 ms SLOWER because the prototype
        chain then started later), nothing of which a static cost model sees.  Results do not depend on the schedule."""
        if autotune:
            import time
            opts = dict(model.__dict__.get("plan_options", {}))

            def timed(o):
                model.plan_options = o
                g = GraphedInference(model, x, img_size, conf_th, iou_th, top_k, masks, warmup=1)
                for _ in range(3):
                    g.replay()
                torch.cuda.synchronize(x.device)
                t0 = time.perf_counter()
                for _ in range(12):
                    g.replay()
                torch.cuda.synchronize(x.device)
                dt = (time.perf_counter() - t0) / 12
                # free this candidate before the next is built: its graph's private pool and the plan's buffer pool (32 GB at
                # batch 64 x 1280^2 -- nine candidates kept alive ran the 288 GB card out of memory, round 3)
                del g
                model.__dict__.get("_plans", {}).clear()
                return dt
            timed(dict(opts))                                 # thrown away: the first candidate of a process reads 2 - 4 % slow (clocks, caches, code
            best = timed(dict(opts))                          # objects): timed once, the defaults lost to whatever came next
            if log:
                log(f"autotune: defaults {best * 1e3:.3f} ms")
            for name, values in AUTOTUNE_KNOBS:
                for v in values:
                    cand = dict(opts, **{name: v})
                    t = timed(cand)
                    if log:
                        log(f"autotune: {name}={v} {t * 1e3:.3f} ms" + (" *" if t < best * 0.995 else ""))
                    if t < best * 0.995:
                        best, opts = t, cand
            model.plan_options = opts
            model.__dict__.get("_plans", {}).clear()          # the losers' plans hold buffer pools
            if log:
                log(f"autotune: kept {opts} ({best * 1e3:.3f} ms)")
        if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("GraphedInference needs a contiguous fp32 CUDA/HIP batch [B,3,S,S] (it is read in place)")
        self.model, self.x = model, x
        self._compiled = None             # the lowered plan the captured kernels point into: pool buffers + folded weights
        self.args = (img_size, conf_th, iou_th, top_k, masks)
        # process-wide capture / side streams, distinct from the plan's lane streams (engine.reserved_stream: why)
        from .engine import reserved_stream
        self.stream = reserved_stream(x.device, "graph_capture")
        self.side = reserved_stream(x.device, "graph_side")   # decode + NMS fork (see _Base.infer_and_detect)
        self.graph = torch.cuda.CUDAGraph()
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(warmup):           # first calls compile the plan and set one-time kernel attributes
                self._step()
            torch.cuda.synchronize(x.device)
            # thread_local: with a process group alive, RCCL's watchdog thread may touch the HIP runtime during the capture
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                self.fwd, self.out = self._step()
            # Keep the plan alive for as long as the graph: the model's plan cache may evict it (weight update, BatchNorm mode flip),
            # which would hand its buffers back to the caching allocator while the captured kernels still read and write them.
            self._compiled = self._eval_compile()
            self._sig = self._compiled.sig
        torch.cuda.current_stream(x.device).wait_stream(self.stream)

    def _eval_compile(self):
        heads = [h for h in (getattr(self.model, "detect", None), self.model.segment) if h is not None]
        flags = [h.training for h in heads]
        try:
            for h in heads:
                h.eval()
            return self.model.compile(self.x)
        finally:
            for h, f in zip(heads, flags):
                h.training = f

    def _step(self):
        img_size, conf_th, iou_th, top_k, masks = self.args
        return self.model.infer_and_detect(self.x, img_size, conf_th, iou_th, top_k, masks, side_stream=self.side,
                                           own_outputs=False)   # static outputs: a replay overwrites them anyway

    def replay(self):
        # The launch goes out FIRST, the staleness check runs while the GPU works: walking ~390 parameters and ~120 BatchNorms costs the host
        # ~2 ms, which in front of the launch is 2 ms of idle GPU whenever the queue is empty (the first replay after a synchronisation: 1.6 %
        # of a 20-step measurement).  A stale replay only rewrites this object's own static outputs, and the caller still gets the error
        # instead of them.
        self.graph.replay()
        if self.model._weights_sig(self._compiled_modes()) != self._sig:
            raise RuntimeError("GraphedInference: the model's weights / BatchNorm statistics changed since the capture (the graph replays "
                               "kernels over the OLD folded weights): build a new GraphedInference")
        return self.out

    def _compiled_modes(self):
        import torch.nn as nn
        heads = {id(m) for h in (getattr(self.model, "detect", None), self.model.segment) if h is not None for m in h.modules()}
        return tuple((False if id(m) in heads else m.training) for m in self.model.modules() if isinstance(m, nn.BatchNorm2d))
