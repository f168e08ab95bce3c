"""Host-side launch engine: NHWC activation views, a static buffer pool and pre-built C-ABI calls.

A forward pass is compiled once per (batch, size, dtype, weights version) into a flat list of
`Launch` records, each a C-ABI function plus its argument block (device pointers already resolved).
Running the plan is a tight loop of ctypes calls on the caller's current HIP stream: no allocation,
no host synchronisation, so it can be captured into a HIP graph.  PyTorch only owns the memory.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch

from . import _lib as L

TORCH_DTYPE = {L.F32: torch.float32, L.BF16: torch.bfloat16, L.F16: torch.float16}
ESIZE = {L.F32: 4, L.BF16: 2, L.F16: 2}


def code_of(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return L.F32
    if dtype == torch.bfloat16:
        return L.BF16
    if dtype == torch.float16:
        return L.F16
    raise ValueError(f"unsupported compute dtype {dtype}: float32 (parity), bfloat16 (throughput) or float16 (inference, BASELINE configs[4])")


@dataclass
class Act:
    """An NHWC activation view: element (n,y,x,c) lives at buf.data_ptr() + (off + n*bs + (y*W+x)*ld + c) * esize.
    ld > C makes it a channel slice of a wider buffer (concat-free C2f); bs lets a pyramid level live
    inside a larger per-image buffer (mask coefficients [N,A,nm])."""
    buf: torch.Tensor
    off: int
    N: int
    H: int
    W: int
    C: int
    ld: int
    bs: int

    @staticmethod
    def of(buf: torch.Tensor) -> "Act":
        N, H, W, Cc = buf.shape
        assert buf.is_contiguous()
        return Act(buf, 0, N, H, W, Cc, Cc, H * W * Cc)

    @property
    def code(self): return code_of(self.buf.dtype)
    @property
    def ptr(self): return self.buf.data_ptr() + self.off * self.buf.element_size()
    @property
    def batch_stride(self): return self.bs
    @property
    def dense(self): return self.ld == self.C and self.bs == self.H * self.W * self.C

    def slice(self, c0, Cc):
        assert 0 <= c0 and c0 + Cc <= self.C
        return Act(self.buf, self.off + c0, self.N, self.H, self.W, Cc, self.ld, self.bs)

    def nchw(self) -> torch.Tensor:
        """Logical [N,C,H,W] view (channels-last strides) of this activation; no copy."""
        return self.buf.as_strided((self.N, self.C, self.H, self.W), (self.bs, 1, self.W * self.ld, self.ld),
                                   self.buf.storage_offset() + self.off)


class Pool:
    """Plan-time buffer pool: buffers freed at plan-build time are handed to later ops of the same
    plan (stream order makes the reuse safe), keeping the working set small and cache-resident."""

    def __init__(self, device):
        self.device = device
        self.free_list = {}
        self.all = []
        self.bytes = 0
        import os
        self.reuse = os.environ.get("MTBT_POOL_REUSE", "1") == "1"  # False: released buffers are never handed out again

    def get(self, shape, dtype) -> torch.Tensor:
        key = (tuple(shape), dtype)
        lst = self.free_list.get(key) if self.reuse else None
        if lst:
            return lst.pop()
        t = torch.empty(shape, dtype=dtype, device=self.device)
        self.all.append(t)
        self.bytes += t.numel() * t.element_size()
        return t

    def put(self, t: torch.Tensor):
        if not self.reuse:
            return
        self.free_list.setdefault((tuple(t.shape), t.dtype), []).append(t)


@dataclass
class Launch:
    fn: Callable
    args: tuple
    name: str
    keep: tuple = ()  # python objects that own the memory the argument block points to
    flops: float = 0.0
    bytes: float = 0.0
    reads: tuple = ()   # memory regions (see _region) this launch reads / writes: the scheduler's dependency source
    writes: tuple = ()
    side: bool = False  # scheduler hint: off the main chain (lane 0) whenever a side lane is allowed -- branch work lowered EARLY in program order


def _region(a):
    """Dependency-tracking key of an activation view (or a plain tensor): (storage address, channel lo, hi, row pitch).
    A channel slice of a wider buffer (ld > C) is tracked by its channel range; everything else as the whole buffer."""
    if isinstance(a, torch.Tensor):
        # a contiguous VIEW into a larger storage (a parameter-gradient slot of a flat bucket): tracked by its element range, so that
        # writers of different slots of one bucket stay independent; everything else as the whole storage
        if a.is_contiguous() and a.numel() * a.element_size() < a.untyped_storage().nbytes():
            off = a.storage_offset()
            return (a.untyped_storage().data_ptr(), off, off + a.numel(), -1)
        return (a.untyped_storage().data_ptr(), 0, 1 << 30, 0)
    key = a.buf.untyped_storage().data_ptr()
    if a.ld != a.C:
        c0 = a.off % a.ld
        return (key, c0, c0 + a.C, a.ld)
    return (key, 0, 1 << 30, 0)


def _overlap(r, s):
    return r[0] == s[0] and (r[3] != s[3] or r[3] == 0 or (r[1] < s[2] and s[1] < r[2]))


def _covers(w, r):
    """True if a write to region w overwrites every byte region r tracks (so older records of r can be dropped)."""
    return w[0] == r[0] and (w[3] == 0 or (w[3] == r[3] and w[1] <= r[1] and r[2] <= w[2]))


class _Schedule:
    pass


_LANE_STREAMS = {}
_RESERVED = {}          # device index -> {role: torch.cuda.Stream}: the process-wide streams of this package, pairwise distinct HIP streams


def reserved_stream(device, role: str) -> torch.cuda.Stream:
    """One process-wide stream per (device, role) whose underlying HIP stream is DISTINCT from every other stream this package reserved.
    torch.cuda.Stream() hands out a pool of 32 streams round-robin: the 33rd object aliases the first.  A long-lived process (a test session:
    dozens of GraphedInference objects, two streams each) therefore ended up with a capture stream or the NMS side stream that WAS one of the
    plan's lane streams -- a fork / join nested below a non-origin stream, which segfaults hipStreamEndCapture on ROCm 7.2 (round 3; tools/
    capture_topo.py has the topology).  Reserved streams are never destroyed: captured graphs may outlive the objects that recorded them."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    table = _RESERVED.setdefault(idx, {})
    s = table.get(role)
    if s is None:
        taken = {t.cuda_stream for t in table.values()}
        for _ in range(256):
            s = torch.cuda.Stream(device=dev)
            if s.cuda_stream not in taken and s.cuda_stream != 0:
                break
        else:
            raise RuntimeError("no distinct HIP stream left in torch's pool")
        table[role] = s
    return s


class Plan:
    def __init__(self, device):
        self.lib = L.load()
        self.device = device
        self.pool = Pool(device)
        self.launches: List[Launch] = []
        self.consts = []  # folded weights etc. (kept alive)
        # development A/B knobs are read ONCE per plan on the host and travel in the argument blocks (the library reads no environment)
        import os
        pol = os.environ.get("MTBT_CONV_POLICY")
        self.conv_policy = (0x100 | (int(pol) & 0xff)) if pol is not None else 0
        if os.environ.get("MTBT_DIRECT_TC64"):
            self.conv_policy = 0x100 | ((self.conv_policy & 0xff) if self.conv_policy else 7) | 32
        self.conv_debug = int(os.environ.get("MTBT_CONV_DEBUG", "0"))
        self.reload_env()

    def reload_env(self):
        """(Re-)read the lane knobs from the environment.  They are read HERE, once per plan, not per step: run() / schedule() are
        per-step host work (an eager training step issues ~950 launches).  Tests and tools that flip a knob on a live plan call this."""
        import os
        self.n_lanes = max(1, int(os.environ.get("MTBT_LANES", "4")))
        self.lane_serial = os.environ.get("MTBT_LANE_SERIAL") == "1"     # dev: total order across lanes (no two launches overlap)
        # launches estimated longer than this fill the machine and stay serialized on lane 0 (round 2: 60 -> 600 us after the kernels got
        # faster: only Proto-class launches stay serialized; 7.13 -> 7.03 ms)
        self.lane_wide_s = float(os.environ.get("MTBT_LANE_WIDE_US", "600")) * 1e-6
        self.lane_window = tuple(int(v) for v in os.environ.get("MTBT_LANE_WINDOW", "0:1000000").split(":"))  # dev: side lanes only in [a, b)
        self.__dict__.pop("_sched", None)
        return self

    # ---- execution ----
    def run(self, stream: Optional[int] = None, start: int = 0, end: Optional[int] = None, marks=None):
        """Issue the plan.  Default: the launches are spread over `lanes()` HIP streams following their data
        dependencies (see schedule()); lane 0 is the caller's current stream, the others fork from it at the start and
        join it at the end, so to the caller the plan still looks like work on its current stream (and a HIP-graph
        capture of that stream records the lanes as parallel branches).  `marks` = launch indices (or a dict of named
        index lists); returns, per list, one event per lane that covers them (recorded after the last marked launch of that
        lane) so that a consumer stream can start as soon as those launches are done.  A sub-range [start, end) or an
        explicit `stream` runs sequentially."""
        groups = marks if isinstance(marks, dict) else ({"_": list(marks)} if marks else {})
        unwrap = (lambda r: r) if isinstance(marks, dict) else (lambda r: r.get("_", []))
        sequential = stream is not None or start != 0 or (end is not None and end != len(self.launches)) or self.lanes() <= 1
        if sequential:
            s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream)
            at = {}
            for name, idx in groups.items():
                if idx:
                    at.setdefault(max(idx), []).append(name)
            out = {name: [] for name in groups}
            for i, l in enumerate(self.launches[start:end], start):
                rc = l.fn(*l.args, s)
                if rc != 0:
                    L.check(rc, l.name)
                for name in at.get(i, ()):
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    out[name].append(ev)
            return unwrap(out)
        sch = self.schedule()
        if sch.events is None:
            sch.events = [torch.cuda.Event() for _ in range(sch.n_events)]
            sch.fork_event = torch.cuda.Event()
            sch.join_events = [torch.cuda.Event() for _ in range(sch.n_lanes - 1)]
        main = torch.cuda.current_stream(self.device)
        streams = [main] + self._side_streams(sch.n_lanes - 1)
        ptrs = [C.c_void_p(st.cuda_stream) for st in streams]
        fork = sch.fork_event
        fork.record(main)
        used = sch.used_side_lanes  # lanes that got no launch stay out of the fork / join (and out of a graph capture)
        for k in used:
            streams[k].wait_event(fork)
        mark_at, out = {}, {name: [] for name in groups}   # launch index -> [(group, event)]
        for name, points in self.mark_points(sch, groups).items():
            for i in points:
                ev = torch.cuda.Event()
                mark_at.setdefault(i, []).append(ev)
                out[name].append(ev)
        launches = self.launches
        serial = self.lane_serial
        prev_ev, prev_ln = None, -1
        for i in sch.order:
            ln = sch.lane[i]
            for e in sch.waits[i]:
                streams[ln].wait_event(sch.events[e])
            if serial and prev_ev is not None and prev_ln != ln:
                streams[ln].wait_event(prev_ev)
            l = launches[i]
            rc = l.fn(*l.args, ptrs[ln])
            if serial:
                prev_ev, prev_ln = torch.cuda.Event(), ln
                prev_ev.record(streams[ln])
            if rc != 0:
                L.check(rc, l.name)
            e = sch.records[i]
            if e >= 0:
                sch.events[e].record(streams[ln])
            for ev in mark_at.get(i, ()):
                ev.record(streams[ln])
        for k in used:
            sch.join_events[k - 1].record(streams[k])
            main.wait_event(sch.join_events[k - 1])
        return unwrap(out)

    def lanes(self) -> int:
        return self.n_lanes

    @staticmethod
    def mark_points(sch, groups):
        """Where run() records the events of each mark group: after the LAST marked launch of EVERY lane that carries one (stream
        order covers that lane's earlier marked launches).  A consumer that waits on all of a group's events is therefore ordered
        behind every launch of the group, whichever lanes the scheduler put them on."""
        out = {}
        for name, idx in groups.items():
            last = {}
            for i in idx:
                last[sch.lane[i]] = max(last.get(sch.lane[i], -1), i)
            out[name] = sorted(last.values())
        return out

    def _side_streams(self, n):
        # process-wide per device and never destroyed: captured graphs may outlive the plan that recorded them
        cur = _LANE_STREAMS.setdefault(torch.device(self.device).index or 0, [])
        while len(cur) < n:
            cur.append(reserved_stream(self.device, f"lane{len(cur) + 1}"))
        return cur[:n]

    def dependencies(self):
        """Per launch, the earlier launches it must wait for: read-after-write, write-after-write and write-after-read on
        the regions the op builders recorded.  Buffers recycled by the pool are tracked by address, so reuse stays safe."""
        hist = {}  # storage address -> (writers [(region, idx)], readers [(region, idx)])
        deps = []
        for i, l in enumerate(self.launches):
            d = set()
            for r in l.reads:
                ws, _ = hist.get(r[0], ((), ()))
                d.update(j for (wr, j) in ws if _overlap(r, wr))
            for w in l.writes:
                ws, rs = hist.setdefault(w[0], ([], []))
                d.update(j for (wr, j) in ws if _overlap(w, wr))
                d.update(j for (rr, j) in rs if _overlap(w, rr))
            for w in l.writes:
                ws, rs = hist[w[0]]
                ws[:] = [(wr, j) for (wr, j) in ws if not _covers(w, wr)] + [(w, i)]
                rs[:] = [(rr, j) for (rr, j) in rs if not _covers(w, rr)]
            for r in l.reads:
                hist.setdefault(r[0], ([], []))[1].append((r, i))
            d.discard(i)
            deps.append(sorted(d))
        return deps

    def schedule(self):
        """List-schedule the launches onto lanes (HIP streams) in program order with a simple duration model: a launch
        starts when its dependencies and its lane are free; it takes the lane where it can start first, preferring the
        lane of the dependency it waits for last (a chain stays on one stream and needs no event).  Cross-lane
        dependencies become event record / wait pairs."""
        n_lanes = self.lanes()
        wide_s = self.lane_wide_s
        c = self.__dict__.get("_sched")
        if c is not None and c.n_launches == len(self.launches) and c.n_lanes == n_lanes and c.wide_s == wide_s:
            return c
        deps = self.dependencies()
        n = len(self.launches)
        win = self.lane_window
        lane, finish = [0] * n, [0.0] * n
        free = [0.0] * n_lanes
        for i, l in enumerate(self.launches):
            dur = max(l.flops / 4e14, l.bytes / 2e12) + 6e-6
            ready, crit = 0.0, -1
            for j in deps[i]:
                if finish[j] >= ready:
                    ready, crit = finish[j], j
            # Side lanes exchange events with lane 0 only: a side stream that waits on another side stream which itself
            # waited on the first (a fork/join nested below a non-origin stream) segfaults hipStreamEndCapture on ROCm 7.2
            # (tools/capture_topo.py).  So a launch may sit on lane 0, or on the one side lane its dependencies live on.
            side = {lane[j] for j in deps[i]} - {0}
            if getattr(self, "lane_any", False):      # never captured into a HIP graph (training plans): any lane may wait on any other
                allowed = range(n_lanes)
            else:
                allowed = range(n_lanes) if not side else ([0] + list(side) if len(side) == 1 else [0])
            if l.side and n_lanes > 1 and any(k != 0 for k in allowed):
                allowed = [k for k in allowed if k != 0]   # (in program order the main chain's next launches come later: lane 0 looks free now)
            best = min(allowed, key=lambda k: (max(ready, free[k]), k))
            if crit >= 0 and lane[crit] in allowed and free[lane[crit]] <= max(ready, free[best]) + 1e-9:
                best = lane[crit]
            if i == 0 or dur - 6e-6 >= wide_s or not (win[0] <= i < win[1]):
                best = 0  # the first launch reads the caller's input; machine-filling launches stay serialized on lane 0
            lane[i] = best
            finish[i] = max(ready, free[best]) + dur
            free[best] = finish[i]
        # events: launch j records one if some launch on another lane depends on it; a lane waits for (lane', j) once
        records, waits, n_events = [-1] * n, [[] for _ in range(n)], 0
        waited = {}
        for i in range(n):
            for j in deps[i]:
                a, b = lane[i], lane[j]
                if a == b or waited.get((a, b), -1) >= j:
                    continue
                waited[(a, b)] = j
                if records[j] < 0:
                    records[j] = n_events
                    n_events += 1
                waits[i].append(records[j])
        sch = _Schedule()
        sch.n_launches, sch.n_lanes, sch.wide_s = n, n_lanes, wide_s
        sch.order, sch.lane, sch.deps = list(range(n)), lane, deps
        sch.records, sch.waits, sch.n_events, sch.events = records, waits, n_events, None  # HIP events: created by run()
        sch.used_side_lanes = sorted(set(lane) - {0})
        sch.est_makespan = max(finish) if n else 0.0
        self.__dict__["_sched"] = sch
        return sch

    def run_timed(self):
        """Replay with a HIP event pair around every launch (all on torch's current stream, which is the stream
        the kernels are launched on).  Returns per-launch milliseconds minus HALF the duration of an empty event pair
        (calibrated in the same pass, ~2.5 us): uncorrected the figures sit ~5 % above rocprofv3's kernel durations of the same
        launches, with the full pair subtracted ~7 % below; half the pair lands within ~1 %.  Slower than run(); for roofline
        accounting."""
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        evs, empty = [], []
        for i, l in enumerate(self.launches):
            if i % 8 == 0:  # calibration pairs, interleaved with the real ones
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                e1.record()
                empty.append((e0, e1))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = l.fn(*l.args, s)
            b.record()
            if rc != 0:
                L.check(rc, l.name)
            evs.append((a, b))
        torch.cuda.synchronize(self.device)
        over = 0.5 * sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2] if empty else 0.0
        return [max(a.elapsed_time(b) - over, 0.0) for a, b in evs]

    # ---- helpers ----
    def const(self, t: torch.Tensor, dtype=None) -> torch.Tensor:
        t = t.detach().to(device=self.device, dtype=dtype or t.dtype).contiguous()
        self.consts.append(t)
        return t

    def new(self, N, H, W, Cc, code) -> Act:
        return Act.of(self.pool.get((N, H, W, Cc), TORCH_DTYPE[code]))

    def release(self, a: Act):
        self.pool.put(a.buf)

    # ---- op builders ----
    def conv(self, x: Act, w: torch.Tensor, y: Act, *, R=1, S=1, stride=1, pad=0, scale=None, shift=None,
             act=L.ACT_NONE, res: Optional[Act] = None, out_mode=L.OUT_NHWC, name="conv", tile_hint=0):
        """w: packed [K, R*S*C] in x's dtype.  y: output view (dtype may be f32)."""
        K = w.shape[0]
        assert w.shape[1] == R * S * x.C, (w.shape, R, S, x.C)
        Ho = (x.H + 2 * pad - R) // stride + 1
        Wo = (x.W + 2 * pad - S) // stride + 1
        a = L.ConvArgs()
        a.x, a.w, a.y = x.ptr, w.data_ptr(), y.ptr
        a.scale = scale.data_ptr() if scale is not None else None
        a.shift = shift.data_ptr() if shift is not None else None
        a.res = res.ptr if res is not None else None
        a.x_batch_stride, a.x_pixel_stride = x.batch_stride, x.ld
        a.y_batch_stride = y.batch_stride
        a.y_pixel_stride = y.ld
        a.res_batch_stride = res.batch_stride if res is not None else 0
        a.res_pixel_stride = res.ld if res is not None else 0
        a.N, a.H, a.W, a.C, a.K, a.R, a.S = x.N, x.H, x.W, x.C, K, R, S
        a.stride, a.pad, a.Ho, a.Wo = stride, pad, Ho, Wo
        a.dtype, a.out_dtype, a.act, a.out_mode, a.tile_hint = x.code, y.code, act, out_mode, tile_hint
        a.policy, a.debug = self.conv_policy, self.conv_debug
        flops = 2.0 * x.N * Ho * Wo * K * R * S * x.C
        byts = (x.N * x.H * x.W * x.C + K * R * S * x.C) * ESIZE[x.code] + x.N * Ho * Wo * K * ESIZE[y.code]
        self.launches.append(Launch(self.lib.mtbt_conv2d_nhwc, (C.byref(a),), name, (a, x.buf, w, y.buf, scale, shift, res), flops, byts))
        self._io([x, res, w, scale, shift], [y])     # (weights / affine vectors too: a training plan rewrites them at its head)
        return a

    def stem(self, x_nchw: torch.Tensor, w, b, lnw, lnb, eps, y: Act, name="stem"):
        N, _, H, W = x_nchw.shape
        args = (x_nchw.data_ptr(), w.data_ptr(), b.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), C.c_float(eps), y.ptr,
                N, H, W, y.C, y.code)
        self.launches.append(Launch(self.lib.mtbt_stem_conv4x4_ln, args, name, (x_nchw, w, b, lnw, lnb, y.buf),
                                    2.0 * N * (H // 4) * (W // 4) * y.C * 48,
                                    N * 3 * H * W * 4 + N * (H // 4) * (W // 4) * y.C * ESIZE[y.code]))
        self._io([], [y])

    def dwconv(self, x: Act, w, y: Act, ksize, *, bias=None, lnw=None, lnb=None, eps=0.0, scale=None, shift=None,
               act=L.ACT_NONE, name="dwconv"):
        assert x.dense and y.dense and x.C == y.C
        p = lambda t: t.data_ptr() if t is not None else None
        args = (x.ptr, w.data_ptr(), p(bias), p(lnw), p(lnb), C.c_float(eps), p(scale), p(shift), act, y.ptr,
                x.N, x.H, x.W, x.C, ksize, x.code)
        n = x.N * x.H * x.W * x.C
        self.launches.append(Launch(self.lib.mtbt_dwconv_nhwc, args, name, (x.buf, w, bias, lnw, lnb, scale, shift, y.buf),
                                    2.0 * n * ksize * ksize, 2.0 * n * ESIZE[x.code]))
        self._io([x, w], [y])

    def layernorm(self, x: Act, w, b, eps, y: Act, name="layernorm"):
        assert x.dense and y.dense
        pixels = x.N * x.H * x.W
        args = (x.ptr, w.data_ptr(), b.data_ptr(), C.c_float(eps), y.ptr, pixels, x.C, x.code)
        self.launches.append(Launch(self.lib.mtbt_layernorm_nhwc, args, name, (x.buf, w, b, y.buf), 0.0,
                                    2.0 * pixels * x.C * ESIZE[x.code]))
        self._io([x], [y])

    def fuse(self, inputs, weights, modes, y: Act, bug=False, name="bifpn_fuse"):
        a = L.FuseArgs()
        for i, (t, wv, m) in enumerate(zip(inputs, weights, modes)):
            assert t.dense
            a.x[i], a.wgt[i], a.resample[i] = t.ptr, float(wv), m
        a.n_in, a.y = len(inputs), y.ptr
        a.N, a.H, a.W, a.C, a.dtype, a.add_weight_bug = y.N, y.H, y.W, y.C, y.code, int(bug)
        n = y.N * y.H * y.W * y.C
        self.launches.append(Launch(self.lib.mtbt_bifpn_fuse, (C.byref(a),), name, (a, y.buf) + tuple(t.buf for t in inputs),
                                    0.0, n * ESIZE[y.code] * (1 + len(inputs))))
        self._io(list(inputs), [y])
        return a

    def gap_fc(self, x: Act, w, b, y: torch.Tensor, name="gap_fc"):
        assert x.dense
        args = (x.ptr, w.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), x.N, x.H * x.W, x.C,
                w.shape[0], x.code)
        self.launches.append(Launch(self.lib.mtbt_gap_fc, args, name, (x.buf, w, b, y), 0.0,
                                    x.N * x.H * x.W * x.C * ESIZE[x.code]))
        self._io([x], [y])

    def bn_train(self, x: Act, y: Act, bn, act, name="bn_train"):
        """BatchNorm with batch statistics + activation over a dense NHWC tensor; updates bn.running_* in place."""
        assert x.dense and y.dense and x.C == y.C and x.code == y.code
        if bn.momentum is None:
            raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not supported")
        pixels = x.N * x.H * x.W
        nbytes = self.lib.mtbt_bn_train_workspace_bytes(pixels, x.C)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=self.device)
        rm = bn.running_mean.data_ptr() if bn.running_mean is not None else None
        rv = bn.running_var.data_ptr() if bn.running_var is not None else None
        g, b = self.const(bn.weight, torch.float32), self.const(bn.bias, torch.float32)
        args = (x.ptr, y.ptr, g.data_ptr(), b.data_ptr(), rm, rv, C.c_float(bn.momentum), C.c_float(bn.eps), act, pixels, x.C,
                x.code, ws.data_ptr(), nbytes)
        self.launches.append(Launch(self.lib.mtbt_bn_train_nhwc, args, name, (x.buf, y.buf, g, b, ws, bn), 0.0,
                                    3.0 * pixels * x.C * ESIZE[x.code]))
        self._io([x], [y])

    def cast(self, x: Act, y: Act, name="cast"):
        assert x.dense and y.dense
        n = x.N * x.H * x.W * x.C
        self.launches.append(Launch(self.lib.mtbt_cast, (x.ptr, y.ptr, n, x.code, y.code), name, (x.buf, y.buf), 0.0,
                                    n * (ESIZE[x.code] + ESIZE[y.code])))
        self._io([x], [y])

    def mlp_fused(self, t: Act, res: Act, w1, b1, w2p, b2, y: Act, name="mlp"):
        """ConvNeXt fc1 + GELU + fc2 (+ residual) in one launch (mlp_fused.hip); bf16 / fp16, d in {96, 192}."""
        assert t.dense and res.dense and y.dense and t.code in (L.BF16, L.F16) and t.C in (96, 192, 384)
        M, D = t.N * t.H * t.W, t.C
        args = (t.ptr, res.ptr, w1.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), y.ptr, M, D, t.code)
        self.launches.append(Launch(self.lib.mtbt_convnext_mlp_fused_dt, args, name, (t.buf, res.buf, w1, b1, w2p, b2, y.buf),
                                    2.0 * M * D * 4 * D * 2, 3.0 * M * D * 2 + 2.0 * 4 * D * D * 2))
        self._io([t, res], [y])

    def node(self, inputs, weights, modes, w: torch.Tensor, shift: torch.Tensor, y: Act, act=L.ACT_ELU, name="bifpn_node"):
        """BiFPN node in one launch (node_gemm.hip): weighted sum of the resampled inputs as the B-operand staging of the 1x1 GEMM + shift + act."""
        K, Cin = w.shape
        assert y.C == K and y.bs == y.H * y.W * y.ld and all(t.dense and t.C == Cin for t in inputs)
        a = L.NodeArgs()
        for i, (t, wv, m) in enumerate(zip(inputs, weights, modes)):
            a.fuse.x[i], a.fuse.wgt[i], a.fuse.resample[i] = t.ptr, float(wv), m
        a.fuse.n_in = len(inputs)
        a.fuse.N, a.fuse.H, a.fuse.W, a.fuse.C, a.fuse.dtype, a.fuse.add_weight_bug = y.N, y.H, y.W, Cin, y.code, 0
        a.w, a.shift, a.y, a.y_pixel_stride, a.K, a.act = w.data_ptr(), shift.data_ptr(), y.ptr, y.ld, K, act
        n = y.N * y.H * y.W
        self.launches.append(Launch(self.lib.mtbt_bifpn_node_nhwc, (C.byref(a),), name, (a, w, shift, y.buf) + tuple(t.buf for t in inputs),
                                    2.0 * n * K * Cin, n * (K + Cin * len(inputs)) * ESIZE[y.code]))
        self._io(list(inputs) + [w, shift], [y])
        return a

    def upconv(self, x: Act, w: torch.Tensor, shift9: torch.Tensor, y: Act, act=L.ACT_SILU, name="upconv"):
        """ConvTranspose2d(2, 2) -> Conv 3x3 + shift + activation as one direct conv on the low-resolution map (upconv_fused.hip).
        w [4*K, 4*C] composed weights, shift9 [9, K] fp32 (model.compose_upconv)."""
        K = w.shape[0] // 4
        assert w.shape[1] == 4 * x.C and tuple(shift9.shape) == (9, K) and y.C == K and y.H == 2 * x.H and y.W == 2 * x.W and x.code == y.code
        a = L.UpconvArgs()
        a.x, a.w, a.y, a.shift = x.ptr, w.data_ptr(), y.ptr, shift9.data_ptr()
        a.x_batch_stride, a.y_batch_stride, a.x_pixel_stride, a.y_pixel_stride = x.batch_stride, y.batch_stride, x.ld, y.ld
        a.N, a.H, a.W, a.C, a.K, a.dtype, a.act = x.N, x.H, x.W, x.C, K, x.code, act
        flops = 2.0 * x.N * x.H * x.W * 4 * K * 4 * x.C
        byts = (x.N * x.H * x.W * x.C + 16 * K * x.C + 4 * x.N * x.H * x.W * K) * ESIZE[x.code]
        self.launches.append(Launch(self.lib.mtbt_convt2x2_conv3x3_nhwc, (C.byref(a),), name, (a, x.buf, w, shift9, y.buf), flops, byts))
        self._io([x, w, shift9], [y])
        return a

    def raw(self, fn, args, name, keep=(), reads=(), writes=()):
        self.launches.append(Launch(fn, args, name, keep))
        self._io(list(reads), list(writes))

    def _io(self, reads, writes):
        """Record what the launch just appended reads and writes (activation views or tensors; None entries are skipped)."""
        l = self.launches[-1]
        l.reads = tuple(_region(a) for a in reads if a is not None)
        l.writes = tuple(_region(a) for a in writes if a is not None)
