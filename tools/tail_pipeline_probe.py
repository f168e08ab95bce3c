"""Would a software-pipelined tail pay?  The step = forward F (5.9 ms, ends with the heads) + tail T (decode, NMS on 16 workgroups, mask assembly: ~0.4 ms
during which nothing else runs).  Here F and T are two captured graphs: serial (T_i behind F_i on one stream) against pipelined (T_i on a second
stream behind an event of F_i, so that it runs under the stem / stage 0 of F_{i+1}).  T works on copies of one forward's head outputs."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_bonetumor_yolo_amd import ConvNeXtBiFPNYOLO, init_synthetic_, postprocess as pp
from multitask_bonetumor_yolo_amd.model import calibrate_synthetic_heads_, synthetic_images
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = init_synthetic_(ConvNeXtBiFPNYOLO(2, 2, pretrained_backbone=False)).to(dev).eval()
m.set_compute_dtype(torch.bfloat16)
B, IMG = 16, 640
x = synthetic_images(B, IMG, 0).to(dev)
calibrate_synthetic_heads_(m, x[:4].contiguous())
with torch.no_grad():
    fwd, out = m.infer_and_detect(x, IMG, own_outputs=True)
torch.cuda.synchronize()
det = [t.clone() for t in fwd["detect_features"]]
feats, mc, protos = fwd["segment_protos"]
mc, protos = mc.clone(), protos.clone()
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def F():
    c = m.compile(x)
    m._bind_input(c, x)
    c.plan.run()
    return m._infer_dict(c, own=False)


def T():
    d = pp.decode_boxes(det, IMG, want_scores=False)
    k = pp.nms_batched(d["boxes"], d["best_score"], d["best_label"], float(IMG), 0.05, 0.6, 100)
    return pp.assemble_masks(protos, mc, k["keep_anchor"], k["counts"], (IMG, IMG))[0], k


def capture(fn, s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s), torch.no_grad():
        fn(); fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            keep = fn()
    return g, keep


m.segment.eval(); m.detect.eval()
gF, kF = capture(F, s1)
gT, kT = capture(T, s2)
gT.replay()
torch.cuda.synchronize()
assert torch.equal(kT[1]["keep_idx"], out["keep_idx"]), "tail graph != step"


def run(n, mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if mode == "F":
            with torch.cuda.stream(s1): gF.replay()
        elif mode == "T":
            with torch.cuda.stream(s1): gT.replay()
        elif mode == "serial":
            with torch.cuda.stream(s1): gF.replay(); gT.replay()
        else:
            with torch.cuda.stream(s1):
                gF.replay()
                ev = torch.cuda.Event(); ev.record(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(ev); gT.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    for md in ("F", "T", "serial", "pipelined"):
        run(5, md)
    r = {md: run(40, md) for md in ("F", "T", "serial", "pipelined")}
    print("  ".join(f"{k} {v:.3f} ms" for k, v in r.items()) + f"   pipelined saves {(r['serial'] - r['pipelined']) * 1e3:+.0f} us per step", flush=True)
