"""Can one jointnet eval forward be captured in a HIP graph and replayed? (probe; run through gpurun under timeout)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morig_amd import models, native, synth  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
d = synth.make_batch(range(nb), n_side=64, with_skin=False).to(dev)
d.num_graphs = nb
m = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval()
synth.load_recipe(m, 0, mild=True).to(dev)
ops = native.get_ops()
with torch.no_grad():
    for _ in range(3):
        ref = m(d, d.pred_flow)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m(d, d.pred_flow)
    torch.cuda.synchronize()
    print("eager  ms/step", (time.perf_counter() - t0) / 10 * 1e3, flush=True)
    pass                                            # weights are packed by the warm-up forwards
    flag = ops._flag(dev)
    flag.zero_()
    st = ops._state()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    from morig_amd.models import basic_modules as bm
    with torch.cuda.stream(side):
        bm._ctx.key = m._param_key()
        st.csr_status = []
        for _ in range(2):
            m._forward(d, d.pred_flow)              # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            out = m._forward(d, d.pred_flow)
    torch.cuda.synchronize()
    print("captured", flush=True)
    g.replay(); torch.cuda.synchronize()
    print("replayed once; flag", int(flag.item()), "max diff", float((out[2] - ref[2]).abs().max()), flush=True)
    for i in range(3):
        g.replay(); torch.cuda.synchronize()
        print("replay", i + 2, "ok; max diff", float((out[2] - ref[2]).abs().max()), flush=True)
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay(); torch.cuda.synchronize()
    print("graph  ms/step (sync between replays)", (time.perf_counter() - t0) / 20 * 1e3, flush=True)
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print("graph  ms/step (back to back)", (time.perf_counter() - t0) / 20 * 1e3, flush=True)
