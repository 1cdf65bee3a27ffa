"""List every GEMM-type launch of one jointnet forward with its shape and (synchronised) time.  Usage: python tools/gemm_shapes.py [batch]"""
import sys, time, collections
import torch
sys.path.insert(0, ".")
from morig_amd import native, synth, models

def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    ops = native.get_ops()
    model = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval()
    synth.load_recipe(model, 0, mild=True).to(dev)
    data = synth.collate([synth.make_mesh(i, n_side=64, with_skin=False) for i in range(batch)]).to(dev)
    flow = data.pred_flow
    with torch.no_grad():
        model(data, flow); model(data, flow); torch.cuda.synchronize()
    rows = []
    def wrap(name):
        fn = getattr(ops, name)
        def timed(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            if name == "gemm":
                x, lin = a[0], a[1]
                desc = f"M={x.rows} K={lin.K} N={lin.N} xs={int(bool(k.get('x_split')))} ys={int(bool(k.get('y_split')))} pool={int(k.get('pool') is not None)} rb={int(k.get('rowbias') is not None)}"
            elif name == "edgeconv":
                desc = f"H={a[3].H if hasattr(a[3], 'H') else '?'} rows={a[2].capacity} x{k.get('replicas', 1)} quad={int(a[2].quad)}"
            else:
                desc = ""
            rows.append((name, desc, dt))
            return r
        setattr(ops, name, timed)
    for n in ("gemm", "edgeconv", "cls_attention", "csr_build", "copy2d", "copy2d_pad", "rownorm", "frame_reduce", "gather_cols"):
        if hasattr(ops, n): wrap(n)
    with torch.no_grad():
        t0 = time.perf_counter(); model(data, flow); torch.cuda.synchronize(); tot = (time.perf_counter() - t0) * 1e3
    agg = collections.OrderedDict()
    for n, d, dt in rows:
        k = (n, d); c, t = agg.get(k, (0, 0.0)); agg[k] = (c + 1, t + dt)
    for (n, d), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t:8.3f} ms  x{c:<3d} {n:14s} {d}")
    print(f"total (serialised, with syncs) {tot:.1f} ms")

main()
