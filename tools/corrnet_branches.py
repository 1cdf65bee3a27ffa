"""Which branch of the CorrNet forward is the critical path? Times the step (async, two streams) with one branch replaced by its
cached output. usage: python tools/corrnet_branches.py [pairs]   (run through gpurun)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morig_amd import models, synth          # noqa: E402
from morig_amd.models.corrnet import CorrNet  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    data = synth.make_batch(range(n), n_side=64, with_skin=False, n_pts=8192).to("cuda")
    model = models.corrnet(input_feature=3, output_feature=64, temprature=0.07).eval()
    synth.load_recipe(model, 0, mild=True).to("cuda")
    run = lambda: model(data, True, False)
    with torch.no_grad():
        print(f"both branches        {timed(run):8.3f} ms")
        pb, vb = CorrNet._point_branch, CorrNet._vertex_branch
        cache = {}
        def cached(key, fn):
            def inner(self, *a):
                if key not in cache:
                    cache[key] = fn(self, *a)
                return cache[key]
            return inner
        CorrNet._point_branch = cached("p", pb)
        run(); torch.cuda.synchronize()
        print(f"vertex branch + tail {timed(run):8.3f} ms")
        CorrNet._point_branch = pb
        CorrNet._vertex_branch = cached("v", vb)
        run(); torch.cuda.synchronize()
        print(f"point branch + tail  {timed(run):8.3f} ms")
        CorrNet._vertex_branch = vb
        os.environ["MORIG_TWO_STREAMS"] = "0"
        print(f"one stream           {timed(run):8.3f} ms")


if __name__ == "__main__":
    main()
