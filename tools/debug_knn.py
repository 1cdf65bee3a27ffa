import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morig_amd import native, synth
from morig_amd.native import Mat
ops = native.get_ops()
old = C.CDLL(os.path.join(ROOT, "morig_amd/lib/variants/lib_oldknn.so"))
data = synth.make_batch(range(2), n_side=64, with_skin=False, n_pts=8192).to("cuda")
pos0 = torch.zeros((data.pts.shape[0], 4), device="cuda"); pos0[:, :3] = data.pts
B = 2
ptr0 = torch.tensor([0, 8192, 16384], dtype=torch.int32, device="cuda")
ptr1 = torch.tensor([0, 4096, 8192], dtype=torch.int32, device="cuda")
idx = ops.fps(Mat.of(pos0, 0, 3), ptr0, ptr1, None, B, 8192, 8192)
pos1 = pos0[idx.long()].contiguous()
i_new, w_new = ops.knn_search(Mat.of(pos1, 0, 3), ptr1, Mat.of(pos0, 0, 3), ptr0, B, 8192, 3)
i_old = torch.empty_like(i_new); w_old = torch.empty_like(w_new)
sig = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
old.morig_knn_search.argtypes = sig; old.morig_knn_search.restype = C.c_int
torch.cuda.synchronize()
st = old.morig_knn_search(pos1.data_ptr(), 4, ptr1.data_ptr(), pos0.data_ptr(), 4, ptr0.data_ptr(), B, 16384, 8192, 3, i_old.data_ptr(), w_old.data_ptr(), None)
torch.cuda.synchronize()
print("status", st, "idx differ:", int((i_new != i_old).sum()), "wgt differ:", int((w_new != w_old).sum()))
bad = torch.nonzero((i_new != i_old).any(1)).flatten()[:5]
for t in bad.tolist():
    print(t, i_new[t].tolist(), i_old[t].tolist(), w_new[t].tolist(), w_old[t].tolist())
