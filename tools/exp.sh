python - <<'PY'
import ctypes, torch, sys, time
sys.path.insert(0, '.')
import bench
from any4_amd import _lib
lib = _lib.load()
dev = torch.device('cuda', 0)
L, m, n, k, g = 64, 1, 4096, 4096, 128
w, x, sz, lut, y = bench.make_batch(L, m, n, k, g, 4, dev, 1)
args = _lib.W4Gemm(x=x.data_ptr(), w=w.data_ptr(), qinfo=sz.data_ptr(), lut=lut.data_ptr(), y=y.data_ptr(), m=m, wrows=n, k=k, group=g,
    qtype=_lib.TG_Q_ANY4_ROWWISE, dtype=_lib.TG_BF16, w_on_right=1, inner_k_tiles=4, batch=L, stride_x=x.stride(0)*2, stride_w=w.stride(0)*4,
    stride_qinfo=sz.stride(0)*2, stride_lut=lut.stride(0)*2, stride_y=y.stride(0)*2)
st = torch.cuda.current_stream()
torch.cuda.synchronize()
N = 600
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
for s in range(N):
    ev[s][0].record(st); lib.tg_gemm_w4(ctypes.byref(args), 0, st.cuda_stream); ev[s][1].record(st)
torch.cuda.synchronize()
ts = [a.elapsed_time(b) * 1e3 for a, b in ev]
for i in range(0, N, 25):
    seg = ts[i:i+25]; print(i, "avg %.1f min %.1f max %.1f us" % (sum(seg)/len(seg), min(seg), max(seg)))
time.sleep(2)
for s in range(50):
    ev[s][0].record(st); lib.tg_gemm_w4(ctypes.byref(args), 0, st.cuda_stream); ev[s][1].record(st)
torch.cuda.synchronize()
ts = [a.elapsed_time(b) * 1e3 for a, b in ev[:50]]
print("after 2 s idle:", " ".join("%.0f" % t for t in ts))
PY
