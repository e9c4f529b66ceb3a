timeout 600 python -m pytest tests/test_gpu_decode.py -x -q 2>&1 | tail -15
python - <<'PY'
import torch, time
from any4_amd import quantize as Q
for n,k in ((4096,4096),(28672,4096),(4096,14336)):
    w = (torch.randn(n,k,device="cuda")*0.02).to(torch.bfloat16)
    torch.cuda.synchronize(); t=time.time(); c,l,s = Q.anyq_quantize_tensor(w); torch.cuda.synchronize()
    print(n,k,"anyq_quantize_tensor on GPU: %.2f s" % (time.time()-t))
PY
