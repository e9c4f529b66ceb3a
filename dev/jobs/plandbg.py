import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from any4_amd import quantize as Q
lin = torch.nn.Linear(4096, 4096, dtype=torch.bfloat16, device="cuda", bias=False)
x = torch.randn(1, 4096, dtype=torch.bfloat16, device="cuda")
for name, q in (("anyq", Q.anyq_layer(lin, pseudo=False)), ("intq", Q.intq_layer(lin, pseudo=False))):
    for _ in range(3): q(x)
    print(name, type(q).__name__, "plan:", q.__dict__.get("_plan", "none")[1] if q.__dict__.get("_plan") else None, "reshaped", q.weight_reshaped, "kernel", q.kernel)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2000): q(x)
    torch.cuda.synchronize()
    print("  wall per call us", (time.perf_counter() - t) / 2000 * 1e6)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): q(x)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
