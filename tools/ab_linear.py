"""Time sr_linear_fwd / bwd_input / bwd_weight on one layer shape: tools/ab_linear.py [P] [K] [N] [act]."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import ops
dev = "cuda:0"
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 512
act = (sys.argv[4] if len(sys.argv) > 4 else "sin")
act = None if act == "none" else act
x = torch.randn(P, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev); gy = torch.randn(P, N, device=dev)
src = [(x, act, 1.0, 1)]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * P * K * N
f = t(lambda: ops.linear_fwd(src, w, b, P))
dx = t(lambda: ops.linear_bwd_input(gy, None, None, w, 0, src[0], P))
dw = t(lambda: ops.linear_bwd_weight(gy, None, None, src, P, N))
print(f"P={P} K={K} N={N} act={act}: fwd {f:.0f} us ({fl/f/1e6:.0f} TF)  dX {dx:.0f} us ({fl/dx/1e6:.0f} TF)  dW {dw:.0f} us ({fl/dw/1e6:.0f} TF)   [fp32-equivalent FLOP, 3 bf16 passes each]")
