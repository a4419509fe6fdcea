"""Single-layer conv micro-benchmark through the C ABI (diagnostic; used under rocprofv3 --pmc)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import ops

def main():
    Cin, Cout, H, W, N, k = [int(v) for v in (sys.argv[1:7] if len(sys.argv) >= 7 else (256, 256, 128, 128, 32, 3))]
    reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn((N, Cin, H, W), generator=g).to(dev)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (Cin * k * k) ** -0.5).to(dev)
    b = torch.zeros(Cout, device=dev)
    plan = ops.ConvPlan(w, b, "conv", pad=(k // 2, k // 2))
    out = plan(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan(x, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * N * H * W * Cout * Cin * k * k
    print(f"conv {Cin}->{Cout} k{k} {H}x{W} N={N}: {ms:.3f} ms  {fl / ms * 1e-9:.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
