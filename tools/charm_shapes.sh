python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from dc_vic_amd import ops
from dc_vic_amd._lib import lib
L = lib()
dev = torch.device("cuda:0")
for (C, H) in ((256, 128), (256, 64), (512, 32)):
    x = torch.randn(32, C, H, H, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.02; b = torch.zeros(C, device=dev)
    plan = ops.ConvPlan(w, b, "conv", pad=(1, 1), upsample=True)
    for dma in (0, 1):
        L.dcvic_conv_set_tuning(dma, 1, -1)
        out = plan(x); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): plan(x, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = 2.0 * 32 * (2 * H) ** 2 * C * C * 4
        print(f"ups conv {C} @{H}->{2*H} dma={dma}: {ms:.3f} ms {fl / ms * 1e-9:.1f} TF (variant {L.dcvic_conv_last_variant()})", flush=True)
PY
