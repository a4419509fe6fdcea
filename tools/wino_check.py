"""Winograd kernel check + micro-benchmark against the direct kernel (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dc_vic_amd import ops


F44 = os.environ.get("WINO_CHECK_F44") == "1"       # the "wino" column runs F(4x4, 3x3) (csrc/wino44.hip) instead of F(2x2, 3x3)


def run(Cin, Cout, H, W, N, reps=5, res=False, act=0, check=True, srcs_split=None):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin * 7 + Cout + H)
    x = torch.randn((N, Cin, H, W), generator=g).to(dev)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * (Cin * 9) ** -0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    r = torch.randn((N, Cout, H, W), generator=g).to(dev) if res else None
    direct = ops.ConvPlan(w, b, "conv", pad=(1, 1))
    wino = ops.ConvPlan(w, b, "conv", pad=(1, 1)); wino.wino = "force"
    if F44:
        wino.wino44 = "force"
    srcs = x if srcs_split is None else list(torch.split(x, srcs_split, dim=1))
    if srcs_split is not None:
        srcs = [s.contiguous() for s in srcs]
    yd = direct(srcs, act=act, res=r)
    yw = wino(srcs, act=act, res=r)
    torch.cuda.synchronize()
    msg = f"{Cin}->{Cout} {H}x{W} N={N} res={int(res)} act={act}:"
    if check:
        ref = F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
        if act == 3:
            ref = ref * torch.sigmoid(ref)
        if res:
            ref = ref + r.double().cpu()
        sc = ref.abs().max().item()
        ed = (yd.double().cpu() - ref).abs().max().item() / sc
        ew = (yw.double().cpu() - ref).abs().max().item() / sc
        msg += f" err direct {ed:.2e} wino {ew:.2e}"
        assert ew < 2e-5, msg
    for name, plan in (("direct", direct), ("wino", wino)):
        out = torch.empty_like(yd)
        plan(srcs, out=out, act=act, res=r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            plan(srcs, out=out, act=act, res=r)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * N * H * W * Cout * Cin * 9
        msg += f" | {name} {ms:.3f} ms {fl / ms * 1e-9:.1f} TF"
    print(msg, flush=True)


def run_ups(Cin, Cout, H, W, N, reps=5, check=True):
    """nearest x2 + conv3x3: structured Winograd against the four 2x2 phase convolutions (and fp64 torch)."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin * 3 + Cout + H)
    x = torch.randn((N, Cin, H, W), generator=g).to(dev)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * (Cin * 9) ** -0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    direct = ops.ConvPlan(w, b, "conv", pad=(1, 1), upsample=True)
    wino = ops.ConvPlan(w, b, "conv", pad=(1, 1), upsample=True); wino.wino = "force"
    yd = direct(x); yw = wino(x)
    torch.cuda.synchronize()
    msg = f"ups {Cin}->{Cout} {H}x{W}->x2 N={N}:"
    if check:
        ref = F.conv2d(F.interpolate(x.double().cpu(), scale_factor=2, mode="nearest"), w.double().cpu(), b.double().cpu(), padding=1)
        sc = ref.abs().max().item()
        ed = (yd.double().cpu() - ref).abs().max().item() / sc
        ew = (yw.double().cpu() - ref).abs().max().item() / sc
        msg += f" err phases {ed:.2e} wino {ew:.2e}"
        assert ew < 2e-5, msg
    for name, plan in (("phases", direct), ("wino", wino)):
        out = torch.empty_like(yd)
        plan(x, out=out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            plan(x, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        msg += f" | {name} {ms:.3f} ms"
    print(msg, flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode == "one":       # one Cin Cout H W N [res] [reps]: a single shape, Winograd only (for rocprofv3 --pmc)
        a = [int(v) for v in sys.argv[2:]]
        run(a[0], a[1], a[2], a[3], a[4], reps=a[6] if len(a) > 6 else 5, res=bool(a[5]) if len(a) > 5 else False, check=False)
    if mode in ("all", "ups"):
        run_ups(8, 64, 4, 16, 1)
        run_ups(16, 64, 6, 12, 2)
        run_ups(64, 128, 17, 36, 2)
        run_ups(256, 256, 128, 128, 32, check=False)
        run_ups(256, 256, 64, 64, 32, check=False)
        run_ups(512, 512, 32, 32, 32, check=False)
    if mode in ("all", "check"):
        run(8, 64, 8, 32, 1)
        run(16, 64, 10, 28, 2)
        run(64, 128, 33, 72, 2, res=True, act=3)
        run(128, 192, 64, 64, 2, res=True)
        run(256, 128, 40, 48, 1, srcs_split=[192, 64])
    if mode in ("all", "bench"):
        run(128, 128, 256, 256, 32, check=False)
        run(256, 128, 256, 256, 32, check=False)
        run(256, 256, 128, 128, 32, check=False)
        run(256, 256, 64, 64, 32, check=False)
        run(512, 512, 32, 32, 32, check=False)
        run(512, 256, 64, 64, 32, check=False)
        run(128, 128, 256, 256, 32, check=False, res=True)
