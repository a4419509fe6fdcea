"""Soak: the bench batch through compress_batch + decompress_batch many times; prints throughput and allocator state per 50 steps
(a leak or a slow drift shows here, not in a 20-step bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import BaseConfig, build_comp_model
from dc_vic_amd.synth import load_synth_weights

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
m = build_comp_model(BaseConfig.fromfile(os.path.join(root, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"}))
load_synth_weights(m, 1234); m.codec_setup()
x = (torch.rand((32, 3, 256, 256), generator=torch.Generator().manual_seed(0)) * 2 - 1).to("cuda:0")
ref = None
t0 = time.perf_counter()
for i in range(1, steps + 1):
    r = m.compress_batch(x, 0)
    out = m.decompress_batch(r["string_lists"])[0]
    if ref is None:
        torch.cuda.synchronize(); ref = (r["string_lists"], out.clone())
    if i % 50 == 0:
        torch.cuda.synchronize()
        same = r["string_lists"] == ref[0] and torch.equal(out, ref[1])
        dt = time.perf_counter() - t0; t0 = time.perf_counter()
        print(f"step {i}: {50 * 32 / dt:.1f} images/s, allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, reserved "
              f"{torch.cuda.memory_reserved() / 2**30:.2f} GiB, identical to step 1: {same}", flush=True)
        assert same
