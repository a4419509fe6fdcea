"""256x256 N=1 compress+decompress x10 (profiling target: is N=1 launch-bound or GPU-bound?)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import BaseConfig, build_comp_model
from dc_vic_amd.synth import load_synth_weights
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
opt = BaseConfig.fromfile(os.path.join(root, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"})
m = build_comp_model(opt); load_synth_weights(m, 1234); m.codec_setup()
x = (torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(3)) * 2 - 1).to("cuda:0")
for _ in range(2):
    r = m.compress(x, 0); m.decompress(r["string_list"])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r = m.compress(x, 0); m.decompress(r["string_list"])
torch.cuda.synchronize()
print("ms per compress+decompress:", 1e3 * (time.perf_counter() - t0) / 10, "graphs:", [k[0] for k in m._graphs.entries])
