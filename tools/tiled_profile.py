"""One tiled image (1280x2048: 28 windows of 512^2) through compress + decompress a few times (for rocprofv3 --stats; diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import BaseConfig, build_comp_model
from dc_vic_amd.synth import load_synth_weights

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = build_comp_model(BaseConfig.fromfile(os.path.join(root, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"}))
load_synth_weights(m, 1234); m.codec_setup()
x = (torch.rand((1, 3, 1280, 2048), generator=torch.Generator().manual_seed(0)) * 2 - 1).to("cuda:0")
for _ in range(4):
    r = m.compress(x, 0); m.decompress(r["string_list"])
torch.cuda.synchronize()
print("ok")
