"""Per conv shape: launches, ms per step, achieved algorithmic TFLOP/s over one compress + decompress step of the bench batch (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import BaseConfig, build_comp_model, ops
from dc_vic_amd.synth import load_synth_weights

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = build_comp_model(BaseConfig.fromfile(os.path.join(root, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"}))
load_synth_weights(m, 1234); m.codec_setup()
x = (torch.rand((32, 3, 256, 256), generator=torch.Generator().manual_seed(0)) * 2 - 1).to("cuda:0")
for it in range(2):
    if it == 1:
        ops.kernel_events_start()
    r = m.compress_batch(x, 0)
    m.decompress_batch(r["string_lists"])
torch.cuda.synchronize()
ops.kernel_events_stop()
print(ops.shape_stats_report())
