"""N=1 latency of compress + decompress (the reference CLI's batch_size=1 regime) at 256x256 and Kodak 512x768."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import BaseConfig, build_comp_model
from dc_vic_amd.synth import load_synth_weights

def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    opt = BaseConfig.fromfile(os.path.join(root, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"})
    m = build_comp_model(opt); load_synth_weights(m, 1234); m.codec_setup()
    out = {}
    for name, shape in (("256x256", (1, 3, 256, 256)), ("kodak_512x768", (1, 3, 512, 768)), ("2k_1280x2048_tiled", (1, 3, 1280, 2048))):
        g = torch.Generator().manual_seed(3)
        x = (torch.rand(shape, generator=g) * 2 - 1).to("cuda:0")
        for _ in range(3):          # lazy weight packs, then hipGraph capture (second sighting), then steady state
            r = m.compress(x, 0); m.decompress(r["string_list"])
        torch.cuda.synchronize()
        reps = 5 if shape[2] <= 512 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            r = m.compress(x, 0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(reps):
            m.decompress(r["string_list"])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        out[name] = {"compress_ms": 1e3 * (t1 - t0) / reps, "decompress_ms": 1e3 * (t2 - t1) / reps,
                     "bytes": sum(len(s) for s in r["string_list"])}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
