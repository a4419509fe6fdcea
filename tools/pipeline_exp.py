"""Experiment: does coding two half-batches on two streams (two host threads) beat one batch of 32?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DCVIC_CHARM_STREAMS"] = "0"
import torch
from dc_vic_amd import BaseConfig, build_comp_model
from dc_vic_amd.synth import load_synth_weights
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
opt = BaseConfig.fromfile(os.path.join(root, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"})
m = build_comp_model(opt); load_synth_weights(m, 1234); m.codec_setup()
x = (torch.rand((32, 3, 256, 256), generator=torch.Generator().manual_seed(1000)) * 2 - 1).to("cuda:0")

def one(xb):
    r = m.compress_batch(xb, 0)
    m.decompress_batch(r["string_lists"])

def seq():
    one(x)

def par(k):
    parts = x.chunk(k)
    streams = [torch.cuda.Stream() for _ in parts]
    def w(i):
        with torch.cuda.stream(streams[i]):
            one(parts[i])
    ths = [threading.Thread(target=w, args=(i,)) for i in range(k)]
    for s in streams: s.wait_stream(torch.cuda.current_stream())
    for t in ths: t.start()
    for t in ths: t.join()
    for s in streams: torch.cuda.current_stream().wait_stream(s)

for name, fn in (("one batch of 32", seq), ("2 x 16 on two streams", lambda: par(2)), ("4 x 8 on four streams", lambda: par(4)), ("one batch of 32", seq)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / 3:.1f} ms per 32 images", flush=True)
