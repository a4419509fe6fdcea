"""HBM-bound kernels at a saturating synthetic size: VQ nearest-codeword search on 2^22 latent
vectors and the Gaussian rate kernel on 2^24 elements; prints achieved algorithmic GB/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dc_vic_amd import ops
from dc_vic_amd.entropy import get_scale_table

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    out = {}
    # VQ: M = 2^22 vectors as [64, 4, 256, 256]; codebook U(-1/256, 1/256)
    z = torch.randn((64, 4, 256, 256), generator=g).to(dev)
    cb = ((torch.rand((256, 4), generator=g) * 2 - 1) / 256).to(dev)
    M = 64 * 256 * 256
    t = timeit(lambda: ops.vq_argmin(z, cb, want_zq=False, want_feat=False))
    out["vq_idx_only"] = {"vectors": M, "s": t, "GBps": M * 24 / t / 1e9, "bytes_per_vector": 24}
    t = timeit(lambda: ops.vq_argmin(z, cb, want_zq=True, want_feat=False))
    out["vq_idx_zq"] = {"vectors": M, "s": t, "GBps": M * 40 / t / 1e9, "bytes_per_vector": 40}
    # product mode: index + z_q + the [z_q | one-hot] feature the ELIC encoder consumes (1040 B written per vector)
    z16 = z[:16].contiguous(); M16 = 16 * 256 * 256
    t = timeit(lambda: ops.vq_argmin(z16, cb, want_zq=True, want_feat=True))
    out["vq_idx_zq_onehot"] = {"vectors": M16, "s": t, "GBps": M16 * 1080 / t / 1e9, "bytes_per_vector": 1080}
    # rate: 2^24 elements, 16 B/element in (y, mu, sigma) + 4 out (y_hat) ... symbols+indexes+likelihood = 28 B
    y = torch.randn((64, 64, 64, 64), generator=g).to(dev) * 2
    mu = torch.randn((64, 64, 64, 64), generator=g).to(dev)
    sg = torch.rand((64, 64, 64, 64), generator=g).to(dev) * 2
    tab = get_scale_table().to(dev)
    yh = torch.empty_like(y); lik = torch.empty_like(y)
    sym = torch.empty(y.shape, dtype=torch.int32, device=dev); ix = torch.empty_like(sym)
    bits = torch.zeros(64, device=dev)
    E = y.numel()
    t = timeit(lambda: ops.gaussian_rate(y, None, mu, sg, tab, yh, None, None, lik, bits))
    out["gaussian_rate_lik"] = {"elements": E, "s": t, "GBps": E * 20 / t / 1e9, "bytes_per_element": 20}
    t = timeit(lambda: ops.gaussian_rate(y, None, mu, sg, tab, yh, sym, ix, None, None))
    out["gaussian_rate_sym"] = {"elements": E, "s": t, "GBps": E * 24 / t / 1e9, "bytes_per_element": 24}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
