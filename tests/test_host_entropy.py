"""Host entropy coder of the product (C++, dc_vic_amd/csrc/host_entropy.cpp) against the oracle's plain-C
restatement (oracle/rans_oracle.c): byte-identical streams, exact round trips, CDF invariants.
CPU only.  Both follow SURVEY App-B (CompressAI 1.2.4, parity unpinned against the real library)."""
import numpy as np
import pytest
import torch

from dc_vic_amd import ops
from dc_vic_amd.entropy import EntropyBottleneck, GaussianMeanScaleConditional, get_scale_table
from oracle import entropy_oracle as eo


@pytest.fixture(scope="module")
def gc_tables():
    g = GaussianMeanScaleConditional(scale_bound=0.11)
    g.update_scale_table(get_scale_table(), force=True)
    return g


def test_pmf_to_cdf_matches_oracle_and_invariants():
    rng = np.random.default_rng(0)
    for n in (2, 3, 17, 300):
        pmf = rng.random(n).astype(np.float32) ** 6
        pmf /= pmf.sum()
        a, b = ops.pmf_to_quantized_cdf(pmf), eo.pmf_to_quantized_cdf(pmf)
        assert np.array_equal(a, b)
        assert a[0] == 0 and a[-1] == 65536 and np.all(np.diff(a) > 0)      # every symbol keeps a non-zero frequency
    # a peaked pmf forces the frequency-stealing loop
    pmf = np.array([1e-9] * 40 + [1.0] + [1e-9] * 40, dtype=np.float32)
    a = ops.pmf_to_quantized_cdf(pmf)
    assert np.array_equal(a, eo.pmf_to_quantized_cdf(pmf)) and np.all(np.diff(a) > 0)


def test_gaussian_tables_match_oracle(gc_tables):
    o = eo.GaussianConditionalOracle()
    assert np.array_equal(gc_tables._quantized_cdf.numpy(), o.cdf)
    assert np.array_equal(gc_tables._cdf_length.numpy(), o.cdf_length)
    assert np.array_equal(gc_tables._offset.numpy(), o.offset)
    assert gc_tables._quantized_cdf.shape[0] == 64
    # Sum of frequencies == 2^16 and strictly increasing inside each table
    for i in range(64):
        n = int(o.cdf_length[i])
        row = o.cdf[i, :n]
        assert row[0] == 0 and row[-1] == 65536 and np.all(np.diff(row) > 0)


def test_entropy_bottleneck_tables_match_oracle(synth_sd):
    eb = EntropyBottleneck(192)
    eb.load_state_dict({k.split(".", 1)[1]: v for k, v in synth_sd.items() if k.startswith("entropy_model_z.")}, strict=False)
    eb.update(force=True)
    o = eo.EntropyBottleneckOracle(synth_sd, "entropy_model_z")
    assert np.array_equal(eb._quantized_cdf.numpy(), o.cdf)
    assert np.array_equal(eb._cdf_length.numpy(), o.cdf_length)
    assert np.array_equal(eb._offset.numpy(), o.offset)
    assert len(set(o.cdf_length.tolist())) > 3          # ragged tables are exercised


@pytest.mark.parametrize("n_streams,threads", [(1, 1), (5, 3)])
def test_rans_bytes_identical_and_roundtrip(gc_tables, n_streams, threads):
    o = eo.GaussianConditionalOracle()
    rng = np.random.default_rng(1)
    n = 6 * 32 * 64
    idx = rng.integers(0, 64, size=(n_streams, n)).astype(np.int32)
    sig = o.scale_table.numpy()[idx]
    sym = np.rint(rng.standard_normal((n_streams, n)) * sig * 1.3).astype(np.int32)
    sym[:, 5] = 40000; sym[:, 6] = -40000; sym[:, 7] = 2 ** 24; sym[0, 8] = -(2 ** 24)     # bypass escapes, many nibbles
    streams = gc_tables.tables().encode(sym, idx, threads=threads)
    for i in range(n_streams):
        ref = o.encode(torch.from_numpy(sym[i]), torch.from_numpy(idx[i]))
        assert streams[i] == ref, f"stream {i} differs from the oracle coder"
    # incremental decode in six slices (CHARM order), product decoder
    dec = gc_tables.tables().decoders(streams)
    out = np.concatenate([dec.decode(idx[:, s * (n // 6):(s + 1) * (n // 6)], threads=threads) for s in range(6)], axis=1)
    dec.close()
    assert np.array_equal(out, sym)
    # cross decode: oracle decoder on the product's bytes
    od = o.stream_decoder(streams[0])
    assert np.array_equal(od.decode(torch.from_numpy(idx[0])).numpy(), sym[0])


def test_rans_empty_and_corrupt(gc_tables):
    t = gc_tables.tables()
    s = t.encode(np.zeros((2, 0), np.int32), np.zeros((2, 0), np.int32))
    assert all(len(x) == 8 for x in s)                                     # just the flushed state
    dec = t.decoders(s)
    assert dec.decode(np.zeros((2, 0), np.int32)).shape == (2, 0)
    dec.close()
    from dc_vic_amd._lib import DcvicError
    with pytest.raises(DcvicError):
        t.decoders([b"\x00\x01\x02"])                                     # not a whole number of words
    with pytest.raises(DcvicError):
        t.encode(np.zeros((1, 4), np.int32), np.full((1, 4), 64, np.int32))   # cdf index out of range
    # truncated stream: decoding more symbols than were coded must fail loudly, not read out of bounds
    idx = np.full((1, 4096), 63, np.int32)
    sym = (np.arange(4096, dtype=np.int32) % 700 - 350)[None]
    full = t.encode(sym, idx)[0]
    dec = t.decoders([full[: len(full) // 2 // 4 * 4]])
    with pytest.raises(DcvicError):
        dec.decode(idx)
    dec.close()


def test_worker_pool_many_calls_and_thread_counts(gc_tables):
    """The coder's persistent worker pool (csrc/host_entropy.cpp): hundreds of back-to-back encode / incremental-decode calls with
    changing thread counts and stream counts give the streams and symbols of the single-threaded coder, from two Python threads at once
    as well (calls are serialised inside the library)."""
    import threading
    T = gc_tables.tables()
    errors = []

    def worker(seed):
        r = np.random.RandomState(seed)
        for it in range(60):
            ns, n = int(r.randint(1, 40)), int(r.randint(1, 600))
            idx = r.randint(0, 64, (ns, n)).astype(np.int32)
            sym = r.randint(-30, 31, (ns, n)).astype(np.int32)
            sym[r.rand(ns, n) < 0.01] = 5000                       # some out of range: bypass escapes
            ref = T.encode(sym, idx, threads=1)
            thr = int(r.choice([2, 3, 8, 16, 32, 64]))
            got = T.encode(sym, idx, threads=thr)
            if got != ref:
                errors.append(("encode", seed, it, thr)); return
            dec = T.decoders(got)
            try:
                half = n // 2
                a = dec.decode(np.ascontiguousarray(idx[:, :half]), threads=thr) if half else np.zeros((ns, 0), np.int32)
                b = dec.decode(np.ascontiguousarray(idx[:, half:]), threads=int(r.choice([1, 5, 32])))
            finally:
                dec.close()
            if not (np.array_equal(a, sym[:, :half]) and np.array_equal(b, sym[:, half:])):
                errors.append(("decode", seed, it, thr)); return

    ts = [threading.Thread(target=worker, args=(s_,)) for s_ in (1, 2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errors, errors
