"""Stage-level and end-to-end parity of the HIP path on a real MI355X.

 * stages vs the committed goldens (produced by the reference's own modules, oracle/gen_golden.py);
 * end-to-end vs the CPU oracle on the same seeded inputs, with integer decisions (VQ indices,
   symbols, cdf indexes) compared exactly up to itemised fp32 near-ties;
 * size-independent properties at the benchmark size: encode -> bytes -> decode round trip is
   bit-exact, results do not depend on the batch an image travels in.
Tolerances are stated where used (fp32 MFMA fma-chains vs MKL/oneDNN summation order)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    assert torch.cuda.is_available()
    from dc_vic_amd import BaseConfig, build_comp_model
    from dc_vic_amd.synth import load_synth_weights
    opt = BaseConfig.fromfile(os.path.join(ROOT, "config", "dc_vic_synthetic.yaml"), {"device": "cuda:0"})
    m = build_comp_model(opt)
    load_synth_weights(m, 1234)
    m.codec_setup()
    return m


@pytest.fixture(scope="module")
def oracle(synth_sd):
    from oracle.dcvic_oracle import Oracle
    return Oracle(synth_sd)


def dev(a):
    return torch.from_numpy(np.asarray(a)).to("cuda:0")


def close(a, b, rtol, atol):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def img(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * 2 - 1


# ------------------------------------------------------------------------------ stages vs goldens
def test_stage_vqgan_encoder(model, golden):
    z = model.vq_model.encode(dev(golden["a4_x"]))
    close(z, golden["a4_z"], rtol=2e-3, atol=2e-4)     # 30 layers of fp32 re-association
    z = model.vq_model.encode(dev(golden["a4b_x"]))
    close(z, golden["a4b_z"], rtol=2e-3, atol=2e-4)


def test_stage_vq_indices(model, golden):
    zq, _, (_, _, idx) = model.vq_model.quantize(dev(golden["a4_z"]))
    assert np.array_equal(idx.cpu().numpy(), golden["a5_idx"])          # same z_e in -> index-exact
    assert np.array_equal(zq.cpu().numpy(), golden["a5_zq"])
    zq, _, (_, _, idx) = model.vq_model.quantize(dev(golden["a5b_z"]))
    assert (idx.cpu().numpy() == golden["a5b_idx"]).mean() >= 0.999
    ok = idx.cpu().numpy() == golden["a5b_idx"]
    assert np.array_equal(zq.cpu().numpy().transpose(0, 2, 3, 1)[ok], golden["a5b_zq"].transpose(0, 2, 3, 1)[ok])


def test_stage_elic_encoder(model, golden):
    from dc_vic_amd import ops
    x = dev(golden["a4_x"])
    _, _, feat = ops.vq_argmin(dev(golden["a4_z"]), model.vq_model.quantize.embedding.weight, want_zq=False, want_feat=True)
    for q in (0, 3):
        y = model.encoder(x, feat, model.selected_beta_rate[q], model.selected_beta_vq[q])
        close(y, golden[f"a6_y_q{q}"], rtol=1e-3, atol=2e-4)
    # per-sample beta tensors
    g = torch.Generator().manual_seed(15)
    f2 = (torch.randn((2, 260, 8, 8), generator=g) * 0.3).to("cuda:0")
    y = model.encoder(img((2, 3, 64, 64), 14).to("cuda:0"), f2, torch.tensor([2.29, 0.62]), torch.tensor([3.0, 1.5]))
    close(y, golden["a6b_y"], rtol=1e-3, atol=2e-4)


def test_stage_hyper(model, golden):
    close(model.hyperencoder(dev(golden["a6_y_q0"])), golden["a7_z"], rtol=1e-3, atol=1e-4)
    close(model.hyperdecoder(dev(golden["a9_zhat"])), golden["a9_out"], rtol=1e-3, atol=1e-4)


def test_stage_elic_decoder_feats(model, golden):
    f1, fd = model.decoder.get_feats(dev(golden["a14_yhat"]), model.selected_beta_rate[1], model.selected_beta_vq[1])
    close(f1, golden["a14_feat1"], rtol=2e-3, atol=2e-3)
    close(fd["block_1_4"][:, :, :8, :8], golden["a14_b14_crop"], rtol=2e-3, atol=2e-3)
    close(fd["block_1_2"][:, :, 10:18, 20:28], golden["a14_b12_crop"], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("tag", ["a15", "a15b"])
def test_stage_swin_estimator(model, golden, tag):
    pe, lg = model.vq_estimator(dev(golden[f"{tag}_feat"]), want_embed=True)
    close(pe, golden[f"{tag}_pred_embed"], rtol=2e-3, atol=2e-3)
    close(lg[:, ::16, :4, :4], golden[f"{tag}_logits_crop"], rtol=5e-3, atol=5e-3)
    assert (lg.argmax(1).cpu().numpy() == golden[f"{tag}_argmax"]).mean() > 0.995


def test_stage_fusion_decoder(model, golden):
    from dc_vic_amd import ops
    pq = model.vq_model.post_quant_conv
    logits = torch.nn.functional.one_hot(dev(golden["a17_idx"]), 256).permute(0, 3, 1, 2).float().contiguous()
    idx, lat = ops.argmax_lut(logits, model.vq_model.quantize.embedding.weight, pq.weight.reshape(4, 4).contiguous(), pq.bias)
    assert np.array_equal(idx.cpu().numpy(), golden["a17_idx"])
    close(lat, golden["a17_lat"], rtol=1e-5, atol=1e-7)
    cf = {k: dev(golden[f"a17_{k}"]) for k in ("block_1_8", "block_1_4", "block_1_2")}
    out = model.fusion_module(lat, cf, model.vq_model.decoder, w=1.0)
    close(out[:, :, 16:48, 30:62], golden["a17_out_crop"], rtol=5e-3, atol=5e-3)
    close(out[:, :, ::4, ::4], golden["a17_out_ds"], rtol=5e-3, atol=5e-3)
    plain = model.vq_model.decoder(lat)
    close(plain[:, :, ::4, ::4], golden["a17p_out_ds"], rtol=5e-3, atol=5e-3)


# ------------------------------------------------------------------------------ end to end vs oracle
def test_compress_vs_oracle(model, oracle):
    """64x96 image (ragged: pads to 64x128): integer decisions vs the oracle, near-ties itemised."""
    x = img((1, 3, 64, 96), 101)
    ro = oracle.compress(x, 0)
    rg = model.compress(x, 0)
    # VQ indices
    gi, oi = rg["vq_indices"].cpu(), ro["vq_indices"]
    assert (gi == oi).float().mean() >= 0.99
    if torch.equal(gi, oi):
        # teacher-forced identical VQ input -> symbols may only differ at rounding near-ties
        ys_g, ys_o = rg["y_symbols"].cpu(), ro["y_symbols"]
        mism = (ys_g != ys_o)
        frac = ((ro["y"] - ro["mu"]) - torch.floor(ro["y"] - ro["mu"]) - 0.5).abs()
        first = mism.reshape(6, -1).any(1).float().argmax().item() if mism.any() else None
        if mism.any():   # the first slice that differs must differ only at near-ties (later ones inherit the change)
            sl = slice(first * 32, (first + 1) * 32)
            assert float(frac[:, sl][mism[:, sl]].max()) < 5e-3
        assert mism.float().mean() < 0.02
        if not mism.any() and torch.equal(rg["z_symbols"].cpu(), ro["z_symbols"]) and torch.equal(rg["y_indexes"].cpu(), ro["y_indexes"]):
            assert rg["string_list"] == ro["string_list"]            # bitstream bytes identical
    assert abs(rg["pred_y_bpp"] + rg["pred_z_bpp"] - ro["pred_y_bpp"] - ro["pred_z_bpp"]) < 0.02 * (ro["pred_y_bpp"] + ro["pred_z_bpp"])


def test_decompress_oracle_stream(model, oracle):
    """The oracle's bitstream decodes on the HIP path: symbols exact, reconstruction within tolerance."""
    x = img((1, 3, 64, 64), 102)
    ro = oracle.compress(x, 2)
    img_o, zh_o, yh_o, idx_o = oracle.decompress(ro["string_list"])
    img_g, zh_g, yh_g = model.decompress(ro["string_list"])
    assert torch.equal(zh_g.cpu(), zh_o)
    # y_hat: symbols come from the same stream; mu is recomputed on the GPU -> close, and the decoded
    # symbols stay consistent as long as no cdf index flips (itemised by the y_hat tolerance)
    close(yh_g, yh_o, rtol=1e-2, atol=2e-2)
    psnr = 10 * np.log10(4.0 / float(((img_g.cpu() - img_o) ** 2).mean()))
    assert psnr > 35.0, f"HIP vs oracle reconstruction PSNR {psnr:.1f} dB"


def test_run_model_vs_oracle(model, oracle):
    x = img((2, 3, 64, 64), 103)
    ro = oracle.run_model(x, 1.51, 2.25)
    rg = model.run_model(x, is_train=False, beta_rate=1.51, beta_vq=2.25)
    assert abs(rg["bpp"] - ro["bpp"]) < 0.02 * ro["bpp"] + 1e-4
    assert (rg["gt_vq_indices"].cpu() == ro["gt_vq_indices"]).float().mean() >= 0.99
    assert rg["fake_images"].shape == ro["fake_images"].shape
    assert float(rg["fake_images"].abs().max()) <= 1.0


# ------------------------------------------------------------------------------ properties at full size
def test_roundtrip_bit_exact_256(model):
    """encode -> bytes -> decode at the benchmark shape: the decoder reproduces the encoder's y_hat / z_hat
    bit for bit (GPU-resident CHARM on both sides), and two identical calls give identical bytes."""
    x = img((4, 3, 256, 256), 104)
    r = model.compress_batch(x, 0)
    r2 = model.compress_batch(x, 0)
    assert r["string_lists"] == r2["string_lists"]
    imgs, z_hat, y_hat = model.decompress_batch(r["string_lists"])
    assert torch.equal(y_hat, r["y_hat"]) and torch.equal(z_hat, r["z_hat"])
    assert imgs.shape == (4, 3, 256, 256) and float(imgs.abs().max()) <= 1.0
    # batch invariance: image 2 alone gives the same bytes and the same reconstruction
    r1 = model.compress(x[2:3], 0)
    assert r1["string_list"] == r["string_lists"][2]
    i1, _, _ = model.decompress(r1["string_list"])
    assert torch.equal(i1[0], imgs[2])
    # real bytes vs predicted bits: rANS overhead is small
    for i in range(4):
        real_bits = 8 * (len(r["string_lists"][i][1]) + len(r["string_lists"][i][2]))
        pred = float(r["pred_y_bit"][i] + r["pred_z_bit"][i])
        # rANS adds <= 2 flush words per stream + 16-bit probability quantisation to the estimate;
        # escape-coded outliers may undercut the 1e-9 likelihood floor, so the band is two-sided
        assert 0.85 * pred - 256 < real_bits < 1.10 * pred + 256, (real_bits, pred)


def test_ragged_and_kodak_shape(model):
    """Non-multiple-of-64 image (reflect pad + crop) and the Kodak shape 512x768 round-trip."""
    for shape, q in (((1, 3, 100, 70), 4), ((1, 3, 512, 768), 2)):
        x = img(shape, 105)
        r = model.compress(x, q)
        out, z_hat, y_hat = model.decompress(r["string_list"])
        assert out.shape == shape
        assert torch.equal(y_hat, r["y_hat"])
        from dc_vic_amd.codec_utils import HeaderHandler
        hd = HeaderHandler().decode(r["string_list"][0])
        assert hd["img_size"] == shape[2:] and hd["quality_ind"] == q


def test_charm_split_and_streams_bit_identical(model, monkeypatch):
    """The stacked hyperprior partial convs (accumulator hand-over through `init`) and the two-stream schedule are
    pure re-schedulings: symbols, cdf indexes and y_hat are bit-identical to the plain sequential CHARM."""
    x = img((3, 3, 128, 192), 106)
    monkeypatch.setenv("DCVIC_CHARM_SPLIT", "0"); monkeypatch.setenv("DCVIC_CHARM_STREAMS", "0")
    a = model.compress_batch(x, 1)
    monkeypatch.setenv("DCVIC_CHARM_SPLIT", "1"); monkeypatch.setenv("DCVIC_CHARM_STREAMS", "1")
    b = model.compress_batch(x, 1)
    assert torch.equal(a["y_symbols"], b["y_symbols"]) and torch.equal(a["y_indexes"], b["y_indexes"])
    assert torch.equal(a["y_hat"], b["y_hat"])
    assert a["string_lists"] == b["string_lists"]


def test_hipgraph_replay_identical_to_eager(model):
    """Opt-in (DCVIC_GRAPHS=1): small batches replay the encoder / decoder networks as captured hipGraphs.  Same
    kernels, same arguments: bytes and reconstructions equal the eager path's, across repeated replays and a second shape."""
    g = model._graphs
    was = g.disabled
    try:
        for shape, q in (((1, 3, 256, 256), 0), ((2, 3, 128, 192), 3)):
            x1, x2 = img(shape, 107), img(shape, 108)
            g.disabled = True
            e1, e2 = model.compress_batch(x1, q), model.compress_batch(x2, q)
            ei1 = model.decompress_batch(e1["string_lists"])[0].clone()
            ei2 = model.decompress_batch(e2["string_lists"])[0].clone()
            g.disabled = False
            g.clear()
            for _ in range(2):      # first pass captures, second replays
                r1, r2 = model.compress_batch(x1, q), model.compress_batch(x2, q)
                assert r1["string_lists"] == e1["string_lists"] and r2["string_lists"] == e2["string_lists"]
                assert torch.equal(r1["y"], e1["y"]) and torch.equal(r2["vq_indices"], e2["vq_indices"])
                i1 = model.decompress_batch(r1["string_lists"])[0].clone()
                i2 = model.decompress_batch(r2["string_lists"])[0].clone()
                assert torch.equal(i1, ei1) and torch.equal(i2, ei2)
            assert not g.disabled, "hipGraph capture failed and fell back to eager"
            kinds = sorted(k[0] for k in g.entries)
            assert "dec" in kinds and "enc" in kinds, kinds
    finally:
        g.disabled = was
