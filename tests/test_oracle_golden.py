"""Pins oracle/dcvic_oracle.py against fixtures produced by the reference's own modules
(oracle/gen_golden.py, build container).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dcvic_oracle as O

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
TOL = dict(rtol=2e-4, atol=2e-4)   # same torch CPU ops; slack only for thread-count dependent reduction order


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, **kw):
    kw = {**TOL, **kw}
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), **kw)


def summ(x):
    x = x.double()
    return np.array([x.sum().item(), x.abs().sum().item(), (x * x).sum().item()])


def test_vqgan_encoder(golden, synth_sd):
    z = O.vq_encode_whole(synth_sd, t(golden["a4_x"]))
    close(z, golden["a4_z"])
    z = O.vq_encode_whole(synth_sd, t(golden["a4b_x"]))
    close(z, golden["a4b_z"])


def test_vq_search_index_exact(golden, synth_sd):
    zq, idx = O.vq_quantize(synth_sd, t(golden["a4_z"]))
    assert np.array_equal(idx.numpy(), golden["a5_idx"])
    assert np.array_equal(zq.numpy(), golden["a5_zq"])
    zq, idx = O.vq_quantize(synth_sd, t(golden["a5b_z"]))
    assert np.array_equal(idx.numpy(), golden["a5b_idx"])
    assert np.array_equal(zq.numpy(), golden["a5b_zq"])


def test_elic_encoder(golden, synth_sd):
    x = t(golden["a4_x"])
    feat = O.onehot_feat(synth_sd, t(golden["a5_zq"]), t(golden["a5_idx"]))
    for q in (0, 3):
        y = O.elic_encoder(synth_sd, x, feat, O.SELECTED_BETA_RATE[q], O.SELECTED_BETA_VQ[q])
        close(y, golden[f"a6_y_q{q}"])


def test_beta_tables(golden, synth_sd):
    assert list(golden["selected_beta_rate"]) == O.SELECTED_BETA_RATE
    assert list(golden["selected_beta_vq"]) == O.SELECTED_BETA_VQ
    for name in ("encoder", "decoder"):
        for q in range(5):
            c = O.beta_cond(synth_sd, name, O.SELECTED_BETA_RATE[q], O.SELECTED_BETA_VQ[q])
            close(c.reshape(-1), golden[f"cond_{name}"][q], rtol=1e-5, atol=1e-6)


def test_elic_encoder_per_sample_beta(golden, synth_sd):
    g = torch.Generator().manual_seed(14)
    x = torch.rand((2, 3, 64, 64), generator=g) * 2 - 1
    g = torch.Generator().manual_seed(15)
    feat = torch.randn((2, 260, 8, 8), generator=g) * 0.3
    y = O.elic_encoder(synth_sd, x, feat, torch.tensor([2.29, 0.62]), torch.tensor([3.0, 1.5]))
    close(y, golden["a6b_y"])


def test_hyper_nets(golden, synth_sd):
    close(O.hyper_encoder(synth_sd, t(golden["a6_y_q0"])), golden["a7_z"])
    close(O.hyper_decoder(synth_sd, t(golden["a9_zhat"])), golden["a9_out"])


def test_elic_decoder_feats(golden, synth_sd):
    f1, fd = O.elic_decoder_feats(synth_sd, t(golden["a14_yhat"]), O.SELECTED_BETA_RATE[1], O.SELECTED_BETA_VQ[1])
    close(f1, golden["a14_feat1"], rtol=1e-3, atol=1e-3)
    assert torch.equal(fd["block_1_8"], f1)
    close(fd["block_1_4"][:, :, :8, :8], golden["a14_b14_crop"], rtol=1e-3, atol=1e-3)
    close(fd["block_1_2"][:, :, 10:18, 20:28], golden["a14_b12_crop"], rtol=1e-3, atol=1e-3)
    close(summ(fd["block_1_4"]), golden["a14_b14_sum"], rtol=1e-4, atol=1e-2)
    close(summ(fd["block_1_2"]), golden["a14_b12_sum"], rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("tag", ["a15", "a15b"])
def test_swin_estimator(golden, synth_sd, tag):
    pe, lg = O.swin_estimator(synth_sd, t(golden[f"{tag}_feat"]))
    close(pe, golden[f"{tag}_pred_embed"], rtol=1e-3, atol=1e-3)
    close(lg[:, ::16, :4, :4], golden[f"{tag}_logits_crop"], rtol=1e-3, atol=1e-3)
    am = lg.argmax(1).numpy()
    assert (am == golden[f"{tag}_argmax"]).mean() > 0.999


def test_fusion_decoder(golden, synth_sd):
    lat = O._conv(synth_sd, "vq_model.post_quant_conv", O.vq_indices_to_latent(synth_sd, t(golden["a17_idx"])))
    close(lat, golden["a17_lat"], rtol=1e-5, atol=1e-7)
    cf = {k: t(golden[f"a17_{k}"]) for k in ("block_1_8", "block_1_4", "block_1_2")}
    out = O.fusion_decode(synth_sd, lat, cf)
    close(out[:, :, 16:48, 30:62], golden["a17_out_crop"], rtol=1e-3, atol=1e-3)
    close(out[:, :, ::4, ::4], golden["a17_out_ds"], rtol=1e-3, atol=1e-3)
    close(summ(out), golden["a17_out_sum"], rtol=1e-4, atol=1e-2)
    plain = O.fusion_decode({k: v for k, v in synth_sd.items() if not k.startswith("fusion_module.")}, lat, {})
    close(plain[:, :, ::4, ::4], golden["a17p_out_ds"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_charm_vs_reference_module(charm_golden, synth_sd, tag):
    """a10: oracle charm_forward vs the reference's own Minnen20CharmContextModel.forward (fixture docstring in
    oracle/gen_golden.py: em = (round(y - mu) + mu, ones)).  Same torch CPU ops -> every slice's mu / sigma / LRP
    and the final y_hat agree to reduction-order slack; the rounding decisions must be identical."""
    G = charm_golden
    r = O.charm_forward(synth_sd, t(G[f"{tag}_y"]), t(G[f"{tag}_hyper_out"]))
    close(r["mu"], G[f"{tag}_mu"]); close(r["sigma"], G[f"{tag}_sigma"])
    sym_ref = np.round(G[f"{tag}_y"] - G[f"{tag}_mu"])
    assert np.array_equal(r["symbols"].numpy(), sym_ref.astype(np.int32))
    close(r["lrp"], 0.5 * np.tanh(G[f"{tag}_lrp"].astype(np.float64)))
    close(r["y_hat"], G[f"{tag}_y_hat"])


def test_charm_manifest_matches_reference_module(manifest):
    """The context-model keys/shapes the synthetic-weight generator derives from the YAML equal the reference
    module's own state dict (now part of the reference-produced manifest)."""
    from dc_vic_amd.synth import charm_manifest
    cm = charm_manifest()
    ref = {k: v for k, v in manifest.items() if k.startswith("context_model.")}
    assert len(ref) == 108 and cm == ref


def test_wire_format():
    with open(os.path.join(os.path.dirname(__file__), "golden", "wire_format.json")) as f:
        W = json.load(f)
    assert O.header_encode(512, 768, 37.9, 0).hex() == W["hdr_512_768_37p9_q0"] == "000200032500"
    assert O.header_encode(256, 256, 3.99, 4).hex() == W["hdr_256_256_3p99_q4"] == "000100010304"
    assert O.header_encode(1, 65535, 0.2, 2).hex() == W["hdr_1_65535_0_q2"]
    strings = [bytes.fromhex(W["hdr_512_768_37p9_q0"]), b"\x01\x02\x03", b"\xaa" * 5]
    assert O.pack_strings(strings).hex() == W["container"]
    assert [s.hex() for s in O.unpack_strings(bytes.fromhex(W["container"]))] == W["container_loaded"]
    assert O.pack_strings([b"", b"\x07"]).hex() == W["container_empty_first"]
    d = O.header_decode(bytes.fromhex(W["hdr_512_768_37p9_q0"]))
    assert list(d["img_size"]) == W["hdr_decode_512_768"]["img_size"]
    assert d["max_sample"] == W["hdr_decode_512_768"]["max_sample"] and d["quality_ind"] == 0


# ------------------------------------------------------------------------------------------- a20: PatchGAN + losses (train.npz)
def test_train_oracle_discriminator_and_losses_vs_reference_modules(train_golden):
    """oracle/train_oracle.py against the reference's own DualBetaCondTamingNLayerDiscriminator / VanillaGANLoss / MSELoss /
    VanillaMSELoss / CrossEntropyLoss (config/exp1_stage1_3.yaml kwargs): D logits for per-sample and scalar betas, every loss
    value calc_g_loss / calc_d_loss forms from them, and d(adv)/d(image) through the discriminator."""
    import torch.nn.functional as F
    from conftest import train_golden_disc_state
    from oracle import train_oracle as T
    G = train_golden
    dsd = train_golden_disc_state(G)
    kw = json.loads(str(G["d_kwargs"]))
    assert (kw["input_nc"], kw["n_layers"], kw["ndf"], kw["norm_type"], kw["cond_ch"], kw["L"]) == (11, 3, 64, "none", 8, 10)
    assert (kw["max_beta_1"], kw["max_beta_2"], kw["use_pi"], kw["include_x"]) == (O.MAX_BETA_1, O.MAX_BETA_2, False, True)
    lk = json.loads(str(G["loss_kwargs"]))
    assert {k: v["loss_weight"] for k, v in lk.items()} == {"distortion_loss": 50, "perceptual_loss": 1.0, "gan_loss": 0.01,
                                                            "code_distortion_loss": 1.0, "code_ce_loss": 0.5}
    assert T.LOSS_W == dict(distortion=50.0, perceptual=1.0, gan=0.01, code_distortion=1.0, code_ce=0.5)
    real, fake, b1, b2 = t(G["real"]), t(G["fake"]), t(G["beta_1"]), t(G["beta_2"])
    fk = fake.clone().requires_grad_(True)
    g_fake = T.discriminator(dsd, fk, b1, b2)
    close(g_fake.detach(), G["d_fake_logits"], rtol=1e-5, atol=1e-6)
    adv = T.LOSS_W["gan"] * F.binary_cross_entropy_with_logits(g_fake, torch.ones_like(g_fake))     # generator_losses' "adv" term
    adv.backward()
    close(adv.detach(), G["adv_loss"], rtol=1e-6, atol=0)
    close(fk.grad, G["adv_grad_fake"], rtol=1e-4, atol=1e-9)
    l_real, l_fake, d_real, d_fake = T.discriminator_losses(dsd, real, fake, b1, b2)
    close(d_real.detach(), G["d_real_logits"], rtol=1e-5, atol=1e-6)
    close(l_real.detach(), G["d_loss_real"], rtol=1e-6, atol=0)
    close(l_fake.detach(), G["d_loss_fake"], rtol=1e-6, atol=0)
    close(T.discriminator(dsd, real, 1.51, 2.25).detach(), G["d_real_logits_scalar_beta"], rtol=1e-5, atol=1e-6)
    # the remaining terms exactly as generator_losses writes them
    close(T.LOSS_W["distortion"] * F.mse_loss((real + 1.0) / 2.0, (fake + 1.0) / 2.0), G["distortion_loss"], rtol=1e-6, atol=0)
    close(T.LOSS_W["code_distortion"] * F.mse_loss(t(G["code_a"]), t(G["code_b"])), G["code_distortion_loss"], rtol=1e-6, atol=0)
    close(T.LOSS_W["code_ce"] * F.cross_entropy(t(G["ce_logits"]), t(G["ce_target"])), G["ce_loss"], rtol=1e-6, atol=0)
