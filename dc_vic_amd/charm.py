"""Channel-wise autoregressive (CHARM) context model on HIP kernels, GPU-resident on BOTH the
encoder and the decoder side.

Mirrors src/models/subnet/context_model/minnen20_charm_context_model.py:19-218 (SliceTransform,
Minnen20CharmContextModel.forward / forward_compress / forward_decompress).  The reference moves
this network (and the hyper-decoder) to the CPU so that encoder and decoder derive identical
entropy parameters (hyperprior_dc_vic_model.py:70-73); here the same guarantee comes from the
kernels being deterministic and batch-invariant (fixed reduction order, no atomics), so the six
sequential slice steps stay on the GPU and only int32 symbols / cdf indexes cross PCIe.
torch.cat of supports is replaced by multi-source convolutions reading channel slices in place.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .entropy import GaussianMeanScaleConditional, host_threads
from .layers import Act, Conv2d
from .registry import CONTEXTMODEL_REGISTRY

Tensor = torch.Tensor


class SliceTransform(nn.Module):
    """minnen20_charm_context_model.py:19-31."""

    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        self.model = nn.Sequential(Conv2d(in_ch, 224, 5, 1, 2), Act(), Conv2d(224, 128, 5, 1, 2), Act(), Conv2d(128, out_ch, 3, 1, 1))

    def forward(self, srcs, out: Optional[Tensor] = None, act: int = ops.ACT_NONE) -> Tensor:
        x = self.model[0](srcs, act=ops.ACT_RELU)
        x = self.model[2](x, act=ops.ACT_RELU)
        return self.model[4](x, out=out, act=act)

    def _rest_plan(self, n_lead: int):
        """First conv restricted to input channels [n_lead:] (the leading hyperprior channels are handled by a
        stacked launch, see Minnen20CharmContextModel._hyper_partials); None when nothing is left."""
        c0 = self.model[0]
        key = (c0.weight.data_ptr(), c0.weight._version, n_lead)
        if getattr(self, "_rest_key", None) != key:
            w = c0.weight.detach()
            self._rest = ops.ConvPlan(w[:, n_lead:].contiguous(), c0.bias, "conv", pad=(2, 2)) if w.shape[1] > n_lead else None
            self._zero = torch.zeros((1, w.shape[0]), dtype=torch.float32, device=w.device)
            self._rest_key = key
        return self._rest

    def forward_from_partial(self, partial: Tensor, n_lead: int, rest_srcs, out: Optional[Tensor] = None, act: int = ops.ACT_NONE) -> Tensor:
        """Same result, bit for bit, as forward(cat[lead, rest]): `partial` holds the first conv's accumulators after
        the `n_lead` leading input channels (no bias); the remaining channels continue the same fma chain."""
        plan = self._rest_plan(n_lead)
        if plan is not None:
            x = plan(rest_srcs, act=ops.ACT_RELU, init=partial)
        else:
            c0 = self.model[0]
            x = ops.chan_affine(partial, self._zero, c0.bias.detach().view(1, -1).contiguous())
            x = ops.activation(x, ops.ACT_RELU, out=x)
        x = self.model[2](x, act=ops.ACT_RELU)
        return self.model[4](x, out=out, act=act)


@CONTEXTMODEL_REGISTRY.register()
class Minnen20CharmContextModel(nn.Module):
    def __init__(self, num_slices: int, bottleneck_y: int, hyper_out_ch: int, max_support_slices: int = 5):
        super().__init__()
        assert bottleneck_y % num_slices == 0
        assert (max_support_slices == -1) or (1 <= max_support_slices <= num_slices)
        slice_ch = bottleneck_y // num_slices
        hm = hyper_out_ch // 2
        self.slice_ch, self.num_slices, self.max_support_slices = slice_ch, num_slices, max_support_slices
        self.hyper_half = hm
        self.mean_slice_transforms = nn.ModuleList()
        self.scale_slice_transforms = nn.ModuleList()
        self.lrp_slice_transforms = nn.ModuleList()
        for i in range(num_slices):
            ns = i if max_support_slices == -1 else min(i, max_support_slices)
            sup = slice_ch * ns
            self.mean_slice_transforms.append(SliceTransform(sup + hm, slice_ch))
            self.scale_slice_transforms.append(SliceTransform(sup + hm, slice_ch))
            self.lrp_slice_transforms.append(SliceTransform(sup + hm + slice_ch, slice_ch))

    def _n_support(self, i: int) -> int:
        return i if self.max_support_slices < 0 else min(i, self.max_support_slices)

    def _hyper_partials(self, hyper_mean: Tensor, hyper_scale: Tensor):
        """The hyperprior channels lead the input of every transform's first conv and do not depend on decoded
        slices, so their contribution is computed for all 3 x num_slices transforms by two large stacked launches
        instead of 18 small ones inside the sequential slice loop.  The accumulators are handed to the per-slice
        convs through `init`, which continue the reduction over the support channels in the same order: results
        are bit-identical to the unsplit convolution."""
        ms, ss, ls = self.mean_slice_transforms, self.scale_slice_transforms, self.lrp_slice_transforms
        key = tuple((t.model[0].weight.data_ptr(), t.model[0].weight._version) for t in list(ms) + list(ss) + list(ls))
        hm = self.hyper_half
        if getattr(self, "_hp_key", None) != key:
            wm = torch.cat([t.model[0].weight.detach()[:, :hm] for t in list(ms) + list(ls)], 0).contiguous()
            ws = torch.cat([t.model[0].weight.detach()[:, :hm] for t in ss], 0).contiguous()
            self._hp_mean = ops.ConvPlan(wm, None, "conv", pad=(2, 2))
            self._hp_scale = ops.ConvPlan(ws, None, "conv", pad=(2, 2))
            self._hp_key = key
        return self._hp_mean(hyper_mean), self._hp_scale(hyper_scale)

    def run(self, y: Optional[Tensor], hyper_out: Tensor, entropy_model_y: GaussianMeanScaleConditional,
            symbols_in: Optional[Callable[[int, Tensor], Tensor]] = None, want_likelihood: bool = True,
            want_symbols: bool = False, bits_out: Optional[Tensor] = None):
        """One routine for forward(eval) / forward_compress / forward_decompress.

        encode side (y given): symbols = rint(y - mu); decode side: `symbols_in(i, indexes_i)` returns
        the int32 symbols of slice i (device tensor [N, slice_ch, H, W]).
        Returns dict(y_hat, y_likelihood, symbols, indexes, mu, sigma)."""
        N, _, H, W = hyper_out.shape
        dev = hyper_out.device
        sc, ns = self.slice_ch, self.num_slices
        Cy = sc * ns
        hm = self.hyper_half
        hyper_mean, hyper_scale = hyper_out[:, :hm], hyper_out[:, hm:]
        y_hat = torch.empty((N, Cy, H, W), dtype=torch.float32, device=dev)
        ms = torch.empty((N, 2 * Cy, H, W), dtype=torch.float32, device=dev)       # [mu | sigma]
        lik = torch.empty((N, Cy, H, W), dtype=torch.float32, device=dev) if want_likelihood else None
        need_sym = want_symbols or symbols_in is not None
        sym = torch.empty((N, Cy, H, W), dtype=torch.int32, device=dev) if need_sym else None
        idx = torch.empty((N, Cy, H, W), dtype=torch.int32, device=dev) if need_sym else None
        table = entropy_model_y._table_dev(hyper_out)
        main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        side = None
        if main is not None and os.environ.get("DCVIC_CHARM_STREAMS", "1") != "0":
            if getattr(self, "_side_stream", None) is None or self._side_stream.device != dev:
                self._side_stream = torch.cuda.Stream(device=dev)
            side = self._side_stream
        split = os.environ.get("DCVIC_CHARM_SPLIT", "1") != "0"
        if split:
            Pm, Ps = self._hyper_partials(hyper_mean, hyper_scale)
            c1 = self.mean_slice_transforms[0].model[0].out_channels      # 224
        for i in range(ns):
            sl = slice(i * sc, (i + 1) * sc)
            k = self._n_support(i)
            support = [y_hat[:, : k * sc]] if k > 0 else []
            mu = ms[:, sl]
            sigma = ms[:, Cy + i * sc: Cy + (i + 1) * sc]
            # the mean and scale transforms of a slice are independent: run them on two HIP streams so the
            # small 16x16-map launches of both fill the chip together
            if side is not None:
                ev_fork = torch.cuda.Event(); ev_fork.record(main)
                side.wait_event(ev_fork)
                with torch.cuda.stream(side):
                    if split:
                        self.scale_slice_transforms[i].forward_from_partial(Ps[:, i * c1:(i + 1) * c1], hm, support, out=sigma)
                    else:
                        self.scale_slice_transforms[i]([hyper_scale] + support, out=sigma)
                    ev_join = torch.cuda.Event(); ev_join.record(side)
                if split:
                    self.mean_slice_transforms[i].forward_from_partial(Pm[:, i * c1:(i + 1) * c1], hm, support, out=mu)
                else:
                    self.mean_slice_transforms[i]([hyper_mean] + support, out=mu)
                main.wait_event(ev_join)
            elif split:
                self.mean_slice_transforms[i].forward_from_partial(Pm[:, i * c1:(i + 1) * c1], hm, support, out=mu)
                self.scale_slice_transforms[i].forward_from_partial(Ps[:, i * c1:(i + 1) * c1], hm, support, out=sigma)
            else:
                self.mean_slice_transforms[i]([hyper_mean] + support, out=mu)
                self.scale_slice_transforms[i]([hyper_scale] + support, out=sigma)
            yq = torch.empty((N, sc, H, W), dtype=torch.float32, device=dev)    # round(y-mu)+mu before the LRP
            if y is not None:
                ops.gaussian_rate(y[:, sl], None, mu, sigma, table, yq, sym[:, sl] if need_sym else None,
                                  idx[:, sl] if need_sym else None, lik[:, sl] if lik is not None else None, bits_out)
            else:
                # index-only pass (sym_in is a placeholder here: no y_hat / likelihood output is requested)
                ops.gaussian_rate(None, sym[:, sl], mu, sigma, table, None, None, idx[:, sl], None, None)
                s_i = symbols_in(i, idx[:, sl])
                sym[:, sl] = s_i
                ops.gaussian_rate(None, sym[:, sl], mu, sigma, table, yq, None, None, None, None)
            # latent residual predictor: y_hat_i = yq + 0.5 tanh(lrp(cat[mean_support, yq]))
            if split:
                lrp = self.lrp_slice_transforms[i].forward_from_partial(Pm[:, (ns + i) * c1:(ns + i + 1) * c1], hm, support + [yq],
                                                                        act=ops.ACT_HALF_TANH)
            else:
                lrp = self.lrp_slice_transforms[i]([hyper_mean] + support + [yq], act=ops.ACT_HALF_TANH)
            ops.add(yq, lrp, out=y_hat[:, sl])
        return dict(y_hat=y_hat, y_likelihood=lik, symbols=sym, indexes=idx, mu=ms[:, :Cy], sigma=ms[:, Cy:])

    # --- reference-named entry points ------------------------------------------------------------
    def forward(self, y: Tensor, hyper_out: Tensor, entropy_model_y, is_train: bool = False, calc_q_likelihood: bool = True,
                bits_out: Optional[Tensor] = None):
        if is_train:
            raise NotImplementedError("dc_vic_amd implements the inference path only")
        r = self.run(y, hyper_out, entropy_model_y, want_likelihood=True, bits_out=bits_out)
        if calc_q_likelihood:
            return r["y_hat"], r["y_likelihood"], r["y_likelihood"]   # eval: both likelihoods are the quantised one
        return r["y_hat"], r["y_likelihood"]

    def forward_compress(self, y: Tensor, hyper_out: Tensor, entropy_model_y, bits_out: Optional[Tensor] = None
                         ) -> Tuple[List[bytes], Tensor, Tensor]:
        r = self.run(y, hyper_out, entropy_model_y, want_likelihood=True, want_symbols=True, bits_out=bits_out)
        N = y.shape[0]
        s = r["symbols"].reshape(N, -1).cpu().numpy()
        ix = r["indexes"].reshape(N, -1).cpu().numpy()
        y_str = entropy_model_y.tables().encode(s, ix, threads=host_threads())
        return y_str, r["y_hat"], r["y_likelihood"]

    def forward_decompress(self, y_str, hyper_out: Tensor, entropy_model_y) -> Tuple[Tensor, Tensor]:
        """y_str: bytes (one image) or a list of bytes (one stream per batch item)."""
        streams = [y_str] if isinstance(y_str, (bytes, bytearray)) else list(y_str)
        N = hyper_out.shape[0]
        assert len(streams) == N
        dec = entropy_model_y.tables().decoders(streams)
        thr = host_threads()

        def pull(i: int, indexes: Tensor) -> Tensor:
            ix = indexes.reshape(N, -1).cpu().numpy()
            out = dec.decode(ix, threads=thr)
            return torch.from_numpy(out).to(hyper_out.device).view(indexes.shape)

        try:
            r = self.run(None, hyper_out, entropy_model_y, symbols_in=pull, want_likelihood=False)
        finally:
            dec.close()
        return r["y_hat"], r["symbols"]
