"""The fp32 parity mode of the engine on the bf16-plane kernels (UMR_F32_X3 / UMR_F32_X3_FAST).

The reference computes in fp32 (object_reasoning.py:74; no autocast in train_objectness_net.py:81,259-260,836), and fp32 is
the mode that carries the 1e-4 / bit-exact-peak contract.  Every f32 value that is a GEMM operand travels as three bf16 planes
(x = h + m + l, lossless; include/umr.h UMR_BF16X3): activations leave the producing kernel already split (GEMM epilogues,
LayerNorm), weights are kept as planes per parameter version, and all Linear / 1x1 / 3x3 layers -- forward, data gradient and
weight gradient -- run on the persistent 256x256 bf16 kernels (csrc/gemm_nt256p.hip X3, csrc/gemm_tn256.hip X3) as six plane-pair
products per f32 product.  Tensors that feed non-GEMM kernels (LayerNorm, attention, resizes, the loss) stay f32.

`XT` carries a value in either or both formats and converts on demand (exactly: split3 / unsplit3), so a layer whose shape the
plane kernels do not take (K not a multiple of 64: the narrow reassemble layers of the small backbones; stride-2 conv; row-gathered
operands) falls back to the 128x128 f32 kernels of csrc/gemm_nt.hip / gemm_tn.hip without the caller noticing.

Same launch-list structure, op order and reference citations as engine.Engine.forward / backward (models/dpt/vit.py:165-201,
86-90,104-145,259-336; blocks.py:290-313,362-383; models.py:74-94; objectness_net.py:109-135,167-183)."""
import torch

from . import _lib as L
from . import ops


class XT:
    """an [..., n] f32 value as an f32 tensor, as a planes tensor ([..., 3n] bf16), or both.  Views share ONE cell, so a conversion
    made through any view (split3 / unsplit3, exact both ways) is made once."""
    __slots__ = ("cell", "shape")

    def __init__(self, f=None, p=None, _cell=None, _shape=None):
        if _cell is not None:
            self.cell, self.shape = _cell, tuple(_shape)
            return
        assert (f is None) != (p is None) and (f if f is not None else p).is_contiguous()
        self.cell = [f, p]
        self.shape = tuple(f.shape) if f is not None else tuple(p.shape[:-1]) + (p.shape[-1] // 3,)

    @property
    def n(self):
        return self.shape[-1]

    @property
    def f(self):
        return None if self.cell[0] is None else self.cell[0].view(self.shape)

    @property
    def p(self):
        return None if self.cell[1] is None else self.cell[1].view(self.shape[:-1] + (3 * self.shape[-1],))

    def F(self):
        if self.cell[0] is None:
            self.cell[0] = ops.unsplit3(self.cell[1])
        return self.f

    def P(self):
        if self.cell[1] is None:
            self.cell[1] = ops.split3(self.cell[0])
        return self.p

    def drop_f(self):
        """the f32 copy is not needed again (saved activations that only feed plane GEMMs)"""
        if self.cell[1] is not None:
            self.cell[0] = None
        return self

    def any(self):
        """whichever format exists (epilogue operands are taken in either)"""
        return self.f if self.cell[0] is not None else self.p

    def view(self, *shape):
        return XT(_cell=self.cell, _shape=shape)

    def mask_source(self):
        """a tensor whose sign is the value's (oracle/mask_parity.py reads the ReLU decisions from saved activations)"""
        return self.f if self.cell[0] is not None else self.p[..., :self.n]


def _any(t):
    return t.any() if isinstance(t, XT) else t


def _f(t):
    return t.F() if isinstance(t, XT) else t


class X3Path:
    """mixed into engine.Engine: forward_x3 / backward_x3 and their helpers"""

    # ------------------------------------------------------------------ helpers
    def _x3_ok(self, K, N, conv):
        return conv in (0, 1) and K % 64 == 0 and N % 8 == 0

    def _mm(self, P, A, wname, kind, bias=None, *, conv=0, act=L.ACT_NONE, want="f", mask=None, dgelu=None, aux=None, aux2=None,
            rowbias=None, rows_per_batch=0, c2_mode=0, c2_want="f", out=None, c_remap=None, aux_mod=0):
        """epi(A . W^T) with W = parameter `wname` in layout `kind` (engine._w kinds; conv = 3x3 mode).  A: XT ([M, K] or NHWC).
        want / c2_want: 'f' (f32) or 'p' (planes) for the output / the second output.  Returns XT, or (XT, XT) with c2_mode."""
        w = P[wname]
        Kc = A.shape[-1]            # K of a plain GEMM, Cin of a conv
        if kind in ("lin", "lin_a", "lin_b", "c3", "ct_d"):
            N = w.shape[0]
        elif kind == "lin_t":
            N = w[0].numel()        # [N0, K0] or a 1x1 conv weight [co, ci, 1, 1]
        elif kind in ("lin_t_a", "lin_t_b"):
            N = w.shape[1] // 2
        elif kind == "c3_d":
            N = w.shape[1]
        else:                       # "ct": ConvTranspose2d [ci, co, s, s] as a GEMM onto (i, j, co)
            N = w.shape[1] * w.shape[2] * w.shape[3]
        if self._x3_ok(Kc, N, conv):
            Ap = A.P()
            r = ops.gemm_nt_x3(Ap, self._wx3(P, wname, kind), bias, act=act, conv=conv, out_planes=(want == "p"), out=out,
                               mask=_any(mask), dgelu=_any(dgelu), aux=_any(aux), aux2=_any(aux2), rowbias=rowbias,
                               rows_per_batch=rows_per_batch, c2_mode=c2_mode, c2_planes=(c2_want == "p"), c_remap=c_remap, aux_mod=aux_mod)
            wrap = lambda t: XT(p=t) if t.dtype == torch.bfloat16 else XT(f=t)
            return (wrap(r[0]), wrap(r[1])) if c2_mode else wrap(r)
        # shapes the plane kernels do not take: the 128x128 f32 kernels (in-register splits), f32 operands
        Wf = self._w_f32(P, wname, kind)
        a_t = mask if mask is not None else (dgelu if dgelu is not None else aux)
        r = ops.gemm_nt(A.F(), Wf, bias, out=out, aux=_f(a_t), aux2=_f(aux2), rowbias=rowbias, rows_per_batch=rows_per_batch, act=act,
                        mask_relu=mask is not None, mask_dgelu=dgelu is not None, c2_mode=c2_mode, conv=conv, c_remap=c_remap, aux_mod=aux_mod)
        return (XT(f=r[0]), XT(f=r[1])) if c2_mode else XT(f=r)

    def _w_f32(self, P, name, kind):
        """f32 kernel-layout weights for the fallback path (the halves of the readout weight are slices of the full pack)"""
        if kind in ("lin_a", "lin_b"):
            w = self._w(P, name, "lin")
            D = w.shape[1] // 2
            return w[:, :D] if kind == "lin_a" else w[:, D:]
        if kind in ("lin_t_a", "lin_t_b"):
            from .engine import _pack_linear_t
            w = P[name].detach()
            D = w.shape[1] // 2
            half = w[:, :D] if kind == "lin_t_a" else w[:, D:]
            return self.cache.get((name, kind, torch.float32), P[name], lambda: _pack_linear_t(half, torch.float32))
        return self._w(P, name, kind)

    def _wgrad(self, dY, X, dW, dbias=None, *, conv=0, accumulate=False, wg=None, then=None):
        """dW (+)= dY^T X (X NHWC with conv); plane kernel where it applies (N, K multiples of 8, no stride-2 conv).
        wg: engine.WgradStream -- the launch (and `then(dW)`, e.g. the unpacking of a conv gradient) goes to the weight-gradient
        stream; the operand conversions stay on the main stream, where the data-gradient GEMMs share them."""
        N = dY.shape[-1]
        Kc = X.shape[-1]
        x3 = conv in (0, 1) and N % 8 == 0 and Kc % 8 == 0
        a, b = (dY.P().reshape(-1, 3 * N), X.P()) if x3 else (dY.F().reshape(-1, N), X.F())
        res = []

        def launch():
            r = ops.gemm_tn(a, b, dW=dW, dbias=dbias, conv=conv, accumulate=accumulate, x3=x3)
            if then is not None:
                then(r)
            res.append(r)
        if wg is None or not wg.on:
            launch()
            return res[0]
        wg.run(launch, a, b)
        return None

    # ------------------------------------------------------------------ forward
    def forward_x3(self, P, images, save, skip=()):
        from .engine import _COMMUTE_RESIZE
        cfg = self.cfg
        assert images.is_cuda and images.dtype == torch.float32 and images.dim() == 4 and images.shape[1] == 3
        images = images.contiguous()
        B, _, H, W = images.shape
        p, D, heads = cfg["patch"], cfg["D"], cfg["heads"]
        gh, gw = H // p, W // p
        assert gh >= 1 and gw >= 1
        g, Nt = gh * gw, gh * gw + 1
        m = "backbone.pretrained.model."
        dev = images.device
        S = {"B": B, "H": H, "W": W, "gh": gh, "gw": gw, "x3": True} if save else None
        mm = lambda *a, **k: self._mm(P, *a, **k)
        b_ = lambda name: self._f32(P, name)

        # ---- patch embed + cls + pos (vit.py:168-193)
        G = cfg["pos_grid"]
        pos = b_(m + "pos_embed")[0]
        if (gh, gw) != (G, G):
            pos_grid = ops.bilinear_fwd(pos[1:].reshape(1, G, G, D), gh, gw, False).reshape(g, D)
        else:
            pos_grid = pos[1:]
        K = 3 * p * p
        ldk = (K + 7) // 8 * 8
        patches = XT(f=ops.patchify(images, p, torch.float32, ldk))
        tokens = torch.empty((B * Nt, D), dtype=torch.float32, device=dev)
        if ldk == K and self._x3_ok(K, D, 0):
            mm(patches, m + "patch_embed.proj.weight", "lin", b_(m + "patch_embed.proj.bias"), out=tokens, aux=pos_grid.contiguous(),
               aux_mod=g, c_remap=(g, Nt, 1))
        else:
            wp = self._w(P, m + "patch_embed.proj.weight", "lin")
            if ldk != K:
                wpad = torch.zeros((D, ldk), dtype=torch.float32, device=dev)
                wpad[:, :K] = wp
                wp = wpad
            ops.gemm_nt(patches.F(), wp, b_(m + "patch_embed.proj.bias"), out=tokens, aux=pos_grid.contiguous(), aux_mod=g, c_remap=(g, Nt, 1))
        ops.fill_cls(tokens, b_(m + "cls_token").reshape(-1), pos[0].contiguous(), B, Nt * D, D)
        if save:
            S["patches"] = patches

        # ---- transformer blocks (timm Block; only up to the last hooked block: later ones feed nothing, vit.py:107)
        x = tokens
        acts, blocks = [], []
        for i in range(max(cfg["hooks"]) + 1):
            b = m + f"blocks.{i}."
            ln1p, mean1, rstd1 = ops.layernorm_fwd(x, b_(b + "norm1.weight"), b_(b + "norm1.bias"), planes=True)
            ln1 = XT(p=ln1p)
            qkv = mm(ln1, b + "attn.qkv.weight", "lin", b_(b + "attn.qkv.bias")).F()
            att_f, lse = ops.attention_fwd(qkv, B, Nt, heads, need_lse=save)
            att = XT(f=att_f)
            x1 = mm(att, b + "attn.proj.weight", "lin", b_(b + "attn.proj.bias"), aux=x).F()
            ln2p, mean2, rstd2 = ops.layernorm_fwd(x1, b_(b + "norm2.weight"), b_(b + "norm2.bias"), planes=True)
            ln2 = XT(p=ln2p)
            if save:
                h, hpre = mm(ln2, b + "mlp.fc1.weight", "lin", b_(b + "mlp.fc1.bias"), act=L.ACT_GELU, c2_mode=2, want="p")
            else:
                h, hpre = mm(ln2, b + "mlp.fc1.weight", "lin", b_(b + "mlp.fc1.bias"), act=L.ACT_GELU, want="p"), None
            x2 = mm(h, b + "mlp.fc2.weight", "lin", b_(b + "mlp.fc2.bias"), aux=x1).F()
            if save:
                blocks.append(dict(x=x, mean1=mean1, rstd1=rstd1, ln1=ln1, qkv=qkv, att=att, lse=lse, x1=x1, mean2=mean2, rstd2=rstd2,
                                   ln2=ln2, hpre=hpre, h=h))
            x = x2
            if i in cfg["hooks"]:
                acts.append(x)
        if save:
            S["blocks"] = blocks

        # ---- readout + reassemble (vit.py:86-90,104-145,259-336)
        pp = "backbone.pretrained."
        Fs = cfg["features"]
        layers, re_saved = [], []
        for k in range(4):
            a = pp + f"act_postprocess{k + 1}."
            wname = a + "0.project.0.weight"
            tok = acts[k]
            # class-token half + bias: B rows (tiny): the 128x128 f32 kernel on the parameter itself
            rb = ops.gemm_nt(tok, self._w(P, wname, "lin")[:, D:], b_(a + "0.project.0.bias"), M=B, lda=Nt * D, out_f32=True)
            tokp = XT(p=ops.split3(tok, remap=(g, Nt, 1)))     # the token rows without the class-token rows, compact
            if save:
                r, rpre = mm(tokp, wname, "lin_a", None, rowbias=rb, rows_per_batch=g, act=L.ACT_GELU, c2_mode=2, want="p")
            else:
                r, rpre = mm(tokp, wname, "lin_a", None, rowbias=rb, rows_per_batch=g, act=L.ACT_GELU, want="p"), None
            F_ = Fs[k]
            f = mm(r, a + "3.weight", "lin", b_(a + "3.bias"), want=("f" if k == 3 else "p"))   # [B*g, F]
            if k in (0, 1):
                s = 4 if k == 0 else 2
                bname = a + "4.bias"
                from .engine import _rep_bias
                brep = self.cache.get((bname, "rep", s), P[bname], lambda: _rep_bias(P[bname], s * s))
                y = mm(f, a + "4.weight", "ct", brep).F()
                lay = XT(f=ops.pixel_shuffle(y, B, gh, gw, s, F_))
            elif k == 2:
                lay = f.view(B, gh, gw, F_)
            else:
                lay = mm(f.view(B, gh, gw, F_), a + "4.weight", "c3", b_(a + "4.bias"), conv=2)
                lay = lay.view(B, (gh - 1) // 2 + 1, (gw - 1) // 2 + 1, F_)
            layers.append(lay)
            if save:
                re_saved.append(dict(r=r, rpre=rpre, f=f, tokp=tokp))
        if save:
            S["re"] = re_saved
            S["acts"] = acts

        # ---- scratch convs + refinenets (models.py:80-91, blocks.py:290-383)
        sc = "backbone.scratch."
        rn, rn_relu = [], []
        for k in range(4):
            lay = layers[k]
            o, orl = mm(lay, sc + f"layer{k + 1}_rn.weight", "c3", None, conv=1, c2_mode=1, c2_want="p")
            shp = (lay.shape[0], lay.shape[1], lay.shape[2], 256)
            rn.append(o.view(*shp))
            rn_relu.append(orl.view(*shp))
        fus_saved = {}
        path = None
        for k in (4, 3, 2, 1):
            r_ = sc + f"refinenet{k}."
            x1_, x1_relu = rn[k - 1], rn_relu[k - 1]
            nb, hh, ww, _ = x1_.shape
            shp = (nb, hh, ww, 256)
            fs = {}
            if path is None:
                s_, s_relu = x1_, x1_relu
            else:
                assert path.shape == x1_.shape, "fusion skip/size mismatch"
                t1 = mm(x1_relu, r_ + "resConfUnit1.conv1.weight", "c3", b_(r_ + "resConfUnit1.conv1.bias"), conv=1, act=L.ACT_RELU, want="p").view(*shp)
                s_, s_relu = mm(t1, r_ + "resConfUnit1.conv2.weight", "c3", b_(r_ + "resConfUnit1.conv2.bias"), conv=1, aux=x1_, aux2=path,
                                c2_mode=1, c2_want="p")
                s_, s_relu = s_.view(*shp), s_relu.view(*shp)
                fs.update(x1_relu=x1_relu, t1=t1)
            t2 = mm(s_relu, r_ + "resConfUnit2.conv1.weight", "c3", b_(r_ + "resConfUnit2.conv1.bias"), conv=1, act=L.ACT_RELU, want="p").view(*shp)
            u = mm(t2, r_ + "resConfUnit2.conv2.weight", "c3", b_(r_ + "resConfUnit2.conv2.bias"), conv=1, aux=s_).view(*shp)
            if k > 1 and cfg["patch"] != 16:
                nxt = rn[k - 2].shape
                Ho, Wo = nxt[1], nxt[2]  # patch-14 extension: resize to the next skip's size
            else:
                Ho, Wo = 2 * hh, 2 * ww      # blocks.py:377-379
            if _COMMUTE_RESIZE:
                # out_conv before the resize (engine._COMMUTE_RESIZE): a quarter of the rows, and its backward reads u
                ul = mm(u.view(nb * hh * ww, 256), r_ + "out_conv.weight", "lin", b_(r_ + "out_conv.bias"))
                path = XT(f=ops.bilinear_fwd(ul.F().view(nb, hh, ww, 256), Ho, Wo, True))
                del ul
                if save:
                    u.drop_f()      # the weight gradient reads the planes
                    fs.update(s_relu=s_relu, t2=t2, u=u, in_hw=(hh, ww))
                    fus_saved[k] = fs
                continue
            up = XT(f=ops.bilinear_fwd(u.F(), Ho, Wo, True))
            path = mm(up.view(nb * Ho * Wo, 256), r_ + "out_conv.weight", "lin", b_(r_ + "out_conv.bias")).view(nb, Ho, Wo, 256)
            if save:
                up.drop_f()     # the weight gradient reads the planes
                fs.update(s_relu=s_relu, t2=t2, up=up, in_hw=(hh, ww))
                fus_saved[k] = fs
        if cfg["patch"] == 16:
            H, W = 2 * path.shape[1], 2 * path.shape[2]   # models.py:70-72
            if save:
                S["H"], S["W"] = H, W
        # the heads' first layer before the final resize (engine._COMMUTE_RESIZE): the interpolated 256-channel map is never formed
        lowres = _COMMUTE_RESIZE
        feat = XT(f=ops.bilinear_fwd(path.F(), H, W, True)) if not lowres else None
        if save:
            S["fus"] = fus_saved
            S["rn_in"] = layers
            S["path1_hw"] = (path.shape[1], path.shape[2])
            S["feat"] = feat
            if lowres:
                S["path"] = path

        # ---- heads (objectness_net.py:109-135)
        from .engine import _ACT
        outs, heads_saved = [], []
        M = B * H * W
        feat2 = None if lowres else feat.view(M, 256)
        nbp, php, pwp = path.shape[0], path.shape[1], path.shape[2]
        for name, lay in (("center_field_prediction_head", self.center_layout), ("sdf_prediction_head", self.sdf_layout)):
            if name in skip:
                outs.append(None)
                continue
            idx = lay["conv_idx"]
            if self._collapse(lay, save):
                if lowres:
                    out, cs = self._linear_head_forward_lowres(P, name, idx, path.F(), H, W, _ACT[lay["final"]], save)
                else:
                    out, cs = self._linear_head_forward(P, name, idx, feat.F(), _ACT[lay["final"]])
                    cs["out"] = out
                outs.append(out)
                if save:
                    heads_saved.append(cs)
                continue
            act = L.ACT_RELU if lay["relu"] else L.ACT_NONE
            algebraic = save and not lay["relu"] and lay["final"] != "sine" and self.linear_head_backward == "algebraic"
            keep = save and not algebraic
            hb = lambda k: b_(f"{name}.{idx[k]}.bias")
            if lowres:
                # resize(W1 path + b1), then the ReLU: written straight as the planes the 3x3 conv stages
                h1l = mm(path.view(nbp * php * pwp, 256), f"{name}.{idx[0]}.weight", "lin", hb(0))
                h1 = XT(p=ops.bilinear_fwd(h1l.F().view(nbp, php, pwp, -1), H, W, True, relu=lay["relu"], planes=True)).view(M, h1l.n)
                del h1l
            else:
                h1 = mm(feat2, f"{name}.{idx[0]}.weight", "lin", hb(0), act=act, want="p")
            h2 = mm(h1.view(B, H, W, 512), f"{name}.{idx[1]}.weight", "c3", hb(1), conv=1, act=act, want="p")
            w4 = b_(f"{name}.{idx[3]}.weight")
            w4 = w4.reshape(w4.shape[0], -1)
            if not keep:
                # the 1024 -> {1,2} output layer rides in the epilogue of the layer that produces its input: h3 is never stored
                parts = ops.gemm_nt_x3(h2.P(), self._wx3(P, f"{name}.{idx[2]}.weight", "lin"), hb(2), act=act, red_w=w4.contiguous())
                out = ops.head_out_finish(parts, hb(3), B, H, W, _ACT[lay["final"]])
                del parts
                outs.append(out)
                if algebraic:
                    heads_saved.append(dict(algebraic=True, act=_ACT[lay["final"]], out=out))
                continue
            h3 = mm(h2, f"{name}.{idx[2]}.weight", "lin", hb(2), act=act).F()     # f32: the output layer and its backward read the values
            out = ops.head_out_fwd(h3, w4, hb(3), B, H, W, _ACT[lay["final"]])
            zpre = ops.head_out_fwd(h3, w4, hb(3), B, H, W, L.ACT_NONE) if lay["final"] == "sine" else None
            outs.append(out)
            heads_saved.append(dict(h1=h1, h2=h2, h3=h3, out=(zpre if zpre is not None else out)))
            del h1, h2, h3
        if save:
            S["heads"] = heads_saved
        return outs[0], outs[1], S

    # ------------------------------------------------------------------ backward
    def backward_x3(self, P, S, d_center, d_sdf, G, stage_cb=None, join_at_stages=False):
        from .engine import _ACT, _LN_PARAMS_SIDE, _unpack_conv3_grad, WgradStream
        cfg = self.cfg
        B, H, W, gh, gw = S["B"], S["H"], S["W"], S["gh"], S["gw"]
        D, heads, p = cfg["D"], cfg["heads"], cfg["patch"]
        g, Nt = gh * gw, gh * gw + 1
        dev = d_center.device
        wg = WgradStream(dev, WgradStream.wanted(B * H * W))
        # LayerNorm's dgamma / dbeta are weight gradients too: in a chain-of-graphs step their reduction pass leaves the data-gradient chain
        # for the weight-gradient lane (48 launches of the reference recipe's step; +0.7 %).  Not in the eager two-stream schedule, whose
        # host is the slower side in backward: a hand-over costs it more than the 4-us kernel costs the GPU.
        ln_via = wg.run if (wg.staged is not None and _LN_PARAMS_SIDE) else None
        mm = lambda *a, **k: self._mm(P, *a, **k)

        def cb(name):
            if join_at_stages:
                wg.join()
            if stage_cb is not None:
                stage_cb(name, wg)
            wg.stage_end()

        def wgrad_lin(name, dy, x, bias_name=None):
            self._wgrad(dy, x, G[name].view(G[name].shape[0], -1), (G[bias_name] if bias_name else None), wg=wg)

        def wgrad_c3(name, dy, x_nhwc, bias_name=None, conv=1):
            co = G[name].shape[0]
            self._wgrad(dy.view(-1, co), x_nhwc, None, (G[bias_name] if bias_name else None), conv=conv, wg=wg,
                        then=lambda dwp: _unpack_conv3_grad(dwp, G[name]))

        # ---- heads
        M = B * H * W
        lowres = S.get("path") is not None        # the heads' first layer ran before the final resize (engine._COMMUTE_RESIZE)
        feat = S["feat"]
        feat2 = None if lowres else feat.view(M, 256)
        if lowres:
            pathx = S["path"]
            nbp, php, pwp = pathx.shape[0], pathx.shape[1], pathx.shape[2]
            Ml = nbp * php * pwp
            pl = pathx.view(Ml, 256)
        dfeat = None          # f32 [M, 256] (lowres: the gradient of the map before the resize, [Ml, 256])
        for hi, (name, lay, dout) in enumerate((("center_field_prediction_head", self.center_layout, d_center),
                                                ("sdf_prediction_head", self.sdf_layout, d_sdf))):
            hs = S["heads"][hi]
            idx = lay["conv_idx"]
            if lowres and hs.get("algebraic"):
                # the head's two reductions as products on the small map (csrc/linear_head.hip, lh_shift9_kernel)
                if "u" not in hs:
                    hs["u"], hs["Vc"], hs["Kw"], _ = self._linear_head_weights(P, name, idx, dev)
                s9, nd = ops.linear_head_shift9(dout.contiguous(), hs["out"], hs["act"], torch.float32)
                E = ops.bilinear_bwd(s9, php, pwp, True).view(Ml, 16)
                del s9
                gm = ops.gemm_tn(E, pl.F())
                self._linear_head_algebra(P, name, idx, hs, gm[:9].reshape(-1), nd[:9], nd[9:10], G)
                kwt = torch.zeros((256, 16), dtype=torch.float32, device=dev)
                kwt[:, :9].copy_(hs["Kw"].view(9, 256).t())
                dfeat = ops.gemm_nt(E, kwt, None) if dfeat is None else ops.gemm_nt(E, kwt, None, aux=dfeat, out=dfeat)
                continue
            if hs.get("collapsed") or hs.get("algebraic"):
                dfeat = self._linear_head_backward(P, name, idx, feat.F().view(B, H, W, 256), hs, dout, dfeat, G)
                continue
            relu = lay["relu"]
            w4 = self._f32(P, f"{name}.{idx[3]}.weight")
            dh3 = XT(f=ops.head_out_bwd(hs["h3"], w4.reshape(w4.shape[0], -1), dout.contiguous(), hs["out"], _ACT[lay["final"]], relu,
                                        G[f"{name}.{idx[3]}.weight"].view(w4.shape[0], -1), G[f"{name}.{idx[3]}.bias"]))
            hs["h3"] = None
            wgrad_lin(f"{name}.{idx[2]}.weight", dh3, hs["h2"], f"{name}.{idx[2]}.bias")
            dh2 = mm(dh3, f"{name}.{idx[2]}.weight", "lin_t", None, mask=(hs["h2"] if relu else None), want="p")
            del dh3
            h1 = hs["h1"].view(B, H, W, 512)
            wgrad_c3(f"{name}.{idx[1]}.weight", dh2, h1, f"{name}.{idx[1]}.bias")
            hs["h2"] = None
            dh1 = mm(dh2.view(B, H, W, 512), f"{name}.{idx[1]}.weight", "c3_d", None, conv=1, mask=(hs["h1"] if relu else None),
                     want=("f" if lowres else "p"))
            del dh2
            hs["h1"] = None
            if lowres:
                dh1 = XT(f=ops.bilinear_bwd(dh1.F().view(B, H, W, dh1.n), php, pwp, True)).view(Ml, dh1.n)
            wgrad_lin(f"{name}.{idx[0]}.weight", dh1, (pl if lowres else feat2), f"{name}.{idx[0]}.bias")
            if dfeat is None:
                dfeat = mm(dh1, f"{name}.{idx[0]}.weight", "lin_t", None).F()
            else:
                dfeat = mm(dh1, f"{name}.{idx[0]}.weight", "lin_t", None, aux=dfeat).F()
            del dh1
        S["feat"] = None
        S["path"] = None
        cb("heads")
        ph, pw = S["path1_hw"]
        dpath = XT(f=dfeat.view(nbp, php, pwp, 256)) if lowres else XT(f=ops.bilinear_bwd(dfeat.view(B, H, W, 256), ph, pw, True))
        del dfeat

        # ---- refinenets + scratch convs
        sc = "backbone.scratch."
        d_rn = {}
        for k in (1, 2, 3, 4):
            r_ = sc + f"refinenet{k}."
            fs = S["fus"][k]
            hh, ww = fs["in_hw"]
            nb, Hp, Wp = dpath.shape[0], dpath.shape[1], dpath.shape[2]
            shp = (nb, hh, ww, 256)
            dp2 = dpath.view(nb * Hp * Wp, 256)
            if "u" in fs:     # out_conv ran before the resize
                dul = XT(f=ops.bilinear_bwd(dpath.F().view(nb, Hp, Wp, 256), hh, ww, True)).view(nb * hh * ww, 256)
                wgrad_lin(r_ + "out_conv.weight", dul, fs["u"].view(nb * hh * ww, 256), r_ + "out_conv.bias")
                du = mm(dul, r_ + "out_conv.weight", "lin_t", None).view(*shp)
                del dul
            else:
                wgrad_lin(r_ + "out_conv.weight", dp2, fs["up"].view(nb * Hp * Wp, 256), r_ + "out_conv.bias")
                dup = mm(dp2, r_ + "out_conv.weight", "lin_t", None).F()
                du = XT(f=ops.bilinear_bwd(dup.view(nb, Hp, Wp, 256), hh, ww, True))
                del dup
            # RCU2: u = conv2(relu(conv1(relu(s)))) + s
            wgrad_c3(r_ + "resConfUnit2.conv2.weight", du, fs["t2"], r_ + "resConfUnit2.conv2.bias")
            dt2 = mm(du, r_ + "resConfUnit2.conv2.weight", "c3_d", None, conv=1, mask=fs["t2"], want="p").view(*shp)
            wgrad_c3(r_ + "resConfUnit2.conv1.weight", dt2, fs["s_relu"], r_ + "resConfUnit2.conv1.bias")
            ds = mm(dt2, r_ + "resConfUnit2.conv1.weight", "c3_d", None, conv=1, mask=fs["s_relu"], aux2=du, want="p").view(*shp)
            del dt2, du
            if "t1" in fs:
                # s = path_prev + RCU1(x1)
                wgrad_c3(r_ + "resConfUnit1.conv2.weight", ds, fs["t1"], r_ + "resConfUnit1.conv2.bias")
                dt1 = mm(ds, r_ + "resConfUnit1.conv2.weight", "c3_d", None, conv=1, mask=fs["t1"], want="p").view(*shp)
                wgrad_c3(r_ + "resConfUnit1.conv1.weight", dt1, fs["x1_relu"], r_ + "resConfUnit1.conv1.bias")
                dx1 = mm(dt1, r_ + "resConfUnit1.conv1.weight", "c3_d", None, conv=1, mask=fs["x1_relu"], aux2=ds, want="p").view(*shp)
                del dt1
                d_rn[k] = dx1
                dpath = ds  # gradient of the previous (coarser) path
            else:
                d_rn[k] = ds
            S["fus"][k] = None

        cb("refine")
        # ---- layerK_rn + reassemble + readout; token gradients collected per hook
        pp = "backbone.pretrained."
        Fs = cfg["features"]
        d_hook = [None] * 4
        for k in range(4):
            lay_in = S["rn_in"][k]
            dr = d_rn.pop(k + 1)
            wgrad_c3(sc + f"layer{k + 1}_rn.weight", dr, lay_in, None)
            a = pp + f"act_postprocess{k + 1}."
            F_ = Fs[k]
            rs = S["re"][k]
            f = rs["f"]
            # dl: gradient of the reassembled map; f32 where a non-GEMM kernel (pixel shuffle, zero stuffing) or the stride-2
            # weight gradient reads it
            dl = mm(dr, sc + f"layer{k + 1}_rn.weight", "c3_d", None, conv=1, want=("p" if k == 2 else "f"))
            del dr
            if k in (0, 1):
                s = 4 if k == 0 else 2
                dyu = XT(f=ops.pixel_shuffle(dl.F().view(B, gh * s, gw * s, F_), B, gh, gw, s, F_, inverse=True))  # [B*g, s*s*F]
                brep = torch.empty(s * s * F_, dtype=torch.float32, device=dev)
                dwp = self._wgrad(dyu, f.view(B * g, F_), None, brep)  # [(i,j,co)][ci]
                gw_ = G[a + "4.weight"]  # [ci, co, s, s]
                ops.permute4(dwp, gw_, (F_, F_, s, s), (1, F_, s * F_ * F_, F_ * F_))
                ops.segsum(brep, 1, s * s, F_, 0, F_, out=G[a + "4.bias"].view(1, F_))
                df = mm(dyu, a + "4.weight", "ct_d", None, want="p")
                del dyu
            elif k == 2:
                df = dl.view(B * g, F_)
            else:
                ho, wo = (gh - 1) // 2 + 1, (gw - 1) // 2 + 1
                wgrad_c3(a + "4.weight", XT(f=dl.F()), XT(f=f.F().view(B, gh, gw, F_)), a + "4.bias", conv=2)
                stuffed = XT(f=ops.zero_stuff2(dl.F().view(B, ho, wo, F_), gh, gw))
                df = mm(stuffed, a + "4.weight", "c3_d", None, conv=1, want="p").view(B * g, F_)
                del stuffed
            del dl
            wgrad_lin(a + "3.weight", df, rs["r"], a + "3.bias")
            d_rpre = mm(df, a + "3.weight", "lin_t", None, dgelu=rs["rpre"], want="p")  # [B*g, D]
            del df
            wname = a + "0.project.0.weight"
            gfull = G[wname]  # [D, 2D]
            self._wgrad(d_rpre, rs["tokp"], gfull[:, :D], None)
            sB32 = ops.segsum(d_rpre.F(), B, g, D, g * D, D)  # [B, D] f32: sum over patches
            ops.segsum(sB32, 1, B, D, 0, D, out=G[a + "0.project.0.bias"].view(1, D))
            ops.gemm_tn(sB32, S["acts"][k], dW=gfull[:, D:], M=B, ldx=Nt * D)   # class-token half: B rows (tiny)
            d_hook[k] = (d_rpre, sB32, wname)
            S["re"][k] = None

        cb("reassemble")

        def add_hook_grad(k, dx):
            d_rpre, sB32, wname = d_hook[k]
            if dx is None:
                dx = torch.zeros((B * Nt, D), dtype=torch.float32, device=dev)
            mm(d_rpre, wname, "lin_t_a", None, out=dx, aux=dx, c_remap=(g, Nt, 1))
            cls_rows = dx.view(B, Nt * D)[:, :D]  # token 0 of every image: row stride Nt*D
            ops.gemm_nt(sB32, self._w_f32(P, wname, "lin_t_b"), None, out=cls_rows, aux=cls_rows)
            d_hook[k] = None
            return dx

        # ---- transformer blocks
        m = "backbone.pretrained.model."
        dx = None
        hooks = cfg["hooks"]
        for i in range(max(hooks), -1, -1):
            if i in hooks:
                dx = add_hook_grad(hooks.index(i), dx)
            b = m + f"blocks.{i}."
            bs = S["blocks"][i]
            dxx = XT(f=dx)
            wgrad_lin(b + "mlp.fc2.weight", dxx, bs["h"], b + "mlp.fc2.bias")
            dhp = mm(dxx, b + "mlp.fc2.weight", "lin_t", None, dgelu=bs["hpre"], want="p")
            wgrad_lin(b + "mlp.fc1.weight", dhp, bs["ln2"], b + "mlp.fc1.bias")
            dln2 = mm(dhp, b + "mlp.fc1.weight", "lin_t", None).F()
            del dhp, dxx
            dx1 = ops.layernorm_bwd(dln2, bs["x1"], self._f32(P, b + "norm2.weight"), bs["mean2"], bs["rstd2"],
                                    G[b + "norm2.weight"], G[b + "norm2.bias"], dres=dx, params_via=ln_via)
            del dln2
            dx1x = XT(f=dx1)
            wgrad_lin(b + "attn.proj.weight", dx1x, bs["att"], b + "attn.proj.bias")
            datt = mm(dx1x, b + "attn.proj.weight", "lin_t", None).F()
            dqkv = XT(f=ops.attention_bwd(bs["qkv"], bs["att"].F(), datt, bs["lse"], B, Nt, heads))
            del datt, dx1x
            wgrad_lin(b + "attn.qkv.weight", dqkv, bs["ln1"], b + "attn.qkv.bias")
            dln1 = mm(dqkv, b + "attn.qkv.weight", "lin_t", None).F()
            del dqkv
            dx = ops.layernorm_bwd(dln1, bs["x"], self._f32(P, b + "norm1.weight"), bs["mean1"], bs["rstd1"],
                                   G[b + "norm1.weight"], G[b + "norm1.bias"], dres=dx1, params_via=ln_via)
            del dln1, dx1
            S["blocks"][i] = None
            cb(f"block{i}")

        # ---- embeddings (vit.py:179-193)
        dpos = ops.segsum(dx, Nt, B, Nt * D, D, D)  # [Nt, D] f32, sum over images
        G[m + "cls_token"].view(-1).copy_(dpos[0])
        gpos = G[m + "pos_embed"]
        Gd = cfg["pos_grid"]
        gpos[0, 0].copy_(dpos[0])
        if (gh, gw) == (Gd, Gd):
            gpos[0, 1:].copy_(dpos[1:])
        else:
            gpos[0, 1:].copy_(ops.bilinear_bwd(dpos[1:].reshape(1, gh, gw, D).contiguous(), Gd, Gd, False).view(Gd * Gd, D))
        K = 3 * p * p
        gwp = G[m + "patch_embed.proj.weight"].view(D, K)
        patches = S["patches"]
        dxp = XT(p=ops.split3(dx, remap=(g, Nt, 1)))      # token gradients without the class-token rows
        if patches.shape[1] == K:
            self._wgrad(dxp, patches, gwp, G[m + "patch_embed.proj.bias"])
        else:
            tmp = self._wgrad(dxp, patches, None, G[m + "patch_embed.proj.bias"])
            gwp.copy_(tmp[:, :K])
        cb("embed")
        wg.join()
        S.clear()
