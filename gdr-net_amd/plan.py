"""Plans of the GDR-Net hot-path executor: static buffers + pre-bound C-ABI call lists for one (batch size, train / eval, losses)
configuration of one Engine (engine.py).  `Plan._build` walks the fixed graph once -- stem, 16 BasicBlocks, geometric head, head tail,
Patch-PnP, pose decode, losses (core/gdrn_modeling/models/GDRN.py:110-306, resnet_backbone.py:69-80, cdpn_rot_head_region.py:182-193,
conv_pnp_net.py:111-157) -- and appends one closure per kernel launch to `fwd` and, group by group in reverse, to `bwd`; `run_forward` /
`run_backward` replay them on the current HIP stream (and the engine's side stream)."""
import ctypes as C
from types import SimpleNamespace as NS

import torch

from . import cabi
from .cabi import ACC_ROWS, PREZEROED, ConvParams, PoseParams, S2Params, S2dParams, WgradParams, check, ptr
from .engine import HEAD_CONVS, RESNET34_LAYERS, RESNET34_PLANES, _ru


def _probed(op, sink):
    """op wrapped in a pair of timing events recorded on the stream it is launched on (bench.py's in-step roofline bracket)"""
    def run(st, ctx):
        s = torch.cuda.ExternalStream(st) if st else torch.cuda.default_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        parts = getattr(op, "parts", None)   # conv launch + the BatchNorm-coefficient launch behind it: the conv alone is bracketed
        a.record(s)
        (parts[0] if parts else op)(st, ctx)
        b.record(s)
        if parts:
            parts[1](st, ctx)
        sink.append((a, b, op.meta, bool(getattr(op, "side", False))))
    run.side = getattr(op, "side", False)
    return run


class Plan:
    """Static buffers + pre-bound call lists for one (batch size, train/eval) configuration."""

    def __init__(self, eng, B, bn_train, with_loss):
        self.e = eng
        self.B = B
        self.bn_train = bn_train
        self.with_loss = with_loss
        self.has_backward = bn_train and with_loss
        self.fwd = []          # callables f(stream, ctx)
        self.bwd_groups = []   # list of lists, appended in forward order, executed reversed
        self.keep = []         # keep ctypes structs alive
        self.bn = {}           # bn key -> NS(mean, invstd, scale, shift, sums)
        self.eval_prep = []        # eval-mode BN scale/shift launches (run when parameters / running stats changed)
        self._fold_hooked = False
        self._eval_sig = None
        self._unpack_pending = []  # (forward group index, layer)
        self._wreduce = {}         # layer key -> (workspace, nsplit, Cout, Cin) of the halo weight-gradient partials
        self._wgrad_deferred = []  # (forward group index, layer, WgradParams, flops) of the halo weight gradients
        self.tensors = {}          # name -> activation / gradient buffer (inspection by the tests)
        self._zero_regions = []    # fp32 tensors the backward accumulates into with atomics: cleared by ONE gdrn_zero_multi launch
        self.generation = 0        # bumped by every run_forward: a backward checks its activations are still the plan's
        self.head_valid = True     # plan.head_out holds the logits of the last forward (False: inference without cfg.TEST.USE_PNP skipped them)
        self.grad_group = {}       # parameter name -> forward index of the backward group that completes its gradient
        self._build()
        if self.has_backward:
            self._finish_unpack()
            self._build_zero_table()
        self.bwd_groups = [[op for op in g if op is not None] for g in self.bwd_groups]
        if self.has_backward:
            self._merge_forks()
        self.bwd = [op for g in reversed(self.bwd_groups) for op in g]

    # ---- op builders -------------------------------------------------------------------------
    def _conv(self, L, x, xs, y, Hi, Wi, Ho, Wo, stride, pad, mode=0, w=None, rows=None, cin=None, cout=None, x_cs=None,
              y_cs=None, bias=None, addend=None, add_cs=0, act=0, out_f32=0, stats=None, KH=None, KW=None, bnb=None, evalw=False, xf=None):
        """bnb = (bn key, raw input of that BN, stored activation or None, affine mask?): data-gradient launch whose
        output is the gradient w.r.t. that BatchNorm(+ReLU)'s output -- the halo kernel's epilogue masks it and
        accumulates the BN-backward sums (see _fusable), so the BN backward is only its apply pass.
        xf = dict(mode, x2, a, b, c, c2, msc, msh, out, relu): the conv's input is v(x, x2) evaluated while the patch is staged
        (gdrn_hip.h, xf_mode), `out` (same shape as x) receives v -- halo launches only."""
        e = self.e
        cp = ConvParams()
        cp.x, cp.w, cp.y = ptr(x), ptr(w if w is not None else L.wf), ptr(y)
        cp.bias, cp.addend, cp.stats = ptr(bias), ptr(addend), ptr(stats)
        cp.Hi, cp.Wi, cp.Cin, cp.x_cs = Hi, Wi, cin or L.cin_f, x_cs or xs
        cp.Ho, cp.Wo, cp.Cout = Ho, Wo, cout or L.O
        cp.y_cs = y_cs or y.shape[-1]
        cp.add_cs = add_cs
        cp.KH, cp.KW, cp.stride, cp.pad = KH or L.KH, KW or L.KW, stride, pad
        cp.mode, cp.act, cp.out_f32 = mode, act, out_f32
        cp.M = self.B * (Ho // 2) * (Wo // 2) if mode == 1 else self.B * Ho * Wo
        cp.w_rows = rows or L.rows_f
        cp.dtype = e.dt
        if bnb is not None and bnb[2] is not None:
            cp.bnb_mask = ptr(bnb[2])
        if xf is not None:
            cp.xf_mode = xf["mode"]     # (the library's kernel choice depends on addend, stored mask and operand transform)
        # eight-wave form of the small-map tile (gdrn_hip.h, halo_waves): the library's choice in the forward pass and in inference (r5, isolated:
        # 256@16x16 22.3 -> 21.7 us plain, 27.2 -> 24.6 with the BatchNorm-backward transform; 512@8x8 25.1 -> 23.8 / 30.5 -> 26.5; inference
        # 1.98 -> 1.96 ms); the data gradients of a two-stream backward pass keep four waves -- two 235-register waves per SIMD leave the
        # side stream's weight gradient no room on the CU (7.37 -> 7.70 ms per step with eight waves everywhere, 7.93 -> 7.85 on one stream)
        cp.halo_waves = e.halo_waves or (4 if (w is not None and (e.wgrad_stream or e.wgrad_force_lds)) else 0)   # ("serial": the two-stream step's launches)
        cp.v3_min_wg = e.v3_min_wg
        self.keep.append(cp)
        ref = C.byref(cp)
        halo_on = e.use_halo and self.B >= e.halo_min_b   # (fp32 parity mode: only plans of the BASELINE sizes, see Engine.__init__)
        if halo_on and L.kind == "conv" and L.wfF is not None and not L.s2:
            which = "e" if evalw else ("f" if w is None else "d")
            # the operand's layout is a property of the LAYER, not of whichever plan happens to be built first (a bs = 4 smoke
            # plan used to pin the first halo kernel's layout for the bs = 64 training plan): the library is asked at the canonical batch
            # size of the path (64 RoIs) unless this plan is bigger
            m_plan = cp.M
            cp.M = max(cp.M, 64 * Ho * Wo)
            cp.w_frag = e.v3_policy   # (the query reads the policy from w_frag; the answer goes back into it below)
            want = int(e.lib.gdrn_conv3x3_wfrag(ref))
            cp.M = m_plan
            if not L.wfmt.get(which):
                L.wfmt[which] = want if want in (1, 2) else 1
                e._pack_dirty = True    # the operand copy is (re)built in that layout by the next repack
            cp.w_frag = L.wfmt[which]  # an operand has ONE layout: later plans (other batch sizes) follow the first one
        # 3x3 stride-1 layers (forward and data-gradient) run on the halo-tiled kernel
        th, tw, hbn = C.c_int(0), C.c_int(0), C.c_int(0)
        e.lib.gdrn_conv3x3_tile(ref, C.byref(th), C.byref(tw), C.byref(hbn))
        use_halo = halo_on and th.value > 0 and L.kind == "conv" and L.wfF is not None and not L.s2
        if halo_on and L.kind == "conv" and L.wfF is not None and not L.s2 and not use_halo:
            raise RuntimeError(f"{L.key}: no halo tiling for {Hi}x{Wi} (the generic-layout operand copy is not maintained)")
        if evalw:      # eval-mode operand with the BatchNorm scale folded in (Engine.fold)
            assert w is None
            cp.w = ptr(L.wfF_e if use_halo else L.wf_e)
        elif use_halo:  # fragment-major operand copy (forward: of wf, data gradient: of wd)
            cp.w = ptr(L.wfF if w is None else L.wdF)
        if bnb is not None:
            assert use_halo or e.gemm_bnb, L.key
            bkey, braw, bmask, baffine = bnb
            sb = self.bn[bkey]
            cp.bnb_x, cp.bnb_mask, cp.bnb_cs = ptr(braw), ptr(bmask), braw.shape[-1]
            cp.bnb_mean, cp.bnb_invstd, cp.bnb_rows = ptr(sb.mean), ptr(sb.invstd), ptr(self.stats)  # stats scratch is idle in backward
            if baffine:
                cp.bnb_scale, cp.bnb_shift = ptr(sb.scale), ptr(sb.shift)
        if xf is not None:
            assert use_halo and e.h16, L.key
            assert xf["out"] is None or (xf["out"].shape == x.shape and xf["out"].dtype == x.dtype), L.key
            assert xf.get("x2") is None or xf["x2"].shape == x.shape, L.key
            cp.xf_mode, cp.xf_relu = xf["mode"], 1 if xf.get("relu") else 0
            cp.xf_x2, cp.xf_a, cp.xf_b, cp.xf_c, cp.xf_c2 = ptr(xf.get("x2")), ptr(xf.get("a")), ptr(xf.get("b")), ptr(xf["c"]), ptr(xf.get("c2"))
            cp.xf_msc, cp.xf_msh, cp.xf_out = ptr(xf.get("msc")), ptr(xf.get("msh")), ptr(xf.get("out"))
        fn = e.lib.gdrn_conv3x3_halo if use_halo else e.lib.gdrn_conv_gemm
        cp._stats_rows = (e.lib.gdrn_conv3x3_stats_rows if use_halo else e.lib.gdrn_conv_stats_rows)(ref)

        if bnb is not None:
            nrows_b = int(cp._stats_rows)
            assert nrows_b * 2 * cp.Cout <= self.stats.numel(), (L.key, nrows_b)
            # the epilogue's per-tile rows -> that BatchNorm's backward coefficients (+ dgamma / dbeta), for its apply pass or
            # for the conv that applies them on load
            coef = self._bn_coef_op(bnb[0], self.stats, nrows_b)

            def conv_only(st, ctx):
                s = fn(ref, st)
                if s:
                    check(s, f"conv {L.key}")

            def run(st, ctx):
                conv_only(st, ctx)
                coef(st, ctx)

            run.parts = (conv_only, coef)  # (the conv launch alone, what follows it): bench.py times the kernel, not the pair
        else:
            def run(st, ctx):
                s = fn(ref, st)
                if s:
                    check(s, f"conv {L.key}")

        # metadata for the roofline measurement in bench.py: kernel instantiation + algorithmic FLOPs
        bm, bn = C.c_int(0), C.c_int(0)
        e.lib.gdrn_conv_tile(ref, C.byref(bm), C.byref(bn))
        # true (unpadded) MACs: the layer's forward MAC count whichever direction this launch computes
        if L.kind == "stem":
            macs = self.B * 128 * 128 * 64 * 3 * 49
        elif L.kind == "fc1":
            macs = self.B * L.O * L.I * L.KK
        else:
            sp = (Ho * Wo) if (mode == 0 and (w is None or L.kind == "convT")) else (Hi * Wi)
            if L.kind == "convT":
                sp = 8 * 8  # Hin*Win*Cin*Cout*k^2
            macs = self.B * sp * L.O * L.I * L.KK
        dn = "bf16" if e.h16 else "f32"   # (the 16-bit instantiation: its name in a trace is the same in both library builds)
        if use_halo and cp.w_frag == 2:
            kname = f"conv3x3_v3_kernel<{th.value},{hbn.value},{'2,4,1' if hbn.value == 256 else '2,2,2'},{cp.xf_mode}>"  # (the template's name in a trace)
        elif use_halo:
            kname = f"conv3x3_halo_kernel<{dn},{th.value},{tw.value},{hbn.value},{cp.xf_mode},{int(e.lib.gdrn_conv3x3_halo_waves(ref)) // 4}>"
        else:
            kname = f"conv_gemm_kernel<{dn},{bm.value},{bn.value}>"
        # algorithmic bytes (SURVEY 8(d) convention: activations in + out once, weights once, at the storage width)
        esz = 2 if e.h16 else 4
        nbytes = (self.B * Hi * Wi * cp.Cin + self.B * Ho * Wo * cp.Cout) * esz + cp.Cout * cp.Cin * cp.KH * cp.KW * esz
        run.meta = dict(kernel=kname, flops=2.0 * macs, bytes=float(nbytes), layer=L.key + (":dgrad" if w is not None else ""))
        return run, cp

    def _s2(self, L, x, x_cs, y, Hi, Ho, Ld=None, yd=None, stats=None, stats_d=None, bias=None, bias_d=None, act=0, evalw=False):
        """forward launch of a 3x3 stride-2 conv on the parity-plane halo kernel (gdrn_conv3x3s2), with the block's 1x1 shortcut conv Ld in the
        same launch when given.  Returns (op, NS(_stats_rows) of the main conv, the same for the shortcut) or None when the library does not
        cover the shape (maps narrower than 16 pixels: the generic kernel keeps them)."""
        e = self.e
        if not (e.s2_halo and L.wfF is not None and L.s2) or (Ho < 16 and not e.s2_tw8):
            return None
        sp = S2Params()
        sp.x, sp.y = ptr(x), ptr(y)
        sp.w = ptr(L.wfF_e if evalw else L.wfF)
        sp.bias, sp.stats = ptr(bias), ptr(stats)
        if Ld is not None:
            sp.wd, sp.yd, sp.bias_d, sp.stats_d = ptr(Ld.wf_e if evalw else Ld.wf), ptr(yd), ptr(bias_d), ptr(stats_d)
            sp.yd_cs, sp.wd_rows = yd.shape[-1], Ld.rows_f
            assert Ld.cin_f == L.cin_f, (L.key, Ld.key)
        sp.Hi = sp.Wi = Hi
        sp.Ho = sp.Wo = Ho
        sp.Cin, sp.x_cs, sp.Cout, sp.y_cs = L.cin_f, x_cs, L.O, y.shape[-1]
        sp.N, sp.w_rows, sp.act, sp.dtype = self.B, L.rows_f, act, e.dt
        ref = C.byref(sp)
        if not int(e.lib.gdrn_conv3x3s2_ok(ref)):
            return None
        self.keep.append(sp)
        rows = int(e.lib.gdrn_conv3x3s2_stats_rows(ref))
        for st_, c_ in ((stats, L.O), (stats_d, L.O)):
            assert st_ is None or rows * 2 * c_ <= st_.numel(), (L.key, rows)

        def run(st, ctx):
            s_ = e.lib.gdrn_conv3x3s2(ref, st)
            if s_:
                check(s_, f"conv3x3s2 {L.key}")

        macs = self.B * Ho * Ho * L.O * L.I * 9 + (self.B * Ho * Ho * Ld.O * Ld.I if Ld is not None else 0)
        esz = 2
        run.meta = dict(kernel=f"conv3x3s2_kernel<{'true' if Ld is not None else 'false'}>", flops=2.0 * macs,
                        bytes=float((self.B * Hi * Hi * L.cin_f + (2 if Ld is not None else 1) * self.B * Ho * Ho * L.O) * esz + L.O * L.cin_f * 9 * esz),
                        layer=L.key + ("+downsample" if Ld is not None else ""))
        return run, NS(_stats_rows=rows), NS(_stats_rows=rows)

    def _s2d(self, L, dy, dx, Ho, Hi, cin, Ld=None, dyd=None, bnb=None):
        """data-gradient launch of a 3x3 stride-2 conv on gdrn_conv3x3s2_dgrad: dx [B, Hi, Hi, cin] from dy [B, Ho, Ho, L.O]; Ld / dyd: the block's
        1x1 shortcut conv and ITS output gradient, added in the same launch; bnb = (bn key, raw input of that BatchNorm, stored activation): dx is
        the gradient w.r.t. that BatchNorm(+ReLU)'s output -- masked, its backward sums reduced in the epilogue, and the coefficient launch
        appended (as _conv does for its bnb).  None when the library does not cover the shape."""
        e = self.e
        if not (e.s2_halo and getattr(L, "wdF", None) is not None and L.s2) or (Ho < 16 and not e.s2_tw8):
            return None
        sp = S2dParams()
        sp.dy, sp.w, sp.dx = ptr(dy), ptr(L.wdF), ptr(dx)
        if Ld is not None:
            sp.dyd, sp.wdd, sp.dyd_cs, sp.wdd_rows = ptr(dyd), ptr(Ld.wd), dyd.shape[-1], Ld.rows_d
            assert Ld.cin_d == L.cin_d, (L.key, Ld.key)
        if bnb is not None:
            bkey, braw, bmask = bnb
            sb = self.bn[bkey]
            sp.bnb_x, sp.bnb_mask, sp.bnb_mean, sp.bnb_invstd, sp.bnb_rows, sp.bnb_cs = ptr(braw), ptr(bmask), ptr(sb.mean), ptr(sb.invstd), ptr(self.stats), braw.shape[-1]
        sp.Hi = sp.Wi = Hi
        sp.Ho = sp.Wo = Ho
        sp.Cin, sp.dx_cs, sp.Cout, sp.dy_cs = cin, dx.shape[-1], L.cin_d, dy.shape[-1]
        sp.N, sp.w_rows, sp.dtype = self.B, L.rows_d, e.dt
        ref = C.byref(sp)
        if not int(e.lib.gdrn_conv3x3s2_dgrad_ok(ref)):
            return None
        self.keep.append(sp)

        def conv_only(st, ctx):
            s_ = e.lib.gdrn_conv3x3s2_dgrad(ref, st)
            if s_:
                check(s_, f"conv3x3s2_dgrad {L.key}")

        run = conv_only
        if bnb is not None:
            nrows = int(e.lib.gdrn_conv3x3s2_dgrad_rows(ref))
            assert nrows * 2 * cin <= self.stats.numel(), (L.key, nrows)
            coef = self._bn_coef_op(bnb[0], self.stats, nrows)

            def run(st, ctx):
                conv_only(st, ctx)
                coef(st, ctx)

            run.parts = (conv_only, coef)
        macs = self.B * Ho * Ho * L.O * L.I * 9 + (self.B * Ho * Ho * Ld.O * Ld.I if Ld is not None else 0)
        run.meta = dict(kernel=f"conv3x3s2_dgrad_kernel<{'true' if Ld is not None else 'false'},{'true' if bnb is not None else 'false'}>", flops=2.0 * macs,
                        bytes=float((self.B * Hi * Hi * cin + (2 if Ld is not None else 1) * self.B * Ho * Ho * L.cin_d) * 2 + L.cin_d * cin * 9 * 2),
                        layer=L.key + ":dgrad" + ("+downsample:dgrad" if Ld is not None else ""))
        return run

    def _convT_dgrad(self, LT, dy, dx, Hi, Ho, bnb):
        """data gradient of the head's ConvTranspose2d = a 3x3 stride-2 pad-1 conv of the output gradient dy [B, Hi, Hi, LT.O] -> dx [B, Ho = Hi / 2, Ho,
        LT.cin_f] on gdrn_conv3x3s2 with the BatchNorm-backward epilogue (bnb = (bn key, raw input of that BatchNorm, stored activation)) and the
        coefficient launch behind it.  None when the library does not cover the shape."""
        e = self.e
        if not (e.s2_halo and e.s2_tw8 and LT.kind == "convT" and getattr(LT, "wdF", None) is not None):
            return None
        bkey, braw, bmask = bnb
        sb = self.bn[bkey]
        sp = S2Params()
        sp.x, sp.w, sp.y = ptr(dy), ptr(LT.wdF), ptr(dx)
        sp.Hi = sp.Wi = Hi
        sp.Ho = sp.Wo = Ho
        sp.Cin, sp.x_cs, sp.Cout, sp.y_cs = LT.cin_d, dy.shape[-1], LT.I, dx.shape[-1]
        sp.N, sp.w_rows, sp.dtype = self.B, LT.rows_d, e.dt
        sp.bnb_x, sp.bnb_mask, sp.bnb_mean, sp.bnb_invstd, sp.bnb_rows, sp.bnb_cs = ptr(braw), ptr(bmask), ptr(sb.mean), ptr(sb.invstd), ptr(self.stats), braw.shape[-1]
        ref = C.byref(sp)
        if not int(e.lib.gdrn_conv3x3s2_ok(ref)):
            return None
        self.keep.append(sp)
        nrows = int(e.lib.gdrn_conv3x3s2_stats_rows(ref))
        assert nrows * 2 * LT.I <= self.stats.numel(), (LT.key, nrows)
        coef = self._bn_coef_op(bkey, self.stats, nrows)

        def conv_only(st, ctx):
            s_ = e.lib.gdrn_conv3x3s2(ref, st)
            if s_:
                check(s_, f"conv3x3s2 (data gradient) {LT.key}")

        def run(st, ctx):
            conv_only(st, ctx)
            coef(st, ctx)

        run.parts = (conv_only, coef)
        run.meta = dict(kernel="conv3x3s2_kernel<false> (data gradient)", flops=2.0 * self.B * Ho * Ho * LT.O * LT.I * 9,
                        bytes=float((self.B * Hi * Hi * LT.O + 2 * self.B * Ho * Ho * LT.I) * 2 + LT.O * LT.I * 9 * 2), layer=LT.key + ":dgrad")
        return run

    def _convT_fwd(self, LT, bnkey, x, y, Hi, Ho, stats=None, evalmode=False):
        """forward launch of the head's ConvTranspose2d(3, stride 2, pad 1, output_padding 1) on gdrn_conv3x3s2_dgrad (the same sum as a stride-2
        conv's data gradient: x [B, Hi, Hi, LT.cin_f] -> y [B, Ho = 2 Hi, Ho, LT.O]) with a forward epilogue: statistics rows (train mode) or the
        folded BatchNorm's shift + ReLU (evalmode).  Returns (op, NS(_stats_rows)) or None when the library does not cover the shape."""
        e = self.e
        if not (e.s2_halo and LT.kind == "convT" and getattr(LT, "wfF", None) is not None) or (Hi < 16 and not e.s2_tw8):
            return None
        sp = S2dParams()
        sp.dy, sp.dx = ptr(x), ptr(y)
        sp.Hi = sp.Wi = Ho
        sp.Ho = sp.Wo = Hi
        sp.Cin, sp.dx_cs, sp.Cout, sp.dy_cs = LT.O, y.shape[-1], LT.cin_f, x.shape[-1]
        sp.N, sp.w_rows, sp.dtype = self.B, LT.rows_f, e.dt
        sp.w = ptr(LT.wfF)
        ref = C.byref(sp)
        if not int(e.lib.gdrn_conv3x3s2_dgrad_ok(ref)):
            return None
        if evalmode:
            f = e.fold(bnkey, LT)
            if not self._fold_hooked:
                self.eval_prep.append(lambda st, ctx: self.e.eval_refresh())
                self._fold_hooked = True
            sp.w, sp.bias, sp.act = ptr(LT.wfF_e), ptr(f.shift), 1
        else:
            sp.stats = ptr(stats)
        self.keep.append(sp)
        rows = int(e.lib.gdrn_conv3x3s2_dgrad_rows(ref))
        assert stats is None or rows * 2 * LT.O <= stats.numel(), (LT.key, rows)

        def run(st, ctx):
            s_ = e.lib.gdrn_conv3x3s2_dgrad(ref, st)
            if s_:
                check(s_, f"conv3x3s2_dgrad (forward) {LT.key}")

        run.meta = dict(kernel="conv3x3s2_dgrad_kernel<false,false> (forward)", flops=2.0 * self.B * Hi * Hi * LT.O * LT.I * 9,
                        bytes=float((self.B * Ho * Ho * LT.O + self.B * Hi * Hi * LT.cin_f) * 2 + LT.O * LT.cin_f * 9 * 2), layer=LT.key)
        return run, NS(_stats_rows=rows)

    def _stats_rows(self, cp):
        return cp._stats_rows

    def _wgrad(self, L, x, dy, Hi, Wi, Ho, Wo, stride, pad, cin, cout, x_cs, dy_cs, KH=None, KW=None, defer=False):
        e = self.e
        wp = WgradParams()
        wp.x, wp.dy, wp.dw = ptr(x), ptr(dy), ptr(L.dwp)
        wp.Hi, wp.Wi, wp.Cin, wp.x_cs = Hi, Wi, cin, x_cs
        wp.Ho, wp.Wo, wp.Cout, wp.dy_cs = Ho, Wo, cout, dy_cs
        wp.KH, wp.KW, wp.stride, wp.pad = KH or L.KH, KW or L.KW, stride, pad
        wp.M, wp.dtype, wp.splits, wp.variant = self.B * Ho * Wo, e.dt, 0, 0
        self.keep.append(wp)
        ref = C.byref(wp)
        # halo-tiled kernel: every 3x3 pad-1 conv, stride 1 or 2 (incl. Patch-PnP's and, roles swapped, the head's ConvTranspose)
        use_halo = e.use_halo and L.kind in ("conv", "convT") and L.KK == 9 and bool(e.lib.gdrn_conv3x3_wgrad_ok(ref))
        fn = e.lib.gdrn_conv3x3_wgrad if use_halo else e.lib.gdrn_conv_wgrad
        if use_halo:
            # deferred: one grouped launch per gradient bucket (see _finish_unpack) -- weight gradients are off the
            # critical path, and a grid over many layers fills the chip with far fewer pixel-range splits per layer
            # (flops, algorithmic bytes: X and dY read once at the storage width, the fp32 gradient written once)
            self._wgrad_deferred.append((len(self.bwd_groups), L, wp, (2.0 * self.B * Ho * Wo * L.O * L.I * L.KK,
                                                                      2.0 * self.B * (Hi * Wi * cin + Ho * Wo * cout) + 4.0 * cout * cin * 9)))
            return None

        self._zero_regions.append(self._pad16(L.dwp, e.dwp_flat))  # accumulated with fp32 atomics

        def run(st, ctx):
            s = fn(ref, st)
            if s:
                check(s, f"conv_wgrad {L.key}")

        if L.kind == "stem":
            macs = self.B * 128 * 128 * 64 * 3 * 49
        elif L.kind == "convT":
            macs = self.B * 64 * L.O * L.I * L.KK
        else:
            macs = self.B * Ho * Wo * L.O * L.I * L.KK
        bco, bci = (64 if cout <= 64 else 128), (128 if cin % 128 == 0 else 64)
        kname = "conv3x3_wgrad_kernel" if use_halo else f"conv_wgrad_kernel<{'bf16' if e.h16 else 'f32'},{bco},{bci}>"
        esz = 2 if e.h16 else 4
        run.meta = dict(kernel=kname, flops=2.0 * macs, bytes=float(self.B * (Hi * Wi * cin + Ho * Wo * cout) * esz + 4 * cout * cin * wp.KH * wp.KW),
                        layer=L.key + ":wgrad")
        run.side = e.side_small  # feeds only the optimizer: off the data-gradient chain (side stream, see run_backward)
        run.defer = bool(run.side and defer)
        return run

    def _side(self, op, defer=False):
        """mark a backward op whose result only the optimizer reads: it may run on the side stream (Engine.side_small).
        defer: the op may wait for the end of its gradient bucket (see _merge_forks)."""
        op.side = self.e.side_small
        op.defer = bool(defer and op.side)
        return op

    def _merge_forks(self):
        """Every switch of run_backward from the main stream to the side stream is an event record on the main stream, and the chain's next
        kernel starts ~7 us late behind it (r6 timeline: 11 such gaps per step).  Side ops marked `defer` -- the small weight / bias gradients of
        the fc layers and of the 1x1 shortcut convs of layer3.0 / layer4.0, whose buffers are their own -- move to the end of their bucket, in
        front of its grouped weight-gradient launch: one fork per bucket instead of one per op."""
        e = self.e
        if not (e.wgrad_stream and e.merge_forks):
            return
        first = list(e.bucket_first_group)
        for bkt, g0 in enumerate(first):
            hi = first[bkt - 1] if bkt else len(self.bwd_groups)
            moved = []
            for gi in range(hi - 1, g0 - 1, -1):   # execution order: last forward group first
                g = self.bwd_groups[gi]
                moved += [op for op in g if getattr(op, "defer", False)]
                self.bwd_groups[gi] = [op for op in g if not getattr(op, "defer", False)]
            if not moved:
                continue
            g = self.bwd_groups[g0]
            at = next((i for i, op in enumerate(g) if getattr(op, "bucket_end", False)), len(g))
            if bkt == len(first) - 1 and getattr(self, "_front_n", 0):
                at = 0   # (last bucket, tail overlap: its bucket-end ops sit in FRONT of the stem group)
            g[at:at] = moved

    @staticmethod
    def _pad16(t, flat):
        """the 16-byte-padded slice of the flat fp32 buffer `flat` that holds the view `t` (offsets in these buffers are multiples of 4 floats)."""
        off = (t.data_ptr() - flat.data_ptr()) // 4
        assert 0 <= off and off % 4 == 0 and off + t.numel() <= flat.numel(), off
        return flat[off: min(off + _ru(t.numel(), 4), flat.numel())]

    def _grad16(self, g):
        return self._pad16(g, self.e.grad_flat)

    def _build_zero_table(self):
        from .cabi import ZeroTask, to_device_table

        e = self.e
        chunk = e.lib.gdrn_zero_chunk()
        tasks, starts = [], [0]
        seen = set()
        for t in self._zero_regions:
            if t.data_ptr() in seen:
                continue
            seen.add(t.data_ptr())
            assert t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 and (t.numel() * 4) % 16 == 0, (t.shape, t.data_ptr() % 16)
            n16 = t.numel() * 4 // 16
            tasks.append(ZeroTask(p=t.data_ptr(), n16=n16))
            starts.append(starts[-1] + (n16 + chunk - 1) // chunk)
        self._zero_tab = (to_device_table(tasks, e.dev), torch.tensor(starts, dtype=torch.int32, device=e.dev), len(tasks), starts[-1])

    def _unpack(self, L):
        """packed fp32 weight gradient -> the parameter's .grad layout.  Every layer except the stem is deferred to one
        multi-tensor launch per gradient bucket (see _finish_unpack); the returned op is then a no-op marker."""
        e = self.e
        lib = e.lib
        O, I, KK = L.O, L.I, L.KK
        for n_ in L.src:
            self.grad_group[n_] = len(self.bwd_groups)
        if L.kind == "stem":
            g = e.grads[L.src[0]]
            # (side like the weight-gradient launch whose result it reads: the two stay in order on one stream)
            return self._side(lambda st, ctx: check(lib.gdrn_unpack_stem_w(ptr(L.dwp), ptr(g), st), "unpack_stem_w"))
        self._unpack_pending.append((len(self.bwd_groups), L))
        return None  # marker, dropped when the groups are flattened

    def _bn_fwd(self, bnkey, raw, cp, C_, npix, y, residual=None, relu=1, stats=None):
        """finalize (train) or eval params, then apply.  Returns list of fwd ops.  stats: the producer's partial-row scratch (default self.stats)."""
        e, lib = self.e, self.e.lib
        stats = self.stats if stats is None else stats
        s = NS(mean=e._empty(C_, dtype=torch.float32), invstd=e._empty(C_, dtype=torch.float32),
               scale=e._empty(C_, dtype=torch.float32), shift=e._empty(C_, dtype=torch.float32), C=C_, npix=npix)
        self.bn[bnkey] = s
        g, b = e.P[bnkey + ".weight"], e.P[bnkey + ".bias"]
        rm, rv, nbt = e.Bf[bnkey + ".running_mean"], e.Bf[bnkey + ".running_var"], e.Bf[bnkey + ".num_batches_tracked"]
        ops = []
        if self.bn_train:
            rows = self._stats_rows(cp)
            assert rows * 2 * C_ <= stats.numel(), (bnkey, rows, C_)  # the producer's partial rows fit the scratch
            ops.append(lambda st, ctx: check(lib.gdrn_bn_finalize(ptr(stats), rows, C_, float(npix), ptr(g), ptr(b), ptr(rm),
                                                                  ptr(rv), ptr(nbt), 0.1, 1e-5, ptr(s.mean), ptr(s.invstd),
                                                                  ptr(s.scale), ptr(s.shift), ptr(e.bn_ws), st), "bn_finalize"))
        else:
            # eval mode: scale/shift only depend on the parameters and running statistics -- recomputed when those changed
            # (see run_forward), not on every call (43 tiny launches = 8 % of an inference forward)
            self.eval_prep.append(lambda st, ctx: check(lib.gdrn_bn_eval_params(ptr(g), ptr(b), ptr(rm), ptr(rv), 1e-5, C_, ptr(s.scale),
                                                                                ptr(s.shift), st), "bn_eval_params"))
        if y is not None:
            ops.append(lambda st, ctx: check(lib.gdrn_bn_apply(ptr(raw), ptr(s.scale), ptr(s.shift), ptr(residual), ptr(y), npix,
                                                               C_, relu, e.dt, st), "bn_apply"))
        return ops

    def _xf_ok(self, L, mode=None, hw=0):
        """can layer L's halo launches (forward and data gradient) take a fused operand transform (of that mode, on hw x hw maps)?"""
        e = self.e
        if mode is not None and (not (e.xf_mask >> (mode - 1)) & 1 or hw > e.xf_maxhw or L.I < e.xf_minc):
            return False
        return e.fuse_xf and self.bn_train and L.kind == "conv" and L.wfF is not None and L.KK == 9 and not L.s2

    def _bn_coef_op(self, bnkey, rows, nrows):
        """rows [nrows][2][C] of BatchNorm-backward sums -> (ka, kb, kc) of dx = ka*g + kb*x + kc, dgamma, dbeta (one launch)."""
        e, lib = self.e, self.e.lib
        s = self.bn[bnkey]
        if getattr(s, "ka", None) is None:
            s.ka, s.kb, s.kc = (e._empty(s.C, dtype=torch.float32) for _ in range(3))
        g = e.P[bnkey + ".weight"]
        dg, db = e.grads[bnkey + ".weight"], e.grads[bnkey + ".bias"]
        self.grad_group[bnkey + ".weight"] = self.grad_group[bnkey + ".bias"] = len(self.bwd_groups)
        return lambda st, ctx: check(lib.gdrn_bn_bwd_coef(ptr(rows), nrows, s.C, s.npix, ptr(g), ptr(s.mean), ptr(s.invstd), ptr(s.ka), ptr(s.kb),
                                                          ptr(s.kc), ptr(dg), ptr(db), st), "bn_bwd_coef")

    def _conv_bn_eval(self, L, bnkey, x, xs, y, Hi, Wi, Ho, Wo, stride, pad, relu, residual=None, add_cs=0, **kw):
        """eval mode: conv -> BatchNorm (-> +residual) (-> ReLU) as ONE launch, y = act(conv_{w*scale}(x) + shift + residual)."""
        f = self.e.fold(bnkey, L)
        if not self._fold_hooked:
            self.eval_prep.append(lambda st, ctx: self.e.eval_refresh())
            self._fold_hooked = True
        op, _ = self._conv(L, x, xs, y, Hi, Wi, Ho, Wo, stride, pad, bias=f.shift, addend=residual, add_cs=add_cs, act=1 if relu else 0,
                           evalw=True, **kw)
        return op

    def _fusable(self, L):
        """can the data gradient of layer L run on the halo kernel (and so carry a fused BN-backward reduction)?"""
        e = self.e
        return e.use_halo and e.h16 and L.kind == "conv" and L.wfF is not None

    def _bn_bwd(self, bnkey, dy, ymask, raw, dx, g_out=None, affine_mask=False, prereduced=False, xf=False, apply=True, sums_done=False):
        """BatchNorm(+ReLU) backward: [reduce -> coef ->] apply.  prereduced: dy arrives masked and the producing data-gradient
        launch has already turned its epilogue rows into the coefficients (see _conv, bnb).
        affine_mask: BN -> ReLU without residual: the ReLU mask is recomputed from raw*scale+shift (what bn_apply
        evaluated) instead of reading the stored activation `ymask` -- one tensor pass less in both kernels.
        xf: the apply pass is left to the consumer halo conv (returns (ops, xf dict for _conv); dx is written by that conv).
        apply=False: only reduce + coef (the stem's weight-gradient kernel applies the coefficients itself)."""
        e, lib = self.e, self.e.lib
        s = self.bn[bnkey]
        msc, msh = (s.scale, s.shift) if affine_mask else (None, None)
        ym = None if affine_mask else ymask
        ops = []
        if prereduced:
            self._bn_coef_op(bnkey, self.stats, 1)  # (allocates the coefficient vectors; the launch itself sits in _conv)
            msc = msh = ym = None                    # dy is already masked
        elif sums_done:
            # the launch that produced dy (gdrn_upsample2x_bwd_bnsums) has written the reduction's rows already; dy itself is NOT masked
            nrows = int(lib.gdrn_bn_bwd_reduce_rows(s.npix, s.C, e.dt))
            assert nrows > 0 and nrows * 2 * s.C <= self.stats.numel(), (bnkey, nrows)
            ops.append(self._bn_coef_op(bnkey, self.stats, nrows))
        else:
            nrows = int(lib.gdrn_bn_bwd_reduce_rows(s.npix, s.C, e.dt))
            assert nrows > 0 and nrows * 2 * s.C <= self.stats.numel(), (bnkey, nrows)
            coef = self._bn_coef_op(bnkey, self.stats, nrows)
            ops += [lambda st, ctx: check(lib.gdrn_bn_bwd_reduce(ptr(dy), ptr(ym), ptr(raw), ptr(s.mean), ptr(s.invstd), ptr(msc), ptr(msh), s.npix,
                                                                s.C, ptr(self.stats), e.dt, st), "bn_bwd_reduce"), coef]
        if xf:
            assert g_out is None and ym is None
            mode = 4 if msc is not None else 3
            return ops, dict(mode=mode, x2=raw, a=s.ka, b=s.kb, c=s.kc, msc=msc, msh=msh, out=dx, relu=False)
        if apply:
            ops.append(lambda st, ctx: check(lib.gdrn_bn_bwd_apply(ptr(dy), ptr(ym), ptr(raw), ptr(s.ka), ptr(s.kb), ptr(s.kc), ptr(msc), ptr(msh), s.npix,
                                                                   s.C, ptr(dx), ptr(g_out), e.dt, st), "bn_bwd_apply"))
        return ops

    def _unpack_task(self, L, packed_ptr, grad, rows_valid, rows_off=0):
        from .cabi import PackTask

        O, I, KK = L.O, L.I, L.KK
        if L.kind == "convT":  # dwp [ci=I][KK][co(out_ch)] -> weight[ci][co][k]
            a = (I, 1, KK, L.out_ch, I, 1, O, O * KK, 0, 1, KK)
        else:
            a = (rows_valid, 1, KK, L.in_ch, rows_valid, 1, I, I * KK, 0, 1, KK)
        A1, A2, T, B, A1v, A2v, Bv, s1, s2, st, sb = a
        return PackTask(src=packed_ptr, dst=grad.data_ptr(), A1=A1, A2=A2, T=T, B=B, A1v=A1v, A2v=A2v, Bv=Bv, flip=0, s1=s1, s2=s2,
                        st=st, sb=sb, n=A1v * A2v * T * Bv, frag=0, pad_=0)

    def _bucket_end(self, bkt, op):
        """place a bucket-end op (grouped weight gradient, its reduction, gradient unpack).  Behind the bucket's last group -- except for the
        LAST bucket with the side stream on: its ops only need the layers' data gradients, which exist before the stem's backward (max-pool
        backward, BatchNorm coefficients, stem weight gradient: ~0.2 ms on the main stream), so they go in FRONT of the stem group and run
        under it instead of behind it."""
        e = self.e
        op.bucket_end = True
        g = self.bwd_groups[e.bucket_first_group[bkt]]
        if e.wgrad_stream and e.tail_overlap and bkt == len(e.bucket_first_group) - 1 and e.bucket_first_group[bkt] == 0:
            n = getattr(self, "_front_n", 0)
            g.insert(n, op)
            self._front_n = n + 1
        else:
            g.append(op)

    def _finish_unpack(self):
        """One gdrn_unpack_multi per gradient bucket (pnp | head | layer4 | layer3 | rest), appended to the backward group that
        completes the bucket, so the RCCL exchange of a bucket still starts as soon as its gradients exist."""
        from .cabi import to_device_table

        e, lib = self.e, self.e.lib
        chunk = lib.gdrn_pack_chunk()
        ng = len(self.bwd_groups)
        assert ng == 29, ng
        # backward groups in forward order: stem(0) layer1(1-3) layer2(4-7) layer3(8-13) layer4(14-16) convT(17) head convs(18-23)
        # head out(24) pnp convs(25-27) fc(28)
        first_group = e.bucket_first_group  # forward index of the LAST-executed group of each bucket
        from .cabi import WreduceTask

        bucket_of = lambda gi: next(i for i, g0 in enumerate(first_group) if gi >= g0)
        # ---- grouped halo weight gradients: a common number of 8x8 pixel patches per workgroup within a bucket, chosen so
        # that the bucket's grid has ~wgrad_blocks workgroups (2 per CU resident); longest-running tasks first
        # launch groups = the buckets
        cuts = sorted(set(first_group) | set(e.wgrad_cuts), reverse=True)
        cut_of = lambda gi: next(c for c in cuts if gi >= c)
        wg_bucket = {c: [] for c in cuts}
        for gi, L, wp, fb in self._wgrad_deferred:
            wg_bucket[cut_of(gi)].append((L, wp, fb))
        self._wgrad_tables = []
        for cut, items_all in wg_bucket.items():
            if not items_all:
                continue
            bkt = bucket_of(cut)
            last_bucket = bkt == len(first_group) - 1
            items, kind, cot, nblocks_target = items_all, 0, 64, e.wgrad_blocks   # one grouped launch of the 64 x 64 tile per bucket
            # work of a layer in 32-pixel k-steps: an 8x8-pixel stage (stride 1) is two, a 4x8-pixel stage (stride 2) one
            geo = []
            for L, wp, flops in items:
                s2 = wp.stride == 2
                npatch = (wp.M // (wp.Ho * wp.Wo)) * (wp.Ho // (4 if s2 else 8)) * (wp.Wo // 8)
                geo.append((npatch, npatch * (1 if s2 else 2), (wp.Cout // cot) * (wp.Cin // 64)))
            # k-steps per workgroup: the grid's target, but no more than the layers of a shape class that carries >= 20 % of the bucket's work
            # have -- a layer with fewer k-steps than `per` runs as ONE split of short workgroups beside the others' long ones (r6: layer4's 128
            # against 171-256 for layer3 in their common launch: 434 us stand-alone against 322 with 128 everywhere, -0.06 ms per step;
            # profiles/r06_wgrad_bucket_balance.txt).  Small classes (the head's ConvTranspose: 1 % of its bucket) do not set the length.
            total = sum(u * t for _, u, t in geo)
            share = {}
            for _, u, t in geo:
                share[u] = share.get(u, 0) + u * t
            # (floor of 64 k-steps: only the Patch-PnP bucket sits on it -- 168 instead of 672 partial tiles for its reduction to read; r6: -0.02 ms)
            per = max(64 if self.B >= 32 else 16, min([total // nblocks_target] + [u for u, w in share.items() if 5 * w >= total]))
            tasks = []
            for (L, wp, flops), (npatch, units, tiles) in zip(items, geo):
                wp.variant = kind
                wp.ws = ptr(e.dwp_flat)  # non-null placeholder for the split query
                wp.splits = max(1, units // per)
                wp.splits = int(lib.gdrn_conv3x3_wgrad_splits(C.byref(wp)))  # normalised: no empty split
                assert wp.splits >= 1, (L.key, kind)
                ws = e._empty(wp.splits * wp.Cout * wp.Cin * 9, dtype=torch.float32)
                self.keep.append(ws)
                wp.ws, wp.dw = ptr(ws), None
                # the kernel's "Cin" role = the parameter's input channels for a conv (69 real of 128 for Patch-PnP's first conv), its
                # second dimension for the ConvTranspose (weight [Cin_w][Cout_w][3][3] with x = output gradient, dy = input)
                self._wreduce[L.key] = (ws, wp.splits, wp.Cout, wp.Cin, L.O if L.kind == "convT" else L.I)
                tasks.append((-(units // wp.splits), len(tasks), wp, tiles, flops))
            tasks.sort(key=lambda t: t[:2])
            starts = [0]
            for _, _, wp, tiles, _ in tasks:
                starts.append(starts[-1] + tiles * wp.splits)
            tab = to_device_table([t[2] for t in tasks], e.dev)
            stt = torch.tensor(starts, dtype=torch.int32, device=e.dev)
            self._wgrad_tables.append((tab, stt))
            nt, nb = len(tasks), starts[-1]

            def run(st, ctx, tab=tab, stt=stt, nt=nt, nb=nb, lds=(0 if last_bucket else e.wgrad_side_lds)):
                # on the side stream (wgrad_stream) a bucket's weight gradients run under the NEXT bucket's data-gradient chain with one
                # workgroup per CU (LDS request), so that the chain's workgroups find room on every CU; the last bucket has nothing to hide under
                check(lib.gdrn_conv3x3_wgrad_multi_lds(ptr(tab), ptr(stt), nt, nb, lds if (e.wgrad_stream or e.wgrad_force_lds) else 0, st), "conv3x3_wgrad_multi")

            run.meta = dict(kernel="conv3x3_wgrad_multi_kernel", flops=sum(t[4][0] for t in tasks), bytes=sum(t[4][1] for t in tasks),
                            layer=f"bucket{bkt}@{cut}:wgrad x{nt} ({nb} wg)")
            run.side = True
            if cut in first_group:
                self._bucket_end(bkt, run)
            else:
                self.bwd_groups[cut].append(run)   # behind the data gradients of group `cut`, the last-executed one of this launch

        per_bucket = {i: [] for i in range(len(first_group))}
        red_bucket = {i: [] for i in range(len(first_group))}
        for gi, L in self._unpack_pending:
            bkt = bucket_of(gi)
            if L.key in self._wreduce:
                ws, nsplit, cout, cin, civ = self._wreduce[L.key]
                assert cout == (L.I if L.kind == "convT" else L.O) and civ <= cin, (L.key, cout, cin, civ)
                assert e.grads[L.src[0]].numel() == cout * civ * 9, L.key
                red_bucket[bkt].append(WreduceTask(ws=ws.data_ptr(), dst=e.grads[L.src[0]].data_ptr(), nsplit=nsplit, Cout=cout, Cin=cin,
                                                   cin_valid=civ, s_co=civ * 9, s_ci=9, s_t=1))
            elif L.key == "pnp_net.fc_rt":
                per_bucket[bkt].append(self._unpack_task(L, L.dwp.data_ptr(), e.grads["pnp_net.fc_r.weight"], 6))
                per_bucket[bkt].append(self._unpack_task(L, L.dwp.data_ptr() + 6 * L.in_ch * 4, e.grads["pnp_net.fc_t.weight"], 3))
            else:
                per_bucket[bkt].append(self._unpack_task(L, L.dwp.data_ptr(), e.grads[L.src[0]], L.O))
        self._unpack_tables = []
        for bkt, tasks in per_bucket.items():
            if not tasks:
                continue
            starts = [0]
            for t in tasks:
                starts.append(starts[-1] + (cabi.transpose_blocks(lib, t) or (t.n + chunk - 1) // chunk))   # (fc1's gradient: 64 x 64 tiles through LDS)
            tab = to_device_table(tasks, e.dev)
            stt = torch.tensor(starts, dtype=torch.int32, device=e.dev)
            self._unpack_tables.append((tab, stt))
            nt, nb = len(tasks), starts[-1]
            def unpack(st, ctx, tab=tab, stt=stt, nt=nt, nb=nb):
                check(lib.gdrn_unpack_multi(ptr(tab), ptr(stt), nt, nb, st), "unpack_multi")

            unpack.side = True
            self._bucket_end(bkt, unpack)
        for bkt, tasks in red_bucket.items():
            if not tasks:
                continue
            starts = [0]
            for t in tasks:
                starts.append(starts[-1] + t.Cout * t.Cin // 256)
            tab = to_device_table(tasks, e.dev)
            stt = torch.tensor(starts, dtype=torch.int32, device=e.dev)
            self._unpack_tables.append((tab, stt))
            nt, nb = len(tasks), starts[-1]
            def wreduce(st, ctx, tab=tab, stt=stt, nt=nt, nb=nb):
                check(lib.gdrn_wgrad_reduce_multi(ptr(tab), ptr(stt), nt, nb, st), "wgrad_reduce_multi")

            wreduce.side = True
            self._bucket_end(bkt, wreduce)

    # ---- graph -------------------------------------------------------------------------------
    def _build(self):
        e, lib, B = self.e, self.e.lib, self.B
        S, T, WL = self.bn_train, self.has_backward, self.with_loss  # batch stats | backward graph | losses
        FOLD = (not S) and e.fold_bn  # eval mode: BatchNorm folded into the preceding conv (weights pre-scaled, shift as bias)

        def E(*shape, dtype=None):
            # every plan buffer is pinned in self.keep: the pre-bound C structs hold RAW device pointers, so a
            # buffer referenced only through them would otherwise be returned to the caching allocator when
            # _build() returns and be handed out again while the kernels still write to it
            t = e._empty(*shape, dtype=dtype)
            self.keep.append(t)
            return t

        F32t = torch.float32
        # per-tile BN partial sums, max over layers: conv tiles need <= B*32768 floats (64-pixel tiles of the 64x64 maps at 256
        # channels); the direct stem kernel writes one [2][64] row per wave
        n_stats = max(B * 32768 + 65536, 1024 * 2 * 512)  # ... and gdrn_bn_bwd_reduce writes <= 1024 rows of 2*C floats
        if e.stem_direct:
            n_stats = max(n_stats, int(lib.gdrn_stem_stats_rows(B)) * 128)
        self.stats = E(n_stats, dtype=F32t)
        self.stats_b = E(B * 4096 + 4096, dtype=F32t)   # rows of a fused 1x1 shortcut conv (gdrn_conv3x3s2: <= B * 16 tiles x 2 x 128 channels)
        nreg = e.nreg

        # ---------------- stem
        Ls = e.layers["backbone.conv1"]
        self.img_p = e._zeros(B, 262, 272, 4)
        fused_stem = e.stem_direct and not S   # eval mode: conv + bn1 + ReLU + max-pool in one kernel (gdrn_stem_conv_pool): no raw0 / idx0
        raw0 = None if fused_stem else E(B, 128, 128, 64)
        p0 = E(B, 64, 64, 64)
        idx0 = None if fused_stem else E(B, 64, 64, 64, dtype=torch.uint8)
        self.fwd.append(lambda st, ctx: check(lib.gdrn_pack_image(ctx["img"], ptr(self.img_p), B, 256, 256, 262, 272, e.dt, st), "pack_image"))
        if fused_stem:
            cp = None
        elif e.stem_direct:
            cp = NS(_stats_rows=int(lib.gdrn_stem_stats_rows(B)))

            def stem(st, ctx):
                check(lib.gdrn_stem_conv(ptr(self.img_p), ptr(e.stem_w32), ptr(raw0), ptr(self.stats) if S else None, B, e.dt, st), "stem_conv")

            stem.meta = dict(kernel="stem_conv_kernel", flops=2.0 * B * 128 * 128 * 64 * 147, layer="backbone.conv1")
            self.fwd.append(stem)
        else:
            op, cp = self._conv(Ls, self.img_p, 4, raw0, 262, 272, 128, 128, 2, 0, cin=64, cout=64, x_cs=4, KH=7, KW=1,
                                stats=self.stats if S else None)
            self.fwd.append(op)
        self.fwd += self._bn_fwd("backbone.bn1", raw0, cp, 64, B * 128 * 128, None)
        s0 = self.bn["backbone.bn1"]
        if fused_stem:
            def stem_pool(st, ctx):
                check(lib.gdrn_stem_conv_pool(ptr(self.img_p), ptr(e.stem_w32), ptr(s0.scale), ptr(s0.shift), ptr(p0), B, e.dt, st), "stem_conv_pool")

            stem_pool.meta = dict(kernel="stem_conv_pool_kernel", flops=2.0 * B * 128 * 128 * 64 * 147, layer="backbone.conv1+bn1+relu+maxpool")
            self.fwd.append(stem_pool)
        else:
            self.fwd.append(lambda st, ctx: check(lib.gdrn_bn_relu_maxpool_fwd(ptr(raw0), ptr(s0.scale), ptr(s0.shift), ptr(p0), ptr(idx0),
                                                                               B, 128, 128, 64, e.dt, st), "bn_relu_maxpool"))
        self.tensors.update({"stem.pool": p0} if fused_stem else {"stem.raw": raw0, "stem.pool": p0})
        if T:
            d_p0 = E(B, 64, 64, 64)
            g_stem = E(B, 128, 128, 64)
            d_raw0 = g_stem  # in place
            self.tensors.update({"stem.d_pool": d_p0, "stem.g": g_stem})
            # with the fused stem weight gradient the max-pool backward also emits the BatchNorm-backward sums of the gradient it
            # writes (one partial row per workgroup): no separate reduce pass over the two 134 MB tensors
            mp_rows = int(lib.gdrn_maxpool_bwd_rows(B, 128, 128, 64, e.dt)) if e.stem_wgrad else 0
            assert mp_rows * 2 * 64 <= self.stats.numel()
            grp = [lambda st, ctx: check(lib.gdrn_maxpool_bwd(ptr(d_p0), ptr(idx0), ptr(raw0), ptr(s0.scale), ptr(s0.shift),
                                                              ptr(g_stem), B, 128, 128, 64, ptr(s0.mean) if mp_rows else None,
                                                              ptr(s0.invstd) if mp_rows else None, ptr(self.stats) if mp_rows else None, e.dt, st),
                                         "maxpool_bwd")]
            if e.stem_wgrad:
                # the stem has no data gradient: its BatchNorm-backward apply is evaluated inside the weight-gradient kernel while
                # the dy tile is staged (no 134 MB d_raw0 round trip, dy read once instead of once per kernel row)
                sb = self.bn["backbone.bn1"]
                gam, dgam, dbet = e.P["backbone.bn1.weight"], e.grads["backbone.bn1.weight"], e.grads["backbone.bn1.bias"]
                gw = e.grads["backbone.conv1.weight"]
                assert gw.is_contiguous() and gw.dtype == torch.float32
                sw_ws = e._empty(int(lib.gdrn_stem_wgrad_parts(B)) * 64 * 224, dtype=torch.float32)
                grp.append(self._bn_coef_op("backbone.bn1", self.stats, mp_rows))  # rows -> (a, b, c), dgamma, dbeta; the apply is fused below

                def stem_wgrad(st, ctx, a=(self.img_p, g_stem, raw0, sb.ka, sb.kb, sb.kc, sw_ws, gw)):
                    # tensors bound as a default argument: _build() reuses short local names further down (late-binding closures)
                    check(lib.gdrn_stem_wgrad(ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), ptr(a[4]), ptr(a[5]), B, ptr(a[6]), ptr(a[7]), e.dt | PREZEROED, st),
                          "stem_wgrad")

                self._zero_regions.append(self._grad16(gw))

                self.grad_group["backbone.conv1.weight"] = len(self.bwd_groups)
                stem_wgrad.meta = dict(kernel="stem_wgrad_kernel", flops=2.0 * B * 128 * 128 * 64 * 147, layer="backbone.conv1:wgrad")
                grp.append(stem_wgrad)
            else:
                grp += self._bn_bwd("backbone.bn1", g_stem, None, raw0, d_raw0)
                grp.append(self._wgrad(Ls, self.img_p, d_raw0, 262, 272, 128, 128, 2, 0, 64, 64, 4, 64, KH=7, KW=1))
                grp.append(self._unpack(Ls))
            self.bwd_groups.append(grp)
        else:
            d_p0 = None

        # ---------------- residual blocks
        # `pend`: the block input `x` is not materialised yet -- it is v = relu(a*x1 + b*x2 + c) of the previous block's bn2 output
        # and identity, evaluated (and written to `x`) by THIS block's conv1 while it stages its patch (xf, see _conv)
        x, d_x, Hc, inpl = p0, d_p0, 64, 64
        pend = None
        for li, (nb, pl) in enumerate(zip(RESNET34_LAYERS, RESNET34_PLANES), start=1):
            for b in range(nb):
                pfx = f"backbone.layer{li}.{b}"
                stride = 2 if (b == 0 and li > 1) else 1
                Ho = Hc // stride
                L1, L2 = e.layers[pfx + ".conv1"], e.layers[pfx + ".conv2"]
                Ld = e.layers.get(pfx + ".downsample.0")
                npo = B * Ho * Ho
                raw1, a1, raw2, out = E(B, Ho, Ho, pl), E(B, Ho, Ho, pl), E(B, Ho, Ho, pl), E(B, Ho, Ho, pl)
                self.tensors.update({pfx + ".raw1": raw1, pfx + ".a1": a1, pfx + ".raw2": raw2, pfx + ".out": out})
                if FOLD and e.block64 and Ld is None and inpl == 64 and pl == 64 and e.use_halo and B >= e.halo_min_b and L1.wfF is not None \
                        and L1.wfmt.get("e", 1) == 1 and L2.wfmt.get("e", 1) == 1 and int(lib.gdrn_block64_eval_ok(B, Hc, Hc, e.dt)):
                    # eval, 64 channels (layer1): the whole BasicBlock as ONE launch, the intermediate in LDS (r6, csrc/block64.hip)
                    f1, f2 = e.fold(pfx + ".bn1", L1), e.fold(pfx + ".bn2", L2)
                    L1.wfmt["e"] = L2.wfmt["e"] = 1   # (fragment-major layout of the first halo kernel: what gdrn_block64_eval reads)
                    if not self._fold_hooked:
                        self.eval_prep.append(lambda st, ctx: self.e.eval_refresh())
                        self._fold_hooked = True

                    def blk(st, ctx, x=x, out=out, L1=L1, L2=L2, f1=f1, f2=f2, Hc=Hc):
                        check(lib.gdrn_block64_eval(ptr(x), ptr(L1.wfF_e), ptr(f1.shift), ptr(L2.wfF_e), ptr(f2.shift), ptr(out), B, Hc, Hc, e.dt, st), "block64_eval")

                    blk.meta = dict(kernel="block64_eval_kernel", flops=2.0 * 2 * B * Hc * Hc * 64 * 64 * 9, layer=pfx + ".conv1+conv2")
                    self.fwd.append(blk)
                    del self.tensors[pfx + ".raw1"], self.tensors[pfx + ".a1"], self.tensors[pfx + ".raw2"]
                    x, Hc, inpl, prev_raw2 = out, Ho, pl, raw2
                    continue
                if FOLD:  # eval: three (four) launches per block, no BatchNorm passes
                    s2op = None
                    if Ld is not None and e.s2_halo and L1.wfF is not None:
                        # stage entry: conv1 (3x3 stride 2) and the 1x1 shortcut in ONE launch of the parity-plane kernel (r6), BatchNorms folded
                        f1, fd = e.fold(pfx + ".bn1", L1), e.fold(pfx + ".downsample.1", Ld)
                        if not self._fold_hooked:
                            self.eval_prep.append(lambda st, ctx: self.e.eval_refresh())
                            self._fold_hooked = True
                        res = E(B, Ho, Ho, pl)
                        s2op = self._s2(L1, x, inpl, a1, Hc, Ho, Ld=Ld, yd=res, bias=f1.shift, bias_d=fd.shift, act=1, evalw=True)
                    if s2op is not None:
                        self.fwd.append(s2op[0])
                    else:
                        self.fwd.append(self._conv_bn_eval(L1, pfx + ".bn1", x, inpl, a1, Hc, Hc, Ho, Ho, stride, 1, relu=True))
                    res = x if Ld is None else (res if s2op is not None else None)
                    if Ld is not None and s2op is None:
                        res = E(B, Ho, Ho, pl)
                        self.fwd.append(self._conv_bn_eval(Ld, pfx + ".downsample.1", x, inpl, res, Hc, Hc, Ho, Ho, stride, 0, relu=False))
                    self.fwd.append(self._conv_bn_eval(L2, pfx + ".bn2", a1, pl, out, Ho, Ho, Ho, Ho, 1, 1, relu=True, residual=res, add_cs=pl))
                    x, Hc, inpl, prev_raw2 = out, Ho, pl, raw2
                    continue
                assert pend is None or self._xf_ok(L1)
                s2op = None
                if Ld is not None and pend is None:
                    # stage entry (r6): conv1 (3x3 stride 2) and the 1x1 shortcut conv in ONE launch of the parity-plane kernel, each with its
                    # own BatchNorm-statistics rows
                    rawd_ = E(B, Ho, Ho, pl)
                    s2op = self._s2(L1, x, inpl, raw1, Hc, Ho, Ld=Ld, yd=rawd_, stats=self.stats if S else None, stats_d=self.stats_b if S else None)
                if s2op is not None:
                    op, cp, cpd_s2 = s2op
                else:
                    op, cp = self._conv(L1, pend["x1"] if pend else x, inpl, raw1, Hc, Hc, Ho, Ho, stride, 1, stats=self.stats if S else None,
                                        xf=dict(pend["xf"], out=x) if pend else None)
                self.fwd.append(op)
                xf1 = self._xf_ok(L2, 1, Ho)  # bn1 + ReLU applied by conv2 on load (which also writes a1 for the weight gradient)
                self.fwd += self._bn_fwd(pfx + ".bn1", raw1, cp, pl, npo, None if xf1 else a1)
                s1 = self.bn[pfx + ".bn1"]
                op, cp = self._conv(L2, raw1 if xf1 else a1, pl, raw2, Ho, Ho, Ho, Ho, 1, 1, stats=self.stats if S else None,
                                    xf=dict(mode=1, a=s1.scale, c=s1.shift, relu=True, out=a1) if xf1 else None)
                self.fwd.append(op)
                # block output relu(bn2(raw2) + identity): left to the next block's conv1 when that is a halo launch
                nxt1 = e.layers.get(f"backbone.layer{li}.{b + 1}.conv1")
                xf_out = nxt1 is not None and self._xf_ok(nxt1, 2, Ho)
                if Ld is not None:
                    rawd, idn = (rawd_ if s2op is not None else E(B, Ho, Ho, pl)), (None if xf_out else E(B, Ho, Ho, pl))
                    self.tensors[pfx + ".rawd"] = rawd
                    if idn is not None:
                        self.tensors[pfx + ".idn"] = idn
                    self.fwd += self._bn_fwd(pfx + ".bn2", raw2, cp, pl, npo, None)
                    if s2op is not None:   # (the shortcut conv ran with conv1: only its BatchNorm's finalize is left, from its own rows)
                        self.fwd += self._bn_fwd(pfx + ".downsample.1", rawd, cpd_s2, pl, npo, idn, relu=0, stats=self.stats_b)
                    else:
                        op, cpd = self._conv(Ld, x, inpl, rawd, Hc, Hc, Ho, Ho, stride, 0, stats=self.stats if S else None)
                        self.fwd.append(op)
                        self.fwd += self._bn_fwd(pfx + ".downsample.1", rawd, cpd, pl, npo, idn, relu=0)
                    s2, sd = self.bn[pfx + ".bn2"], self.bn[pfx + ".downsample.1"]
                    if xf_out:   # relu(scale2*raw2 + scale_d*rawd + shift2 + shift_d)
                        nxt_pend = dict(x1=raw2, xf=dict(mode=2, x2=rawd, a=s2.scale, b=sd.scale, c=s2.shift, c2=sd.shift, relu=True))
                    else:
                        self.fwd.append(lambda st, ctx, raw2=raw2, s2=s2, idn=idn, out=out, npo=npo, pl=pl: check(
                            lib.gdrn_bn_apply(ptr(raw2), ptr(s2.scale), ptr(s2.shift), ptr(idn), ptr(out), npo, pl, 1, e.dt, st), "bn_apply"))
                else:
                    self.fwd += self._bn_fwd(pfx + ".bn2", raw2, cp, pl, npo, None if xf_out else out, residual=x)
                    s2 = self.bn[pfx + ".bn2"]
                    if xf_out:   # relu(scale2*raw2 + x + shift2)
                        nxt_pend = dict(x1=raw2, xf=dict(mode=2, x2=x, a=s2.scale, c=s2.shift, relu=True))
                if not xf_out:
                    nxt_pend = None
                if T:
                    d_out = E(B, Ho, Ho, pl)
                    d_raw2, d_a1, d_raw1 = E(B, Ho, Ho, pl), E(B, Ho, Ho, pl), E(B, Ho, Ho, pl)
                    self.tensors.update({pfx + ".d_out": d_out, pfx + ".d_raw2": d_raw2, pfx + ".d_a1": d_a1, pfx + ".d_raw1": d_raw1})
                    # d_out comes from the next block's conv1 data gradient when that is a plain (stride-1) block of the
                    # same layer: its halo epilogue has then already applied this block's output ReLU mask and reduced
                    # the bn2-backward sums (and d_out itself is the residual-path gradient g2)
                    # ... or, for the last block of a layer, the generic kernel's epilogue (next layer's stride-2 conv1 / the head's ConvTranspose)
                    pre2 = self._fusable(e.layers[f"backbone.layer{li}.{b + 1}.conv1"]) if b + 1 < nb else e.gemm_bnb
                    # BatchNorm-backward apply passes fused into the data-gradient conv that consumes their result (xf modes 3 / 4)
                    xfb2 = pre2 and self._xf_ok(L2, 3, Ho)
                    need_dx = d_x is not None
                    xfb1 = self._fusable(L2) and self._xf_ok(L1, 3, Ho) and Ld is None and need_dx
                    xd2 = xd1 = None
                    if xfb2:
                        g2 = d_out
                        grp, xd2 = self._bn_bwd(pfx + ".bn2", d_out, None, raw2, d_raw2, prereduced=True, xf=True)
                    elif pre2:
                        g2 = d_out
                        grp = self._bn_bwd(pfx + ".bn2", d_out, None, raw2, d_raw2, prereduced=True)
                    else:
                        g2 = E(B, Ho, Ho, pl)
                        self.tensors[pfx + ".g2"] = g2
                        grp = self._bn_bwd(pfx + ".bn2", d_out, out, raw2, d_raw2, g_out=g2)
                    grp.append(self._wgrad(L2, a1, d_raw2, Ho, Ho, Ho, Ho, 1, 1, pl, pl, pl, pl))
                    grp.append(self._unpack(L2))
                    pre1 = self._fusable(L2)
                    op, _ = self._conv(L2, d_out if xfb2 else d_raw2, pl, d_a1, Ho, Ho, Ho, Ho, 1, 1, w=L2.wd, rows=L2.rows_d, cin=L2.cin_d, cout=pl,
                                       bnb=(pfx + ".bn1", raw1, None, True) if pre1 else None, xf=xd2)
                    grp.append(op)
                    if xfb1:
                        ops1, xd1 = self._bn_bwd(pfx + ".bn1", d_a1, a1, raw1, d_raw1, affine_mask=True, prereduced=True, xf=True)
                        grp += ops1
                    else:
                        grp += self._bn_bwd(pfx + ".bn1", d_a1, a1, raw1, d_raw1, affine_mask=True, prereduced=pre1)
                    grp.append(self._wgrad(L1, x, d_raw1, Hc, Hc, Ho, Ho, stride, 1, inpl, pl, inpl, pl))
                    grp.append(self._unpack(L1))
                    if Ld is not None:
                        d_rawd, d_xd = E(B, Ho, Ho, pl), E(B, Hc, Hc, inpl)
                        self.tensors.update({pfx + ".d_rawd": d_rawd, pfx + ".d_xd": d_xd})
                        grp += self._bn_bwd(pfx + ".downsample.1", g2, None, rawd, d_rawd)
                        grp.append(self._wgrad(Ld, x, d_rawd, Hc, Hc, Ho, Ho, stride, 0, inpl, pl, inpl, pl, defer=li >= 3))
                        grp.append(self._unpack(Ld))
                        pbn = f"backbone.layer{li - 1}.{RESNET34_LAYERS[li - 2] - 1}.bn2"
                        # (r6) conv1's data gradient, the shortcut conv's data gradient and the previous block's ReLU mask + bn2-backward sums as
                        # ONE launch of the parity-class kernel: no d_xd tensor (three quarters zeros), no addend pass
                        s2d = self._s2d(L1, d_raw1, d_x, Ho, Hc, inpl, Ld=Ld, dyd=d_rawd, bnb=(pbn, prev_raw2, x)) if e.gemm_bnb else None
                        if s2d is not None:
                            grp.append(s2d)
                            del self.tensors[pfx + ".d_xd"]
                        else:
                            op, _ = self._conv(Ld, d_rawd, pl, d_xd, Ho, Ho, Hc, Hc, 2, 0, mode=1, w=Ld.wd, rows=Ld.rows_d, cin=Ld.cin_d, cout=inpl)
                            grp.append(op)
                            # d_x = gradient w.r.t. the previous layer's last block output (mask = x): its bn2 backward is reduced here
                            op, _ = self._conv(L1, d_raw1, pl, d_x, Ho, Ho, Hc, Hc, 2, 1, mode=1, w=L1.wd, rows=L1.rows_d, cin=L1.cin_d,
                                               cout=inpl, addend=d_xd, add_cs=inpl, bnb=(pbn, prev_raw2, x, False) if e.gemm_bnb else None)
                            grp.append(op)
                    elif need_dx:
                        # previous block of the same layer: d_x is the gradient w.r.t. its output (mask = x, stored) and
                        # feeds its bn2 backward (raw input prev_raw2)
                        fuse_prev = b >= 1 and self._fusable(L1)
                        op, _ = self._conv(L1, d_a1 if xfb1 else d_raw1, pl, d_x, Ho, Ho, Hc, Hc, 1, 1, w=L1.wd, rows=L1.rows_d, cin=L1.cin_d, cout=inpl,
                                           addend=g2, add_cs=pl,
                                           bnb=(f"backbone.layer{li}.{b - 1}.bn2", prev_raw2, x, False) if fuse_prev else None, xf=xd1)
                        grp.append(op)
                    self.bwd_groups.append(grp)
                    d_x = d_out
                x, Hc, inpl, prev_raw2, pend = out, Ho, pl, raw2, nxt_pend
        assert pend is None
        feat, d_feat = x, d_x

        # ---------------- geometric head
        h = "rot_head_net.features."
        LT = e.layers[h + "0"]
        rawt, h0 = E(B, 16, 16, 256), E(B, 16, 16, 256)
        self.tensors.update({h + "0.raw": rawt, h + "0.act": h0})
        xf_first = (not FOLD) and (not HEAD_CONVS[0][2]) and self._xf_ok(e.layers[h + str(HEAD_CONVS[0][0])], 1, 16)
        # (r6) the ConvTranspose's forward pass on the parity-class kernel (8 x 8 maps: two images per tile); the generic kernel keeps odd batch sizes
        if FOLD:
            tf = self._convT_fwd(LT, h + "1", feat, h0, 8, 16, evalmode=True)
            self.fwd.append(tf[0] if tf is not None else
                            self._conv_bn_eval(LT, h + "1", feat, 512, h0, 8, 8, 16, 16, 2, 1, relu=True, mode=1, cin=512, cout=256))
        else:
            tf = self._convT_fwd(LT, h + "1", feat, rawt, 8, 16, stats=self.stats if S else None)
            op, cp = tf if tf is not None else self._conv(LT, feat, 512, rawt, 8, 8, 16, 16, 2, 1, mode=1, cin=512, cout=256, stats=self.stats if S else None)
            self.fwd.append(op)
            self.fwd += self._bn_fwd(h + "1", rawt, cp, 256, B * 256, None if xf_first else h0)
        if T:
            d_h0, d_rawt = E(B, 16, 16, 256), E(B, 16, 16, 256)
            self.tensors.update({h + "0.d_act": d_h0, h + "0.d_raw": d_rawt})
            pre_t = (not HEAD_CONVS[0][2]) and self._fusable(e.layers[h + str(HEAD_CONVS[0][0])])
            grp = self._bn_bwd(h + "1", d_h0, h0, rawt, d_rawt, affine_mask=True, prereduced=pre_t)
            # ConvT weight grad = conv wgrad with roles swapped: "input" d_rawt (16x16, 256), "output grad" feat (8x8, 512)
            grp.append(self._wgrad(LT, d_rawt, feat, 16, 16, 8, 8, 2, 1, 256, 512, 256, 512))
            grp.append(self._unpack(LT))
            # (r6) the ConvTranspose's data gradient = a stride-2 conv of the output gradient: the parity-plane kernel with the BatchNorm-backward epilogue
            op = self._convT_dgrad(LT, d_rawt, d_feat, 16, 8, ("backbone.layer4.2.bn2", prev_raw2, feat)) if e.gemm_bnb else None
            if op is None:
                op, _ = self._conv(LT, d_rawt, 256, d_feat, 16, 16, 8, 8, 2, 1, mode=0, w=LT.wd, rows=LT.rows_d, cin=256, cout=512,
                                   bnb=("backbone.layer4.2.bn2", prev_raw2, feat, False) if e.gemm_bnb else None)
            grp.append(op)
            self.bwd_groups.append(grp)
        hx, d_hx, Hh = h0, (d_h0 if T else None), 16
        prev_bn, prev_raw = h + "1", rawt
        # pend_h: hx = relu(bn(prev_raw)) is evaluated (and written to hx) by the next head conv while it stages its patch
        pend_h = dict(x1=rawt, xf=dict(mode=1, a=self.bn[h + "1"].scale, c=self.bn[h + "1"].shift, relu=True)) if xf_first else None
        up_next = False
        for hi, (ci, bi, up) in enumerate(HEAD_CONVS):
            Lc = e.layers[h + str(ci)]
            grp = []
            fused_up, pre_sums = bool(up and up_next), False
            if up:
                assert pend_h is None
                u = E(B, 2 * Hh, 2 * Hh, 256)
                self.tensors[h + f"{ci}.up"] = u
                sp_ = self.bn.get(prev_bn)
                if fused_up:
                    # (r6) BatchNorm + ReLU of the previous conv evaluated by the upsampling itself (its activation tensor is never stored), and
                    # in the backward pass the upsampling's adjoint also reduces that BatchNorm's backward sums: two launches and two tensor
                    # round trips less per upsampling
                    self.fwd.append(lambda st, ctx, r_=prev_raw, sp_=sp_, u=u, Hh=Hh: check(
                        lib.gdrn_bn_relu_upsample2x_fwd(ptr(r_), ptr(sp_.scale), ptr(sp_.shift), ptr(u), B, Hh, Hh, 256, e.dt, st), "bn_relu_upsample_fwd"))
                else:
                    self.fwd.append(lambda st, ctx, hx=hx, u=u, Hh=Hh: check(lib.gdrn_upsample2x_fwd(ptr(hx), ptr(u), B, Hh, Hh, 256, e.dt, st), "upsample_fwd"))
                if T:
                    d_u = E(B, 2 * Hh, 2 * Hh, 256)
                    self.tensors[h + f"{ci}.d_up"] = d_u
                    if fused_up:
                        up_bwd = (lambda st, ctx, d_u=d_u, d_hx=d_hx, Hh=Hh, r_=prev_raw, sp_=sp_: check(
                            lib.gdrn_upsample2x_bwd_bnsums(ptr(d_u), ptr(d_hx), ptr(r_), ptr(sp_.mean), ptr(sp_.invstd), ptr(sp_.scale), ptr(sp_.shift), B, Hh, Hh,
                                                           256, ptr(self.stats), e.dt, st), "upsample_bwd_bnsums"))
                    else:
                        up_bwd = (lambda st, ctx, d_u=d_u, d_hx=d_hx, Hh=Hh: check(lib.gdrn_upsample2x_bwd(ptr(d_u), ptr(d_hx), B, Hh, Hh, 256, e.dt, st), "upsample_bwd"))
                    d_in = d_u
                xin, Hh = u, 2 * Hh
            else:
                xin = hx
                d_in = d_hx
                up_bwd = None
            raw, act = E(B, Hh, Hh, 256), E(B, Hh, Hh, 256)
            self.tensors[h + f"{ci}.raw"] = raw
            nxt = HEAD_CONVS[hi + 1] if hi + 1 < len(HEAD_CONVS) else None
            # this conv's BatchNorm + ReLU is applied by the next head conv on load when no upsampling sits in between
            xf_next = (not FOLD) and nxt is not None and not nxt[2] and self._xf_ok(e.layers[h + str(nxt[0])], 1, Hh)
            # the NEXT conv sits behind an upsampling that applies this conv's BatchNorm + ReLU itself (see `fused_up` above)
            up_next = (not FOLD) and S and e.fuse_up and nxt is not None and nxt[2] and (T or not WL)
            if FOLD:
                self.fwd.append(self._conv_bn_eval(Lc, h + str(bi), xin, 256, act, Hh, Hh, Hh, Hh, 1, 1, relu=True))
            else:
                op, cp = self._conv(Lc, pend_h["x1"] if pend_h else xin, 256, raw, Hh, Hh, Hh, Hh, 1, 1, stats=self.stats if S else None,
                                    xf=dict(pend_h["xf"], out=xin) if pend_h else None)
                self.fwd.append(op)
                self.fwd += self._bn_fwd(h + str(bi), raw, cp, 256, B * Hh * Hh, None if (xf_next or up_next) else act)
            if not up_next:
                self.tensors[h + f"{ci}.act"] = act   # (behind a fused upsampling the activation is never stored)
            if T:
                d_act, d_raw = E(B, Hh, Hh, 256), E(B, Hh, Hh, 256)
                self.tensors.update({h + f"{ci}.d_act": d_act, h + f"{ci}.d_raw": d_raw})
                # d_act is produced by the NEXT head conv's data gradient; when no upsampling sits in between, that
                # launch masks it and reduces this BN's backward sums
                pre = (not nxt[2] and self._fusable(e.layers[h + str(nxt[0])])) if nxt is not None else e.gemm_bnb  # (last conv: the 1x1 output conv's data gradient)
                xfb = self._xf_ok(Lc, 3 if pre else 4, Hh)  # this BN's backward apply inside Lc's data-gradient launch (mode 3, or 4 = with the ReLU mask)
                xd = None
                sd_ = bool(up_next and not pre)   # the fused upsampling's adjoint behind this conv has reduced this BatchNorm's backward sums
                if xfb:
                    ops, xd = self._bn_bwd(h + str(bi), d_act, act, raw, d_raw, affine_mask=True, prereduced=pre, xf=True, sums_done=sd_)
                    grp += ops
                else:
                    grp += self._bn_bwd(h + str(bi), d_act, act, raw, d_raw, affine_mask=True, prereduced=pre, sums_done=sd_)
                grp.append(self._wgrad(Lc, xin, d_raw, Hh, Hh, Hh, Hh, 1, 1, 256, 256, 256, 256))
                grp.append(self._unpack(Lc))
                fuse_in = (not up) and self._fusable(Lc)  # d_in is the gradient w.r.t. the previous BN+ReLU's output
                op, _ = self._conv(Lc, d_act if xfb else d_raw, 256, d_in, Hh, Hh, Hh, Hh, 1, 1, w=Lc.wd, rows=Lc.rows_d, cin=256, cout=256,
                                   bnb=(prev_bn, prev_raw, None, True) if fuse_in else None, xf=xd)
                grp.append(op)
                if up_bwd is not None:
                    grp.append(up_bwd)
                self.bwd_groups.append(grp)
                d_hx = d_act
            pend_h = dict(x1=raw, xf=dict(mode=1, a=self.bn[h + str(bi)].scale, c=self.bn[h + str(bi)].shift, relu=True)) if xf_next else None
            hx, prev_bn, prev_raw = act, h + str(bi), raw
        assert pend_h is None
        LO = e.layers[h + "23"]
        M = B * 64 * 64
        self.hs = 72
        self.head_out = E(M, self.hs, dtype=F32t)
        bias_o = e.P[h + "23.bias"]
        # 64 regions, 16-bit: the 1x1 output conv and the head tail as ONE kernel (gdrn_head_conv_tail_fwd; with losses gdrn_head_conv_tail_loss_fwd,
        # which also accumulates the map-loss sums) -- the fp32 logits still go to head_out (the backward pass and callers that return the maps
        # read them), but the forward pass does not read them back
        fused_tail = e.h16 and nreg == 64
        if not fused_tail:
            op, _ = self._conv(LO, hx, 256, self.head_out, 64, 64, 64, 64, 1, 0, bias=bias_o, out_f32=1, y_cs=self.hs, cout=e.head_c)
            self.fwd.append(op)
        self.pnp_in = e._zeros(M, 128)   # channels >= 5 + nreg stay zero (PREZEROED: the kernels do not re-write the pad)
        self.keep.append(self.pnp_in)
        self.tensors.update({"head_out": self.head_out, "pnp_in": self.pnp_in})
        if fused_tail and WL:
            self.acc_rows = int(lib.gdrn_head_conv_tail_loss_rows(B, 4096))
            assert self.acc_rows > 0
            self.acc = E(8 + 8 * self.acc_rows, dtype=torch.float64)
            self.losses = e._zeros(8, dtype=F32t)

            def head_conv_tail(st, ctx, hx=hx):
                check(lib.gdrn_head_conv_tail_loss_fwd(ptr(hx), 256, ptr(LO.wf), LO.rows_f, ptr(bias_o), ctx["coord2d"], ctx["extents"], ptr(self.head_out), self.hs,
                                                       ptr(self.pnp_in), 128, ctx["gt_xyz"], ctx["mask_visib"], ctx["mask_trunc"], ctx["gt_region"], ptr(self.acc),
                                                       B, 4096, nreg, e.dt | PREZEROED, st), "head_conv_tail_loss_fwd")

            head_conv_tail.meta = dict(kernel="head_conv_tail64_kernel<bf16,true>", flops=2.0 * M * 256 * e.head_c, layer=h + "23+tail+losses")
            self.fwd.append(head_conv_tail)
        elif fused_tail:
            def head_conv_tail(st, ctx, hx=hx):
                # ctx["want_maps"] False (GDRN.forward without cfg.TEST.USE_PNP): nobody reads the logits -- head = NULL saves their 75 MB at bs = 64
                # (head_valid: does plan.head_out hold THIS forward's logits?  readers -- GDRN._maps, tools -- assert it: ADVICE r5)
                self.head_valid = bool(ctx.get("want_maps", True))
                check(lib.gdrn_head_conv_tail_fwd(ptr(hx), 256, ptr(LO.wf), LO.rows_f, ptr(bias_o), ctx["coord2d"], ctx["extents"],
                                                  ptr(self.head_out) if ctx.get("want_maps", True) else None, self.hs,
                                                  ptr(self.pnp_in), 128, B, 4096, nreg, e.dt | PREZEROED, st), "head_conv_tail_fwd")

            head_conv_tail.meta = dict(kernel="head_conv_tail64_kernel<bf16,false>", flops=2.0 * M * 256 * e.head_c, layer=h + "23+tail")
            self.fwd.append(head_conv_tail)
        elif WL:
            # map-loss sums: totals in acc[0..7], behind them one partial row per workgroup of the kernel (ACC_ROWS: stored, then added in a fixed
            # order by map_loss_finalize_rows -- no memset launch, no atomics, run-to-run identical losses)
            rows_on = True
            self.acc_rows = int(lib.gdrn_head_tail_loss_rows(B, 4096, nreg, self.hs, 128)) if rows_on else 0
            if rows_on and self.acc_rows <= 0:
                check(self.acc_rows or -1, "head_tail_loss_rows")
            self.acc = E(8 + 8 * self.acc_rows, dtype=torch.float64)
            self.losses = e._zeros(8, dtype=F32t)
            # head tail + map-loss sums in one pass over the logits (train mode)
            self.fwd.append(lambda st, ctx: check(lib.gdrn_head_tail_loss_fwd(ptr(self.head_out), self.hs, ctx["coord2d"], ctx["extents"], ptr(self.pnp_in), 128,
                                                                              ctx["gt_xyz"], ctx["mask_visib"], ctx["mask_trunc"], ctx["gt_region"], ptr(self.acc),
                                                                              B, 4096, nreg, e.dt | PREZEROED | (ACC_ROWS if self.acc_rows else 0), st), "head_tail_loss_fwd"))
        else:
            self.fwd.append(lambda st, ctx: check(lib.gdrn_head_tail_fwd(ptr(self.head_out), self.hs, ctx["coord2d"], ctx["extents"],
                                                                         ptr(self.pnp_in), 128, B, 4096, nreg, e.dt | PREZEROED, st), "head_tail_fwd"))
        if T:
            self.d_head = e._zeros(M, 128)   # pad channels zero-filled once (PREZEROED)
            self.keep.append(self.d_head)
            self.d_pnp_in = E(M, 128)
            self.gw = e._zeros(8, dtype=F32t)
            self.tensors.update({"d_head": self.d_head, "d_pnp_in": self.d_pnp_in})
            grp = [lambda st, ctx: check(lib.gdrn_head_tail_bwd(ptr(self.head_out), self.hs, ptr(self.pnp_in), ptr(self.d_pnp_in), 128,
                                                                ctx["extents"], ctx["gt_xyz"], ctx["mask_visib"], ctx["mask_trunc"],
                                                                ctx["gt_region"], ptr(self.acc), ptr(self.gw), ptr(self.d_head), 128, B,
                                                                4096, nreg, e.dt | PREZEROED, st), "head_tail_bwd")]
            grp.append(self._wgrad(LO, hx, self.d_head, 64, 64, 64, 64, 1, 0, 256, e.head_c, 256, 128))
            grp.append(self._unpack(LO))
            gb = e.grads[h + "23.bias"]
            self.grad_group[h + "23.bias"] = len(self.bwd_groups)
            self._zero_regions.append(self._grad16(gb))
            grp.append(self._side(lambda st, ctx: check(lib.gdrn_bias_grad(ptr(self.d_head), 128, M, e.head_c, ptr(gb), e.dt | PREZEROED, st), "bias_grad")))
            if e.h16 and e.gemm_bnb:
                # the 1x1 output conv's data gradient on its own kernel (K = 69: the generic 128 x 128 tile ran it at 83 TFLOP/s), with the ReLU mask and
                # the BatchNorm-backward sums of the head's last BatchNorm in the epilogue as the generic kernel's bnb_* epilogue has them
                sbp = self.bn[prev_bn]
                nrows_h = int(lib.gdrn_head_out_dgrad_rows(B, 4096))
                assert 0 < nrows_h and nrows_h * 2 * 256 <= self.stats.numel()

                def head_dgrad(st, ctx, d_hx=d_hx, prev_raw=prev_raw, sbp=sbp):
                    check(lib.gdrn_head_out_dgrad(ptr(self.d_head), 128, ptr(LO.wd), LO.cin_d, ptr(prev_raw), 256, ptr(sbp.mean), ptr(sbp.invstd), ptr(sbp.scale),
                                                  ptr(sbp.shift), ptr(d_hx), 256, ptr(self.stats), B, 4096, e.dt, st), "head_out_dgrad")

                coef_h = self._bn_coef_op(prev_bn, self.stats, nrows_h)

                def op(st, ctx):
                    head_dgrad(st, ctx)
                    coef_h(st, ctx)

                op.parts = (head_dgrad, coef_h)
                op.meta = dict(kernel="head_out_dgrad64_kernel<bf16>", flops=2.0 * M * 256 * e.head_c, bytes=float(M * (128 + 256 + 256) * 2), layer=h + "23:dgrad")
            else:
                op, _ = self._conv(LO, self.d_head, 128, d_hx, 64, 64, 64, 64, 1, 0, w=LO.wd, rows=LO.rows_d, cin=128, cout=256,
                                   bnb=(prev_bn, prev_raw, None, True) if e.gemm_bnb else None)
            grp.append(op)
            self.bwd_groups.append(grp)

        # ---------------- Patch-PnP
        q = "pnp_net.features."
        px, d_px, Hp, cin = self.pnp_in, (self.d_pnp_in if T else None), 64, 128
        for ci, gi in ((0, 1), (3, 4), (6, 7)):
            Lc = e.layers[q + str(ci)]
            Ho = Hp // 2
            r, gact = E(B, Ho, Ho, 128), E(B, Ho, Ho, 128)
            mr = E(B, 32, 2, dtype=F32t)
            self.tensors.update({q + f"{ci}.raw": r, q + f"{ci}.act": gact})
            s2op = self._s2(Lc, px, cin, r, Hp, Ho) if cin == Lc.cin_f else None   # (r6: the parity-plane kernel on the 64 -> 32 and 32 -> 16 convs)
            if s2op is not None:
                op = s2op[0]
            else:
                op, _ = self._conv(Lc, px, cin, r, Hp, Hp, Ho, Ho, 2, 1, cin=cin, cout=128)
            self.fwd.append(op)
            gam, bet = e.P[q + f"{gi}.weight"], e.P[q + f"{gi}.bias"]
            self.fwd.append(lambda st, ctx, r=r, gam=gam, bet=bet, gact=gact, mr=mr, Ho=Ho: check(
                lib.gdrn_gn_relu_fwd(ptr(r), ptr(gam), ptr(bet), ptr(gact), ptr(mr), B, Ho * Ho, 128, 32, 1e-5, e.dt, st), "gn_relu_fwd"))
            if T:
                d_g, d_r = E(B, Ho, Ho, 128), E(B, Ho, Ho, 128)
                self.tensors.update({q + f"{ci}.d_act": d_g, q + f"{ci}.d_raw": d_r})
                dgam, dbet = e.grads[q + f"{gi}.weight"], e.grads[q + f"{gi}.bias"]
                self.grad_group[q + f"{gi}.weight"] = self.grad_group[q + f"{gi}.bias"] = len(self.bwd_groups)
                grp = [lambda st, ctx, d_g=d_g, gact=gact, r=r, gam=gam, mr=mr, d_r=d_r, dgam=dgam, dbet=dbet, Ho=Ho: check(
                    lib.gdrn_gn_relu_bwd(ptr(d_g), ptr(gact), ptr(r), ptr(gam), ptr(mr), ptr(d_r), ptr(dgam), ptr(dbet), B, Ho * Ho, 128, 32,
                                         e.dt | PREZEROED, st), "gn_relu_bwd")]
                self._zero_regions += [self._grad16(dgam), self._grad16(dbet)]
                grp.append(self._wgrad(Lc, px, d_r, Hp, Hp, Ho, Ho, 2, 1, cin, 128, cin, 128))
                grp.append(self._unpack(Lc))
                op = self._s2d(Lc, d_r, d_px, Ho, Hp, cin) if cin == Lc.cin_f else None   # (r6: the parity-class kernel where it covers the map)
                if op is None:
                    op, _ = self._conv(Lc, d_r, 128, d_px, Ho, Ho, Hp, Hp, 2, 1, mode=1, w=Lc.wd, rows=Lc.rows_d, cin=128, cout=cin)
                grp.append(op)
                self.bwd_groups.append(grp)
                d_px = d_g
            px, Hp, cin = gact, Ho, 128
        g2act, d_g2 = px, d_px
        L1, L2, L3 = e.layers["pnp_net.fc1"], e.layers["pnp_net.fc2"], e.layers["pnp_net.fc_rt"]
        f1, f2 = E(B, 1024), E(B, 256)
        self.fc_out = e._zeros(B, 64, dtype=F32t)   # (columns 9 .. 63 are never written)
        self.keep.append(self.fc_out)
        self.tensors.update({"pnp_net.fc1.act": f1, "pnp_net.fc2.act": f2, "fc_out": self.fc_out})
        b1, b2 = e.P["pnp_net.fc1.bias"], e.P["pnp_net.fc2.bias"]
        if e.h16 and B <= 64 and e.fc_splitk:
            # fc1 is bound by reading its 16.8 MB of weights once: split-K skinny GEMM instead of 8 gather workgroups.
            # L1.wf = [1024 rows][64 taps][128 ch] = row-major [N][K] in the (pixel, channel) order of the NHWC input
            ws1 = e._zeros(16 * B * 1024 + 64, dtype=F32t)  # GDRN_LINEAR_MAX_SPLITS slabs + tickets
            self.keep.append(ws1)

            def fc1_fwd(st, ctx):
                check(lib.gdrn_linear_splitk(ptr(g2act), ptr(L1.wf), ptr(b1), ptr(f1), B, 8192, 1024, 8192, 8192, 1024, 2, ptr(ws1), e.dt, st),
                      "linear_splitk fc1")

            fc1_fwd.meta = dict(kernel="linear_splitk_kernel", flops=2.0 * B * 8192 * 1024, layer="pnp_net.fc1")
            self.fwd.append(fc1_fwd)
        else:
            op, _ = self._conv(L1, g2act, 128, f1, 8, 8, 1, 1, 1, 0, bias=b1, act=2, cin=128, cout=1024)
            self.fwd.append(op)
        if e.h16 and B <= 64 and e.fc_splitk:
            # fc2 (64 x 1024 -> 256) on the gather kernel is two workgroups walking K = 1024 serially (20 us for 34 MFLOP): the same split-K
            # kernel, 16 column tiles x 8 K ranges (r4)
            ws2 = e._zeros(16 * B * 256 + 64, dtype=F32t)
            self.keep.append(ws2)

            # (r6) fc2's finish pass (bias, LeakyReLU), fc_r | fc_t and the pose decode are ONE launch (gdrn_pose_loss, fc2_ws): fc2 leaves its slabs
            self.fc_tail = (ws2, int(lib.gdrn_linear_splits(1024, 256))) if e.fc_tail else None
            act2 = -1 if self.fc_tail else 2

            def fc2_fwd(st, ctx):
                check(lib.gdrn_linear_splitk(ptr(f1), ptr(L2.wf), ptr(b2), ptr(f2), B, 1024, 256, 1024, 1024, 256, act2, ptr(ws2), e.dt, st),
                      "linear_splitk fc2")

            fc2_fwd.meta = dict(kernel="linear_splitk_kernel", flops=2.0 * B * 1024 * 256, layer="pnp_net.fc2")
            self.fwd.append(fc2_fwd)
        else:
            op, _ = self._conv(L2, f1, 1024, f2, 1, 1, 1, 1, 1, 0, bias=b2, act=2, cin=1024, cout=256)
            self.fwd.append(op)
        if getattr(self, "fc_tail", None) is None:
            op, _ = self._conv(L3, f2, 256, self.fc_out, 1, 1, 1, 1, 1, 0, bias=e.rt_b, out_f32=1, cin=256, cout=9, y_cs=64)
            self.fwd.append(op)
        self._fc2 = (f2, b2, L3)
        if T:
            self.dfc3 = e._zeros(3, B, 64, dtype=F32t)
            d_fc32 = e._zeros(B, 64, dtype=F32t)
            d_fc = E(B, 64)
            d_f2, d_f2p, d_f1, d_f1p = E(B, 256), E(B, 256), E(B, 1024), E(B, 1024)
            self.tensors.update({"d_fc": d_fc, "pnp_net.fc2.d_act": d_f2, "pnp_net.fc2.d_pre": d_f2p, "pnp_net.fc1.d_act": d_f1, "pnp_net.fc1.d_pre": d_f1p})
            o_rt = e.grad_offsets["pnp_net.fc_r.bias"]   # the two bias gradients' shared slot of the flat gradient buffer (Engine.__init__)
            assert e.grad_offsets["pnp_net.fc_t.bias"] == o_rt + 6 and o_rt % 4 == 0
            self._rt_gb_full = e.grad_flat[o_rt:o_rt + 12]
            self.rt_gb = self._rt_gb_full[:9]
            self._zero_regions += [self._rt_gb_full, self._grad16(e.grads["pnp_net.fc1.bias"]), self._grad16(e.grads["pnp_net.fc2.bias"])]
            g_b1, g_b2 = e.grads["pnp_net.fc1.bias"], e.grads["pnp_net.fc2.bias"]
            for n_ in ("pnp_net.fc1.bias", "pnp_net.fc2.bias", "pnp_net.fc_r.bias", "pnp_net.fc_t.bias"):
                self.grad_group[n_] = len(self.bwd_groups)
            # (skipped when the forward pass was seeded: gdrn_pose_loss has then written dL/dfc itself -- ctx["seeded"], GDRN.train_step)
            comb = lambda st, ctx: check(lib.gdrn_combine3(ptr(self.dfc3), self.gw.data_ptr() + 20, ptr(d_fc32), B * 64, st), "combine3")
            cast = lambda st, ctx: check(lib.gdrn_cast_from_f32(ptr(d_fc32), ptr(d_fc), B * 64, e.dt, st), "cast")
            comb.seed = cast.seed = True
            self.d_fc = d_fc
            grp = [
                comb,
                cast,
                self._wgrad(L3, f2, d_fc, 1, 1, 1, 1, 1, 0, 256, 9, 256, 64, defer=True),
                self._unpack(L3),
                lambda st, ctx: check(lib.gdrn_bias_grad(ptr(d_fc), 64, B, 9, ptr(self.rt_gb), e.dt | PREZEROED, st), "bias_grad"),   # = both .grad views
            ]
            op, _ = self._conv(L3, d_fc, 64, d_f2, 1, 1, 1, 1, 1, 0, w=L3.wd, rows=L3.rows_d, cin=64, cout=256)
            grp.append(op)
            grp.append(lambda st, ctx: check(lib.gdrn_leaky_bwd(ptr(d_f2), ptr(f2), ptr(d_f2p), B * 256, e.dt, st), "leaky_bwd"))
            grp.append(self._wgrad(L2, f1, d_f2p, 1, 1, 1, 1, 1, 0, 1024, 256, 1024, 256, defer=True))
            grp.append(self._unpack(L2))
            grp.append(self._side(lambda st, ctx: check(lib.gdrn_bias_grad(ptr(d_f2p), 256, B, 256, ptr(g_b2), e.dt | PREZEROED, st), "bias_grad"), defer=True))
            op, _ = self._conv(L2, d_f2p, 256, d_f1, 1, 1, 1, 1, 1, 0, w=L2.wd, rows=L2.rows_d, cin=256, cout=1024)
            grp.append(op)
            grp.append(lambda st, ctx: check(lib.gdrn_leaky_bwd(ptr(d_f1), ptr(f1), ptr(d_f1p), B * 1024, e.dt, st), "leaky_bwd"))
            grp.append(self._wgrad(L1, g2act, d_f1p, 8, 8, 1, 1, 1, 0, 128, 1024, 128, 1024, defer=True))
            grp.append(self._unpack(L1))
            grp.append(self._side(lambda st, ctx: check(lib.gdrn_bias_grad(ptr(d_f1p), 1024, B, 1024, ptr(g_b1), e.dt | PREZEROED, st), "bias_grad"), defer=True))
            if e.h16 and B <= 64 and e.fc_splitk and e.fc_tail:
                # (r6) fc1's data gradient is bound by reading its 16.8 MB operand once, like its forward pass: the split-K skinny GEMM (L1.wd =
                # [8192 rows = (pixel, channel)][1024] row-major) instead of 64 gather workgroups walking K = 1024 each (39 us -> 15 + 7)
                ws1d = e._zeros(16 * B * 8192 + 64, dtype=F32t)
                self.keep.append(ws1d)

                def fc1_dgrad(st, ctx):
                    check(lib.gdrn_linear_splitk(ptr(d_f1p), ptr(L1.wd), None, ptr(d_g2), B, 1024, 8192, 1024, 1024, 8192, 0, ptr(ws1d), e.dt, st),
                          "linear_splitk fc1:dgrad")

                fc1_dgrad.meta = dict(kernel="linear_splitk_kernel", flops=2.0 * B * 8192 * 1024, layer="pnp_net.fc1:dgrad")
                grp.append(fc1_dgrad)
            else:
                op, _ = self._conv(L1, d_f1p, 1024, d_g2, 1, 1, 1, 1, 1, 0, w=L1.wd, rows=L1.rows_d, cin=1024, cout=8192, KH=1, KW=1, y_cs=8192)
                grp.append(op)
            self.bwd_groups.append(grp)

        # ---------------- pose decode (+ pose / map losses in train mode)
        self.rot = E(B, 3, 3, dtype=F32t)
        self.trans = E(B, 3, dtype=F32t)
        self.vis = e._zeros(B, 2, dtype=F32t)
        self.pose_p = PoseParams()

        def pose(st, ctx):
            pp = self.pose_p
            pp.fc, pp.fs = ptr(self.fc_out), 64
            pp.cams, pp.centers, pp.whs, pp.ratios, pp.extents = ctx["cams"], ctx["centers"], ctx["whs"], ctx["ratios"], ctx["extents"]
            pp.gt_rot, pp.gt_trans, pp.gt_trans_ratio = ctx.get("gt_rot"), ctx.get("gt_trans"), ctx.get("gt_trans_ratio")
            pp.points, pp.npts = ctx.get("points"), ctx.get("npts", 0)
            pp.sym, pp.sym_count, pp.Kmax = ctx.get("sym"), ctx.get("sym_count"), ctx.get("Kmax", 0)
            pp.N, pp.train = B, 1 if WL else 0
            pp.rot, pp.trans = ptr(self.rot), ptr(self.trans)
            pp.losses = (self.losses.data_ptr() + 20) if WL else None
            pp.dfc = ptr(self.dfc3) if T else None
            pp.vis = ptr(self.vis) if WL else None
            pp.loss_rows = ptr(self.pose_rows) if (WL and self.acc_rows) else None
            seeded = bool(T and WL and e.h16 and ctx.get("seeded"))   # the fused train step: dL/dloss is known now, dL/dfc leaves this launch
            pp.gw, pp.dfc_comb = ((self.gw.data_ptr() + 20), ptr(self.d_fc)) if seeded else (None, None)
            if getattr(self, "fc_tail", None) is not None:
                f2_, b2_, L3_ = self._fc2
                pp.fc2_ws, pp.fc2_splits, pp.fc2_bias, pp.f2_out = ptr(self.fc_tail[0]), self.fc_tail[1], ptr(b2_), ptr(f2_)
                pp.w_rt, pp.b_rt, pp.fc_w = ptr(L3_.wf), ptr(e.rt_b), ptr(self.fc_out)
            check(lib.gdrn_pose_loss(C.byref(pp), st), "pose_loss")

        self.fwd.append(pose)
        if WL:
            if self.acc_rows:
                self.pose_rows = E(B, 4, dtype=F32t)
                # ctx["loss_w"] / ctx["weighted"] (device pointers, the fused train step): the weighted loss vector it returns, written here
                self.fwd.append(lambda st, ctx: check(lib.gdrn_loss_finalize(ptr(self.acc), self.acc_rows, B, 4096, ptr(self.pose_rows), ptr(self.losses),
                                                                             ctx.get("loss_w"), ctx.get("weighted"), st), "loss_finalize"))
            else:
                self.fwd.append(lambda st, ctx: check(lib.gdrn_map_loss_finalize(ptr(self.acc), B, 4096, ptr(self.losses), st), "map_loss_finalize"))

    # ---- execution ---------------------------------------------------------------------------
    def run_forward(self, ctx):
        e = self.e
        if e.dry:
            raise cabi.GdrnHipError("dry (build-only) engine: there is no CPU execution path")
        st = e._stream()
        self.generation += 1
        if self.bn_train:
            e.bn_epoch += 1  # the kernels update the running statistics behind autograd's back
        elif self.eval_prep:
            sig = (e.bn_epoch, tuple(t._version for t in e.P.values()), tuple(t._version for t in e.Bf.values()))
            if sig != self._eval_sig:
                for op in self.eval_prep:
                    op(st, ctx)
                self._eval_sig = sig
        for op in self.fwd:
            op(st, ctx)

    def run_backward(self, ctx, on_bucket=None):
        """ctx as in forward; self.gw must hold dL/dloss_k.  on_bucket(i) is called after the ops that
        complete gradient bucket i have been enqueued (used to overlap the RCCL all-reduce)."""
        e = self.e
        if e.dry:
            raise cabi.GdrnHipError("dry (build-only) engine: there is no CPU execution path")
        main = torch.cuda.current_stream(e.dev)
        st = main.cuda_stream
        ztab, zst, znt, znb = self._zero_tab
        check(e.lib.gdrn_zero_multi(ptr(ztab), ptr(zst), znt, znb, st), "zero_multi")
        marks = self._bucket_marks() if on_bucket is not None else {}
        # The bucket-end work (grouped weight gradients, their reduction, gradient unpack) only feeds the optimizer /
        # the RCCL exchange: it goes to a second stream behind an event, so the next bucket's dependent chain of short
        # data-gradient / BatchNorm kernels (one workgroup per CU on the small feature maps) shares the CUs with it.
        side = e.side_stream() if e.wgrad_stream else None
        used_side = in_side = False
        probe = getattr(self, "op_events", None)   # bench.py: [(start, end, meta, on_side)] HIP events around the conv launches of THIS pass
        seeded = bool(ctx.get("seeded")) and self.e.h16
        for i, op in enumerate(self.bwd):
            if seeded and getattr(op, "seed", False):
                continue   # dL/dfc was written by the forward pass's pose kernel
            if probe is not None and getattr(op, "meta", None) is not None:
                op = _probed(op, probe)
            if side is not None and getattr(op, "side", False):
                if not in_side:
                    side.wait_stream(main)  # everything enqueued so far on the main stream
                    in_side = used_side = True
                op(side.cuda_stream, ctx)
                if i in marks:
                    with torch.cuda.stream(side):
                        on_bucket(marks[i])
            else:
                in_side = False
                op(st, ctx)
                if i in marks:
                    if used_side:     # the bucket's side-stream work (weight-gradient reduction) is part of the bucket
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            on_bucket(marks[i])
                    else:
                        on_bucket(marks[i])
        if used_side:
            main.wait_stream(side)  # the optimizer (or the caller) sees complete gradients on the main stream

    def walk_backward(self, on_bucket, before_bucket=None):
        """The backward launch list WITHOUT launching (works on a dry engine): calls before_bucket(i) / on_bucket(i) where
        run_backward calls on_bucket(i), i.e. behind the last op of gradient bucket i.  Returns [(op index, bucket)] in call order.
        For the CPU tests of the data-parallel protocol (tests/test_dist_cpu.py)."""
        marks = self._bucket_marks()
        fired = []
        for i, op in enumerate(self.bwd):
            if i in marks:
                if before_bucket is not None:
                    before_bucket(marks[i])
                on_bucket(marks[i])
                fired.append((i, marks[i]))
        return fired

    def _bucket_marks(self):
        """index of the last backward op of each gradient bucket (pnp | head | layer4 | layer3 | rest)."""
        if hasattr(self, "_marks"):
            return self._marks
        n_groups = len(self.bwd_groups)
        sizes = [len(g) for g in reversed(self.bwd_groups)]
        # groups in forward order: stem(1) + 16 blocks + convT(1) + 6 head convs + head out(1) + 3 pnp convs + fc(1)
        cum, ends = 0, []
        for s in sizes:
            cum += s
            ends.append(cum - 1)
        # reversed order: fc, pnp x3 | head-out, head convs x6, convT | layer4 (3) | layer3 (6) | layer2 (4), layer1 (3), stem
        bounds = [n_groups - g0 for g0 in self.e.bucket_first_group]
        self._marks = {ends[b - 1]: i for i, b in enumerate(bounds)}
        return self._marks
