"""Every buffer the sweeps touch -- owned by the engine and STATIC -- and the launches that fill them per batch /
per parameter point: split-K slab buffers, [t_x | x] / [W | v_W] operands, (I,H,W,O) weight copies, targets.

Mixin of ``FusedGGNEngine`` (engine/core.py); see that class for the sweeps' overall structure."""

import os

import torch
from torch import nn

from .. import _lib
from .common import _Unsupported, _flat_view, _live_taps, _pair, _ptr


class _Buffers:
    ADJ_BYTES_PER_WG = 32768  # the BatchNorm adjoint's row-major kernel: one workgroup per this many bytes of the map
    # ---- buffers -------------------------------------------------------------------------
    def _plan(self, direction, u, forward=False):
        n, c, h, w = u.x.shape
        k, _, r, s = u.conv.weight.shape
        cin = 2 * c if (direction == 0 and not forward) else c
        sp = _lib.load().hf_conv2d_nhwc_plan(direction, n, h, w, cin, k, r, s, u.conv.stride[0], u.conv.stride[1],
                                             u.conv.padding[0], u.conv.padding[1],
                                             0)
        if sp < 1:
            raise _Unsupported(f"{u.name}: convolution geometry refused ({sp})")
        return sp

    def _allocate(self):
        """Every buffer the sweeps touch is owned by the engine and STATIC: the activations are
        recomputed in place by ``forward_own`` (own kernels), so one engine -- and the hipGraphs
        captured over it -- can serve every Newton step and every trial point of a step."""
        dev, f32 = self.dev, torch.float32
        cl = torch.channels_last

        def nhwc(shape):
            return torch.empty(shape, dtype=f32, device=dev).contiguous(memory_format=cl)

        self._nhwc = nhwc
        self.x_in = nhwc(self._in_shape)
        if self.pool_args is not None:
            self.pool_out = nhwc(self.pool_out.shape)
        for u in self.units:
            u.a, u.y = nhwc(u.a.shape), nhwc(u.y.shape)
            if u.train:
                u.rstd_version = None  # (u.rstd: batch statistics -- recorded by the model's pass or the own one's)
            elif u.bn is not None:
                u.rstd = torch.rsqrt(u.bn.running_var + u.bn.eps)
                u.rstd_version = u.bn.running_var._version
        tails = {id(c[-1]): c[0] for c, _, _ in self.blocks}
        for u in self.units:  # the convolution's input IS its producer's output buffer
            u.x = self.x_in if u.src == "input" else self.pool_out if u.src == "pool" else u.src.y
            if u.res_unit is not None:
                u.res = u.res_unit.y
            elif u.res_identity:
                u.res = tails[id(u)].x  # the block input
            else:
                u.res = None
        xcats = {}
        self._tangent_slots = {}
        for u in self.units:
            n, c, h, w = u.x.shape
            k, _, r, s = u.conv.weight.shape
            if u.im2col:
                if c * r * s > 256:
                    raise _Unsupported(f"{u.name}: too many taps for the im2col formulation")
                oh, ow = u.a.shape[2], u.a.shape[3]
                j = c * r * s
                u.cols = torch.empty((n, oh * ow, j), dtype=f32, device=dev)  # [N, OH*OW, c*r*s]
                u.geo = (n * oh * ow, 1, 1, j, k, 1, 1, (1, 1), (0, 0))
                # the weight gradient reads the im2col with rows padded to 16-byte multiples (zero
                # channels): 16-byte gathers instead of element-wise ones (28 -> 11 us for the 49-tap stem)
                jp = -(-j // 4) * 4
                u.cols_pad = torch.zeros((n, oh * ow, jp), dtype=f32, device=dev)
                u.geo_w = (n * oh * ow, 1, 1, jp, k, 1, 1, (1, 1), (0, 0))
                u.jcols = j
                if not u.conv.weight.is_contiguous():
                    raise _Unsupported(f"{u.name}: weight layout")
            else:
                if c % 4 or k % 4:
                    raise _Unsupported(f"{u.name}: channel counts must be multiples of 4")
                u.geo = (n, h, w, c, k, r, s, tuple(u.conv.stride), tuple(u.conv.padding))
                key = id(u.x)
                if key not in xcats:
                    xcats[key] = torch.zeros((n, 2 * c, h, w), dtype=f32, device=dev).contiguous(memory_format=cl)
                u.xcat = xcats[key]
                # [W | v_W]; slices of taps that never meet data stay 0
                u.wcat = torch.zeros((k, 2 * c, r, s), dtype=f32, device=dev).contiguous(memory_format=cl)
                u.wT = torch.empty((c, r, s, k), dtype=f32, device=dev)  # (I, H, W, O)
                u.live = _live_taps(h, w, r, s, u.conv.stride, u.conv.padding)
                if u.pw is not None:  # (a frozen weight: its W half is copied from the parameter, its v_W half stays 0)
                    self._tangent_slots[id(u)] = (self._offs[u.pw], u.wcat, c, u.live)
            oh, ow = u.a.shape[2], u.a.shape[3]
            u.rows, u.cout = n * oh * ow, k
            # split-K slab buffers
            if u.im2col:
                u.sT = u.sF = self._plan_stem(0, u.geo)
                u.sW = self._plan_stem(2, u.geo_w)
                u.sD = 0
            else:
                u.sT, u.sW = self._plan(0, u), self._plan(2, u)
                u.sD = 0 if (u.first or u.no_dgrad or u.dead) else self._plan(1, u)
                u.sF = self._plan(0, u, forward=True)
            u.tbuf = torch.empty((max(u.sT, u.sF), u.rows * k), dtype=f32, device=dev)
            # Hessian products add, per layer, conv_W(t_x, g) to the weight gradient and conv_D(g, V) to the
            # data gradient (g: the step's first-order cotangent): as MORE SLABS of the same buffers, which
            # the consumers sum anyway
            u.nW = u.sW * (2 if (self.hessian and not u.first) else 1)
            u.nD = u.sD * (2 if self.hessian else 1)
            u.wbuf = torch.zeros((u.nW, u.conv.weight.numel()), dtype=f32, device=dev)  # dead taps stay 0
            if u.sD:
                u.dbuf = torch.empty((u.nD, u.x.numel()), dtype=f32, device=dev)
            if self.hessian:
                u.ga1 = torch.empty_like(u.a)  # first-order cotangent of the convolution output (per step)
                if u.bn is not None:
                    u.g1 = torch.empty_like(u.a)   # ... and of the BatchNorm output (masked), per step
                    u.gah = torch.empty_like(u.a)  # g_a' + g_z * rstd * v_gamma: what the convolutions' adjoints read
                if u.sD:
                    # V as (I, H, W, O), per product (a frozen weight has V = 0: zeros, never scattered into)
                    u.vT = torch.zeros((c, r, s, k), dtype=f32, device=dev)
            u.g = torch.empty_like(u.a) if u.needs_g else None  # masked cotangent of the unit's output
            u.ga = torch.empty_like(u.a)   # cotangent of the convolution output
            # the BatchNorm adjoint shares the rows among `rb` workgroups per channel column; the
            # per-channel sums arrive as rb partial rows that hf_pack_ex adds up
            u.rb = 1
            if k % 4 == 0 and k // 4 <= 256 and u.rows >= 64:
                # row-major adjoint kernel: ~64 workgroups, each reading whole contiguous rows, one
                # pass of the row loop where the map is small enough (measured on the ResNet-18
                # bench: 32 workgroups x 2 passes 1124, 64 x 1 1150, 128 x 1 the same, 256 x 1 1138)
                rp = 256 // (k // 4)
                # (64 workgroups suit the <= 1.6 MB maps of ResNet-18; a 12.6 MB map of All-CNN-C needs the
                # whole chip: one workgroup per 32 KB of the map, 64 ... 1024)
                tgt = min(1024, max(64, u.a.numel() * 4 // self.ADJ_BYTES_PER_WG))
                per = max(rp, -(-u.rows // tgt))
                u.rb = -(-u.rows // per)
                if u.rb < 2:
                    u.rb = 1
            # (Hessian: the scale's second-order term arrives as `rb` more partial rows for hf_pack_ex to add)
            u.gw_rows = u.rb * (2 if (self.hessian and u.bn is not None) else 1)
            if self.hessian and u.train:
                u.gw_rows += 1  # (+ the closed-form share of the scale's second-order term: hf_bn_train_hessian_coeffs)
            u.gw = torch.empty((u.gw_rows, k), dtype=f32, device=dev)
            u.gb = torch.empty((u.rb, k), dtype=f32, device=dev)
            if u.train:
                # The per-channel finalisation of a train-mode tangent / adjoint runs in the PROLOGUE of the elementwise
                # pass: its workgroups add the reduction's partial rows up themselves (hf_chan_affine_train).  The
                # forms of round 4 that handed over inside a launch or took a launch of their own were measured slower
                # (profiles/r04_train_bn_forms.jsonl) and are gone.
                if not (k % 4 == 0 and k // 4 <= 256):
                    raise _Unsupported(f"{u.name}: train-mode BatchNorm over {k} channels (the prologue form takes "
                                       "multiples of 4 up to 1024)")
                u.stat_part = torch.empty((u.rb, 2, k), dtype=torch.float64, device=dev)  # one-pass statistics
                # ... and the tangent's partial sums by the convolution's own epilogue (64x64-tile launches; one row per
                # (row tile, split): beyond 256 rows the separate reduction's `rb` rows are cheaper
                # for the elementwise pass to add up)
                tp_rows = -(-u.rows // 64) * u.sT
                u.epi = (not u.im2col and not u.first and hasattr(u, "xcat")
                         and tp_rows <= 256
                         and os.environ.get("HF_BN_EPILOGUE", "1") != "0")
                if self.hessian:
                    # Hessian products (hf_bn_train_hessian_*): the tangent sweep's partial sums must outlive the
                    # adjoint's (which share gw / gb in a GGN product); six coefficient vectors; the layer's
                    # first-order parameter gradients (per step)
                    if u.bn.weight is None or u.bn.bias is None:  # (frozen scale / shift are fine: constants)
                        raise _Unsupported(f"{u.name}: Hessian products need an affine train-mode BatchNorm")
                    u.hx = torch.empty((u.rb, k), dtype=f32, device=dev)
                    u.h1 = torch.empty((u.rb, k), dtype=f32, device=dev)
                    u.hcoef = torch.empty((6, k), dtype=f32, device=dev)
                    u.gg1 = torch.zeros(k, dtype=f32, device=dev)
                    u.gb1 = torch.zeros(k, dtype=f32, device=dev)
                if u.epi:
                    u.tp1 = torch.empty((tp_rows, k), dtype=f32, device=dev)
                    u.tpx = torch.empty((tp_rows, k), dtype=f32, device=dev)
        # where each unit's output goes besides its own dense buffer: the [t_x | x] operand of its
        # consumer -- the tangent into the first half, the value (forward pass) into the second
        for u in self.units:
            xc = xcats.get(id(u.y))
            c = u.y.shape[1]
            if xc is not None:
                u.tout, u.tout_ld, u.yout2 = xc[:, :c], 2 * c, xc[:, c:]
            else:
                u.tout, u.tout_ld, u.yout2 = torch.empty_like(u.y), 0, None
        self._xcats = xcats
        self._slot_list = list(self._tangent_slots.values())
        self._carry_ok = True  # (the stem's launch carries the v_W scatter; False once the library refused it)
        # (I, H, W, O) copies: the weights (once per step) and, for Hessian products, V (per product)
        self._wt_slots = [(self._offs[u.pw], u.wT, u.x.shape[1]) for u in self.units
                          if not u.im2col and not u.first and u.pw is not None and u.sD]
        # frozen convolution weights: constants the flat parameter vector does not hold -- their W half / (I, H, W, O)
        # copy come from the parameter itself (`refresh_frozen`: at construction and whenever it was written to)
        self._frozen_w = [u for u in self.units if not u.im2col and u.pw is None]
        self._frozen_seen = {}
        self._vt_slots = [(self._offs[u.pw], u.vT, u.x.shape[1]) for u in self.units
                          if self.hessian and not u.im2col and u.sD and u.pw is not None]
        self._allocate_pool()
        self._allocate_head()
        # the parameters as ONE flat vector, when they are consecutive views of one (the optimizer's
        # arena): every W half is then refreshed by a single scatter launch
        self._flat_params = _flat_view(self.params, self.n)

    def _allocate_pool(self):
        self.pool_t = self._xcats.get(id(self.pool_out))
        if self.pool_t is None:
            raise _Unsupported("nothing consumes the pooled stem output")
        if _pair(self.pool_args[3]) != [1, 1] or self.pool_args[4]:
            raise _Unsupported("max-pool with dilation / ceil_mode")
        self.pool_idx32 = torch.empty(tuple(self.pool_out.permute(0, 2, 3, 1).shape), dtype=torch.int32,
                                      device=self.dev)
        self._g_stem = torch.empty_like(self.stem.y)

    def _allocate_head(self):
        """Classifier head: (global average pool ->) linear layer."""
        f32, dev = torch.float32, self.dev
        tail = self.tail
        n, k = tail.y.shape[0], tail.y.shape[1]
        self._head_hw = tail.y.shape[2] * tail.y.shape[3]
        if self._head_hw == 1:
            self.feat = tail.y.permute(0, 2, 3, 1).reshape(n, k)  # a view: NHWC with a 1x1 map is [n, k]
            if self.feat.data_ptr() != tail.y.data_ptr():
                raise _Unsupported("feature view")
        else:
            self.feat = torch.empty((n, k), dtype=f32, device=dev)
        self.logits = torch.empty((n, self.fc.weight.shape[0]), dtype=f32, device=dev)
        self._p = torch.empty_like(self.logits)
        self.loss_buf = torch.zeros((), dtype=f32, device=dev)

    def _plan_stem(self, direction, geo):
        n, h, w, c, k, r, s, st, pd = geo
        sp = _lib.load().hf_conv2d_nhwc_plan(direction, n, h, w, c, k, r, s, 1, 1, 0, 0,
                                             0)
        if sp < 1:
            raise _Unsupported(f"stem geometry refused ({sp})")
        return sp

    # ---- own forward pass ------------------------------------------------------------------
    def set_batch(self, x, targets=None):
        """A new input batch of the same shape (and its targets): static input, the stem's im2col."""
        if tuple(x.shape) != tuple(self.x_in.shape):
            raise RuntimeError("engine: input shape changed")
        self.x_in.copy_(x)
        s = self.units[0]
        if s.im2col:
            cols = torch.nn.functional.unfold(self.x_in.contiguous(), tuple(s.conv.kernel_size),
                                              padding=tuple(s.conv.padding), stride=tuple(s.conv.stride))
            s.cols.copy_(cols.transpose(1, 2))
            s.cols_pad[:, :, :s.jcols].copy_(s.cols)
        for u in self.units:  # eval-mode statistics are constants -- unless somebody retrained them
            if u.bn is not None and not u.train and u.bn.running_var._version != u.rstd_version:
                torch.rsqrt(u.bn.running_var + u.bn.eps, out=u.rstd)
                u.rstd_version = u.bn.running_var._version
        self.refresh_frozen()
        if targets is not None:
            self.set_targets(targets)

    def refresh_frozen(self):
        """W halves and (I, H, W, O) copies of FROZEN convolution weights (not part of the flat vector the scatter
        launches read): copied when the parameter was written to or re-bound since the last look (eager, once per
        batch: never inside a captured graph)."""
        for u in self._frozen_w:
            w = u.conv.weight
            seen = (w._version, w.data_ptr())
            if self._frozen_seen.get(id(u)) != seen:
                c = u.x.shape[1]
                u.wcat[:, :c].copy_(w.detach())
                if u.sD:
                    u.wT.copy_(w.detach().permute(1, 2, 3, 0))
                self._frozen_seen[id(u)] = seen

    def set_targets(self, targets):
        if targets.dtype.is_floating_point:  # a mean-squared-error loss: targets of the outputs' shape
            if getattr(self, "_targets", None) is None:
                self._targets = torch.empty_like(self.logits)
                self._onehot = None
                self.bad_targets = torch.zeros((), dtype=torch.bool, device=self.dev)
            self._targets.copy_(targets)
            return
        if getattr(self, "_targets", None) is None:
            self._targets = torch.empty_like(targets)
            self._onehot = torch.zeros_like(self.logits)
        self._targets.copy_(targets)
        k = self.logits.shape[1]
        # class indices outside [0, K) (an ``ignore_index``) are not covered by the closed forms: the
        # flag is read back by the caller together with the loss value (no extra sync)
        self.bad_targets = ((self._targets < 0) | (self._targets >= k)).any()
        self._onehot.zero_()
        self._onehot.scatter_(1, self._targets.clamp(0, k - 1).view(-1, 1), 1.0)

    def refresh_weights(self, transposed=False):
        """The W halves of all [W | v_W] operands from the CURRENT parameters (one scatter launch
        when the parameters are views of one flat vector); ``transposed``: also the (I, H, W, O)
        copies the data-gradient convolutions read (once per step; trial points need only W)."""
        flat = self._flat_params
        if flat is not None and flat.data_ptr() == self.params[0].data_ptr():
            _lib.unpack_tangent(flat, self._slot_list, half=0)
        else:
            for u in self.units:
                if not u.im2col and u.pw is not None:
                    c = u.x.shape[1]
                    u.wcat[:, :c].copy_(self.params[u.pw].detach())
        if transposed:
            if flat is not None and flat.data_ptr() == self.params[0].data_ptr() and self._wt_slots:
                _lib.unpack_tangent(flat, self._wt_slots, half=2)
            else:
                for u in self.units:
                    if not u.im2col and not u.first and u.pw is not None and u.sD:
                        u.wT.copy_(self.params[u.pw].detach().permute(1, 2, 3, 0))

    def _zeros(self, k):
        cache = self.__dict__.setdefault("_zeros_cache", {})
        if k not in cache:
            cache[k] = torch.zeros(k, dtype=torch.float32, device=self.dev)
        return cache[k]

    # ---- tangent sweep -------------------------------------------------------------------------
    def _pool_geometry(self):
        """(n, h, w, oh, ow, c) of the stem's max-pool (window maxima's positions: ``forward_own``)."""
        pn, _, ph, pw = self.stem.y.shape
        return pn, ph, pw, self.pool_out.shape[2], self.pool_out.shape[3], self.pool_out.shape[1]

    def _tgeo(self, u):
        n, h, w, c, k, r, s, st, pd = u.geo
        return (n, h, w, 2 * c, k, r, s, st, pd)

    # ---- kernels ---------------------------------------------------------------------------
    def _conv_slabs(self, direction, out, act, mat, geo, splits, act_ld=0, out_c=0, mat_ld=0):
        n, h, w, c, k, r, s, st, pd = geo
        _lib.check(_lib.load().hf_conv2d_nhwc_slabs(
            direction, _ptr(out), _ptr(act), _ptr(mat), n, h, w, c, k, r, s, st[0], st[1], pd[0], pd[1], act_ld,
            mat_ld, out_c, splits, out.shape[1] if out.dim() == 2 else 0, _lib.HF_F32,
            _lib.current_stream_ptr(self.dev)), "hf_conv2d_nhwc_slabs")

    def _conv_carrying_scatter(self, u, mat, splits, src, half):
        """The im2col'd first layer's convolution (its weight operand ``mat`` is a slice of a flat vector, no
        scattered operand is read) in ONE launch with the scatter of ``src`` into the v_W (``half=1``) / W
        (``half=0``) halves of every other layer's operand (``hf_conv2d_nhwc_slabs_unpack``): the scatter hides
        behind the latency-bound convolution.  False: not taken (switched off, no scatter to carry, or a
        geometry / tensor count the merged launch refuses) -- the caller issues the two launches."""
        if not (self._carry_ok and u.im2col and self._slot_list):
            return False
        n, h, w, c, k, r, s_, st, pd = u.geo
        rc = _lib.load().hf_conv2d_nhwc_slabs_unpack(
            _ptr(u.tbuf), _ptr(u.cols), _ptr(mat), n, h, w, c, k, r, s_, st[0], st[1], pd[0], pd[1], 0, 0, splits,
            u.tbuf.shape[1], _ptr(src), *_lib.unpack_table(src, self._slot_list, half=half), _lib.HF_F32,
            _lib.current_stream_ptr(self.dev))
        if rc == _lib.HF_ERR_ARG:
            self._carry_ok = False
            return False
        _lib.check(rc, "hf_conv2d_nhwc_slabs_unpack")
        return True
