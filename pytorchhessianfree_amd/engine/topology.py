"""Layer topology of a prepared ResNet-family model, read off its module tree and the activations its patched
layers recorded (identity by storage); what the captured graphs bake in about the layers; the two-phase split.

Mixin of ``FusedGGNEngine`` (engine/core.py); see that class for the sweeps' overall structure."""

from torch import nn

from .common import _Unit, _Unsupported, _cl, _same


class _Topology:
    # ---- topology ---------------------------------------------------------------------
    def _param(self, p):
        """Index of a layer parameter in the optimizer's (trainable) list; ``None``: the layer has no such parameter
        or it is FROZEN (``requires_grad = False``: it is a constant of the product -- no tangent, no gradient entry;
        reference utils.py:31-32 skips it the same way)."""
        if p is None:
            return None
        i = self._index.get(id(p))
        if i is None:
            if p.requires_grad:
                raise _Unsupported("a trainable layer parameter is not among the optimizer's parameters")
            self.frozen_any = True
        return i

    def _layout(self, model):
        need = ("conv1", "bn1", "maxpool", "avgpool", "fc")
        if not all(isinstance(getattr(model, a, None), nn.Module) for a in need):
            raise _Unsupported("not a ResNet-family module tree")
        if hasattr(model, "layers") and isinstance(model.layers, nn.Sequential):
            blocks = list(model.layers)
        elif all(hasattr(model, f"layer{i}") for i in range(1, 5)):
            blocks = [b for i in range(1, 5) for b in getattr(model, f"layer{i}")]
        else:
            raise _Unsupported("no block list")
        x_in = getattr(self.outputs, "_hf_input", None)
        if x_in is None:
            raise _Unsupported("no recorded input")

        def io(m, n):
            rec = getattr(m, "_hf_io", None)
            if rec is None or len(rec) != n:
                raise _Unsupported(f"{type(m).__name__} has no record of this forward pass")
            return rec

        units = []
        group_is_set = self.group is not None
        self.frozen_any = False

        def make_unit(name, conv, bn, relu_expected):
            if type(conv) is not nn.Conv2d or conv.bias is not None or conv.groups != 1:
                raise _Unsupported(f"{name}: unsupported convolution")
            if not getattr(conv, "_hf_channels_last", False) or tuple(conv.dilation) != (1, 1):
                raise _Unsupported(f"{name}: needs prepare_model(channels_last=True)")
            if not isinstance(bn, nn.BatchNorm2d):
                raise _Unsupported(f"{name}: not a BatchNorm2d")
            u = _Unit(name, conv, bn)
            cx, cy = io(conv, 2)
            rec = getattr(bn, "_hf_io", None)
            if bn.training:
                # train mode: the layer ran on stock ops and recorded its batch statistics (modelprep)
                if rec is None or len(rec) != 6 or group_is_set:
                    raise _Unsupported(f"{name}: train-mode BatchNorm without a record (or under data "
                                       "parallelism: batch statistics couple the samples of a shard)")
                bx, bres, by, brelu, rstd, mean_t = rec
                # (engine-owned static buffers: the own forward pass of a session rewrites them per batch)
                u.rec_stats = (mean_t, rstd)
                rstd, u.mean_t = rstd.clone(), mean_t.clone()
                u.train = True
            else:
                bx, bres, by, brelu, rstd = io(bn, 5)
            if not _same(cy, bx):
                raise _Unsupported(f"{name}: the BatchNorm does not consume the convolution's output")
            if brelu != relu_expected:
                raise _Unsupported(f"{name}: unexpected activation")
            # identity of activations: the RAW records (a 1-channel stem runs NCHW, its records are
            # converted below); the kernels get NHWC copies / views
            u.kx, u.ky, u.rx, u.ry, u.ra = cx.data_ptr(), by.data_ptr(), cx, by, cy
            u.x, u.a, u.y, u.relu, u.rstd, u.res = _cl(cx), _cl(cy), _cl(by), brelu, rstd, bres
            u.pw, u.pg, u.pb = self._param(conv.weight), self._param(bn.weight), self._param(bn.bias)
            units.append(u)
            return u

        # stem: conv1 -> bn1 (+ relu, fused by fuse_bn_relu or by a block-style forward) -> maxpool
        stem = make_unit("stem", model.conv1, model.bn1, True)
        if not _same(stem.rx, x_in.detach()) or stem.res is not None:
            raise _Unsupported("stem does not start at the network input")
        stem.src, stem.im2col, stem.first = "input", True, True
        self.model_ref, self._in_shape = model, tuple(x_in.shape)
        mp_x, mp_y = io(model.maxpool, 2)
        if not _same(mp_x, stem.ry):
            raise _Unsupported(f"maxpool does not follow the stem ({tuple(mp_x.shape)} {mp_x.stride()} "
                               f"{mp_x.data_ptr():x} vs {tuple(stem.y.shape)} {stem.y.stride()} {stem.y.data_ptr():x})")
        mp = model.maxpool
        self.stem, self.pool_args = stem, (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode)
        cur = mp_y
        self.pool_out, self.pool_key, self._rec_pool = _cl(mp_y), mp_y.data_ptr(), mp_y
        self.blocks = []
        prev = "pool"  # producer of the current block input
        for bi, b in enumerate(blocks):
            convs = [n for n in ("conv1", "conv2", "conv3") if isinstance(getattr(b, n, None), nn.Conv2d)]
            if not getattr(b, "_hf_block_patched", False) or len(convs) < 2:
                raise _Unsupported(f"block {bi}: not a fused residual block")
            chain, inp = [], cur
            for k, cn in enumerate(convs):
                u = make_unit(f"block{bi}.{cn}", getattr(b, cn), getattr(b, "bn" + cn[-1]), True)
                if not _same(u.rx, inp):
                    raise _Unsupported(f"block {bi}.{cn}: input is not the previous activation")
                last = k == len(convs) - 1
                if (u.res is not None) != last:
                    raise _Unsupported(f"block {bi}.{cn}: unexpected residual")
                u.src = chain[-1] if chain else prev
                chain.append(u)
                inp = u.ry
            tail = chain[-1]
            ds = None
            if b.downsample is not None:
                d = b.downsample
                if not (isinstance(d, nn.Sequential) and len(d) == 2):
                    raise _Unsupported(f"block {bi}: unsupported downsample")
                ds = make_unit(f"block{bi}.downsample", d[0], d[1], False)
                if not _same(ds.rx, cur) or ds.res is not None or not _same(tail.res, ds.ry):
                    raise _Unsupported(f"block {bi}: downsample wiring")
                tail.res_unit = ds
                ds.src = prev
            else:
                if not _same(tail.res, cur):
                    raise _Unsupported(f"block {bi}: identity wiring")
                tail.res_identity = True
            self.blocks.append((chain, ds, cur))
            cur = tail.ry
            prev = tail
        ap_x, ap_y = io(model.avgpool, 2)
        if not _same(ap_x, cur):
            raise _Unsupported("avgpool does not follow the last block")
        fc = model.fc
        fc_x, fc_y = io(fc, 2)
        if fc_x.dim() != 2 or fc_x.shape[0] != cur.shape[0] or fc_x.shape[1] != cur.shape[1]:
            raise _Unsupported("the classifier does not take the pooled features")
        if not _same(fc_y, self.outputs.detach()):
            raise _Unsupported("the network output is not the classifier's output")
        self.fc, self.feat = fc, fc_x
        self.pfw, self.pfb = self._param(fc.weight), self._param(fc.bias)
        if self.pfw is None or (fc.bias is not None and self.pfb is None):
            raise _Unsupported("the classifier layer is frozen (the engine covers frozen convolution / BatchNorm layers)")
        self.units = units
        self.tail = self.blocks[-1][0][-1]
        self.train_bn = any(u.train for u in units)
        used = {i for u in units for i in (u.pw, u.pg, u.pb) if i is not None} | {self.pfw} | (
            {self.pfb} if self.pfb is not None else set())
        if used != set(range(len(self.params))):
            raise _Unsupported("the parameter list has entries the engine's layers do not cover")
        self._mark_dead_prefix()

    def _mark_dead_prefix(self):
        """Frozen layers at the INPUT end of the net: a unit all of whose parameters are frozen and whose input carries
        no tangent produces no tangent either -- the tangent sweep starts at the first unit with a trainable parameter
        (its input tangent is zero: only ``conv(x, v_W)`` is left) and the adjoint sweep stops there (that unit needs
        its weight gradient, nobody needs its data gradient).  ``dead_blocks``: how many leading blocks are dead
        (only behind a dead stem); ``u.dead``; ``u.no_dgrad`` (a live unit that reads a tangent-free input)."""
        def frozen(u):
            return u.pw is None and u.pg is None and u.pb is None

        for u in self.units:
            u.dead = u.no_dgrad = False
        self.dead_blocks = 0
        self.stem.dead = frozen(self.stem)
        if not self.stem.dead:
            return
        for chain, ds, _x in self.blocks:
            if not all(frozen(u) for u in chain + ([ds] if ds is not None else [])):
                break
            for u in chain + ([ds] if ds is not None else []):
                u.dead = True
            self.dead_blocks += 1
        if self.dead_blocks == len(self.blocks):
            raise _Unsupported("every convolution / BatchNorm layer is frozen (only the classifier is trained)")
        chain, ds, _x = self.blocks[self.dead_blocks]
        chain[0].no_dgrad = True
        if ds is not None:
            ds.no_dgrad = True

    def layer_signature(self):
        """What the captured graphs of a session bake in about the model's layers besides shapes: module
        identities and every BatchNorm's mode / eps / momentum (kernel arguments).  Compared per step."""
        return tuple((id(u.conv), id(u.bn), None if u.bn is None else (u.bn.training, u.bn.eps, u.bn.momentum))
                     for u in self.units) + (bool(self.model_ref.training) if not self.units[0].bn is None else None,)

    # ---- the product in two phases, for overlapping the all-reduce with the rest of the sweep ----
    def phase_split(self, tail_fraction=0.7):
        """Block index ``cut`` such that the parameters of blocks ``cut ...`` and of the classifier are a
        contiguous SUFFIX of the flat vector holding at least ``tail_fraction`` of the entries that travel
        (the adjoint sweep finishes them first: ResNet-18 on 28x28 inputs, layer3 + layer4 + fc = 14 of
        17 MB after ~60 % of the product), with the flat offset of that suffix; ``None`` if the layout does
        not allow it."""
        if not self.blocks or self.fc is None:
            return None
        live = self._live_counts()
        total = sum(live)
        acc = live[self.pfw] + (live[self.pfb] if self.pfb is not None else 0)
        cut = None
        for bi in range(len(self.blocks) - 1, 0, -1):
            chain, ds, _ = self.blocks[bi]
            acc += sum(live[i] for u in chain + ([ds] if ds is not None else []) for i in (u.pw, u.pg, u.pb)
                       if i is not None)  # (frozen tensors: no entry)
            if acc >= tail_fraction * total:
                cut = bi
                break
        if cut is None or cut <= self.dead_blocks:
            return None  # (a frozen prefix that reaches the cut: nothing is left for the second phase to overlap with)
        late = {i for bi in range(cut, len(self.blocks)) for u in self.blocks[bi][0] + ([self.blocks[bi][1]] if self.blocks[bi][1] is not None else [])
                for i in (u.pw, u.pg, u.pb) if i is not None} | {self.pfw} | ({self.pfb} if self.pfb is not None else set())
        first = min(late)
        if late != set(range(first, len(self.params))):
            return None  # the late layers' parameters are not a suffix of the vector
        return cut, first, self._offs[first]

    def _live_counts(self):
        counts = [p.numel() for p in self.params]
        for u in self.units:
            if not u.im2col and getattr(u, "live", 0) and u.pw is not None:
                rs = u.conv.weight.shape[2] * u.conv.weight.shape[3]
                counts[u.pw] = u.conv.weight.numel() // rs * bin(u.live).count("1")
        return counts
