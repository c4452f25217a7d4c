"""Data parallelism (optimizer.py:677-684 across GPUs): only the entries of the product that can be non-zero travel;
the product in two phases so that the all-reduce of the late layers overlaps the rest of the sweep.

Mixin of ``FusedGGNEngine`` (engine/core.py); see that class for the sweeps' overall structure."""

import os

import torch

from .. import _lib
from ..curvature import _all_reduce_sum
from .common import _live_taps, _ptr


class _DataParallel:
    def local_phase_a(self, v, out, split):
        """Tangent sweep, head, adjoint sweep of blocks ``cut ...``, and the suffix of the product they
        determine (``out[offset:]``)."""
        cut, first, offset = split
        v = v.detach()
        self._tangent_stem(v, carry_scatter=True)
        self._tangent_blocks(v)
        g_last, g_fw, g_fb = self._head(v)
        self._phase_state = self._adjoint_blocks(g_last, last_block=cut)
        self._gather_range(out, g_fw, g_fb, first, len(self.params))

    def local_phase_b(self, out, split):
        """The rest of the adjoint sweep and ``out[:offset]``."""
        cut, first, offset = split
        pool_srcs = self._adjoint_blocks(None, first=cut - 1, last_block=0, incoming=self._phase_state)
        self._adjoint_stem(pool_srcs)
        self._gather_range(out, None, None, 0, first)

    # ---- data parallelism: only the entries that can be non-zero travel ----------------------
    def _live_segments(self):
        """Description of the product's entries that are not structurally zero -- the weight slices of
        kernel taps that never meet data are zero on every rank (``_live_taps``) --, or ``None`` when
        (almost) everything is live.  ResNet-18 on 28x28 inputs: 4.3 M of 11.2 M entries.

        Layout of what travels: the dense PREFIX of the vector (everything before the first tensor with dead
        taps: stem .. layer3, 2.8 M entries) is all-reduced IN PLACE in the full vector -- no copy at all;
        the rest (layer4's live taps, its BatchNorm vectors, the classifier: 1.5 M entries) is gathered into
        the compact staging vector by ``hf_live_copy``, all-reduced there and scattered back."""
        if not hasattr(self, "_live_segs"):
            self._live_segs = None
            masked = {u.pw: u for u in self.units if not u.im2col and getattr(u, "live", 0) and u.pw is not None}
            segs, dead, run_start = [], 0, None  # (full offset, count in the full vector, period, mask)
            brk = getattr(self, "_seg_break", None)  # parameter index at which a dense run must end
            self._seg_cut = None                     # (chunked all-reduce: the suffix starts a segment)
            for i, p in enumerate(self.params):
                off = self._offs[i]
                if i == brk:
                    if run_start is not None:
                        segs.append((run_start, off - run_start, 0, 0))
                        run_start = None
                    self._seg_cut = (len(segs), off - dead)  # (segment index, compact offset) of the suffix
                if i in masked:
                    if run_start is not None:
                        segs.append((run_start, off - run_start, 0, 0))
                        run_start = None
                    u = masked[i]
                    rs = p.shape[2] * p.shape[3]
                    segs.append((off, p.numel(), rs, u.live))
                    dead += p.numel() // rs * (rs - bin(u.live).count("1"))
                elif run_start is None:
                    run_start = off
            if run_start is not None:
                segs.append((run_start, self.n - run_start, 0, 0))
            # the in-place prefix: leading dense segments (at most two: a chunk break may cut the run), each
            # worth a collective of its own (>= 1 MB) and 16-byte aligned
            n_pre, prefix = 0, 0
            # (reduced in place: +59 -> +27 us per iteration on a 1-rank RCCL group, round 3)
            while (n_pre < min(2, len(segs) - 1) and segs[n_pre][2] == 0 and segs[n_pre][1] >= (1 << 18)
                   and (segs[n_pre][0] + segs[n_pre][1]) % 4 == 0):
                prefix = segs[n_pre][0] + segs[n_pre][1]
                n_pre += 1
            if masked and dead >= 0.2 * self.n and len(segs) - n_pre <= 24:
                self._prefix_runs = [(sg[0], sg[0] + sg[1]) for sg in segs[:n_pre]]
                if self._seg_cut is not None:
                    k, coff = self._seg_cut
                    # (cut in segment units of the STAGED list and staged offsets; k < n_pre: inside the prefix)
                    self._seg_cut = (k - n_pre, coff - prefix) if k >= n_pre else (k - n_pre, 0)
                segs = segs[n_pre:]
                arr = lambda col: (_lib.c_int64 * len(segs))(*[sg[col] for sg in segs])
                self._live_segs = (arr(0), arr(1), arr(2), arr(3), len(segs))
                self._n_live = self.n - dead
                self._compact = torch.empty(self.n - dead - prefix, dtype=torch.float32, device=self.dev)
        return self._live_segs

    def _reduce_pieces(self, full, part=None):
        """The tensors one product's all-reduce consists of: in-place slices of ``full`` (the dense prefix) and
        the compact staging vector; ``part``: "head" / "tail" of the chunked layout (``_seg_cut``)."""
        runs = [full[a:b] for a, b in self._prefix_runs]
        if part is None:
            pieces = runs + [self._compact]
        else:
            k, coff = self._seg_cut
            if k < 0:  # the cut lies inside the prefix: runs[:cut] are the head, everything else the tail
                cut = len(runs) + k
                pieces = runs[:cut] if part == "head" else runs[cut:] + [self._compact]
            else:
                pieces = runs + [self._compact[:coff]] if part == "head" else [self._compact[coff:]]
        return [t for t in pieces if t.numel() > 0]

    def _live_copy(self, full, scatter, part=None):
        """Gather (``scatter=False``) the staged live entries of ``full`` into the compact vector, or scatter
        them back; ``part``: "head" / "tail" of the chunked layout (segments before / from ``_seg_cut``)."""
        offs, counts, periods, masks, ns = self._live_segs
        lo, hi, comp = 0, ns, self._compact
        if part is not None:
            k, coff = self._seg_cut
            k = max(k, 0)
            lo, hi, comp = (0, k, self._compact[:coff]) if part == "head" else (k, ns, self._compact[coff:])
        if hi <= lo:
            return
        sub = lambda a: (_lib.c_int64 * (hi - lo))(*a[lo:hi])  # noqa: E731
        _lib.check(_lib.load().hf_live_copy(_ptr(full), _ptr(comp), int(scatter), sub(offs), sub(counts), sub(periods),
                                            sub(masks), hi - lo, _lib.HF_F32, _lib.current_stream_ptr(self.dev)),
                   "hf_live_copy")

    @property
    def reduce_bytes(self):
        return 4 * (self.n if self._live_segments() is None else self._n_live)

    def reduce(self, t, group=None):
        """Sum of the local products over the ranks.  The structurally-zero entries are zero on
        every rank, so only the live ones travel: the dense prefix in place, the rest through the compact
        staging vector (17 MB instead of 44.7 MB per product on the ResNet-18 workload)."""
        group = self.group if group is None else group
        if group is None:
            return t
        if (self._live_segments() is None or t.dtype != torch.float32 or t.numel() != self.n
                or not t.is_contiguous() or not t.is_cuda):
            return _all_reduce_sum(t, group)
        from ..distributed import all_reduce_sum_multi

        self._live_copy(t, False)
        all_reduce_sum_multi(self._reduce_pieces(t), group)  # (direct RCCL: one grouped launch)
        self._live_copy(t, True)
        return t
