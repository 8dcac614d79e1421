"""Optimiser and weight repack (TrainEngine mixin): per gradient bucket ONE launch updates the fp32 masters (torch.optim.SGD
semantics with the reference's parameter groups, solver/build.py:8-26), writes the forward-form packed weights and zeroes the
consumed gradients; a second launch writes the data-gradient form.  Runs on the update stream behind the bucket's exchange."""
import os

import torch

from . import streams

from . import ops
from .ops import PackedConv


class Update(object):
    def _build_pack_tables(self):
        """Allocate ONE flat packed buffer per form (forward, data gradient) and, per gradient bucket, the table that lets a
        single launch repack all of the bucket's convs from the flat fp32 masters (osd_pack_multi)."""
        import numpy as np
        mult = 64 if self.dtype == torch.bfloat16 else 16
        tr = [c for c in self.convs.values() if c.trainable]
        scale_off, scales = {}, []
        off = 0
        for c in tr:
            if c.bn_scale is not None:
                scale_off[c.name] = off
                scales.append(c.bn_scale)
                off += c.cout
        self._flat_scale = torch.cat(scales) if scales else torch.zeros(1, device=self.device)
        base = self.flat_w.data_ptr()
        self._pack = {}
        for form in (0, 1):
            entries, dst_off = [], 0
            for c in tr:
                if form == 0:
                    rows, kpad = ops._round_up(c.cout, 16), ops._round_up(c.cin, mult)
                else:
                    rows, kpad = ops._round_up(c.cin, 16), ops._round_up(c.cout, mult)
                numel = rows * c.r * c.s * kpad
                nb = max(1, min(64, (numel + 256 * 16 - 1) // (256 * 16)))
                entries.append(dict(c=c, src=(c.w.data_ptr() - base) // 4, dst=dst_off, scale=scale_off.get(c.name, -1), rows=rows,
                                    kpad=kpad, nb=nb, numel=numel))
                dst_off += (numel + 63) // 64 * 64
            flat = torch.zeros(dst_off, device=self.device, dtype=self.dtype)
            tables = {}
            for bucket in self.exchange.ranges:
                sub = [e for e in entries if self._bucket_of(e["c"].w) == bucket]
                if not sub:
                    continue
                # 3 int64 offsets + 8 int32 (cout, cin, r, s, rows, kpad, first_block, n_blocks) = 7 x 8 bytes per entry
                tab = np.zeros((len(sub), 7), dtype=np.int64)
                blocks = []
                for i, e in enumerate(sub):
                    c = e["c"]
                    tab[i, 0:3] = (e["src"], e["dst"], e["scale"])
                    tab[i, 3:7] = np.frombuffer(np.array([c.cout, c.cin, c.r, c.s, e["rows"], e["kpad"], len(blocks), e["nb"]],
                                                         dtype=np.int32).tobytes(), dtype=np.int64)
                    blocks += [i] * e["nb"]
                tables[bucket] = dict(table=torch.from_numpy(tab).to(self.device),
                                      blocks=torch.tensor(blocks, dtype=torch.int32, device=self.device), n=len(blocks))
            self._pack[form] = dict(flat=flat, tables=tables)
            if form == 0:
                self._pack_fwd_entry = {id(e["c"]): e for e in entries}      # conv -> its forward-form entry (the fused update)
            for e in entries:
                c = e["c"]
                view = flat[e["dst"]:e["dst"] + e["numel"]].view(e["rows"], c.r, c.s, e["kpad"])
                if form == 0:
                    cout_store = ops._round_up(c.cout, 4)
                    if c.has_bias and c.cout % 16 == 0:
                        bias = c.b                                   # the fp32 master bias IS the epilogue's bias vector
                    else:
                        bias = torch.zeros(ops._round_up(cout_store, 16), device=self.device, dtype=torch.float32)
                        if c.bn_shift is not None:
                            bias[:c.cout] = c.bn_shift
                    c.pc = PackedConv(view, bias, c.cout, cout_store, e["rows"], e["kpad"], c.r, c.s, cin_real=c.cin)
                else:
                    zb = torch.zeros(ops._round_up(c.cin, 16), device=self.device, dtype=torch.float32)
                    c.pd = PackedConv(view, zb, c.cin, c.cin, e["rows"], e["kpad"], c.r, c.s, cin_real=c.cout)
        self._padded_bias = {}
        for c in tr:                            # the two prediction convs keep a padded copy of their 2 / 4 biases
            if c.has_bias and c.cout % 16 != 0:
                self._padded_bias.setdefault(self._bucket_of(c.w), []).append(c)

    def repack(self, buckets=None, forms=(0, 1)):
        """fp32 masters -> kernel-layout weights of the compute dtype: per gradient bucket two launches (forward and
        data-gradient forms).  buckets=None: all of them; forms=(1,): the data-gradient form only (the fused update has
        already written the forward form)."""
        for c in self.convs.values():
            if not c.trainable and c.pc is None:
                c.pc = ops.pack_conv(self._frozen_sd[c.name + ".weight"], bn=None if c.bn_scale is None else tuple(
                    self._frozen_sd[c.name.replace("conv", "bn").replace("downsample.0", "downsample.1") + k]
                    for k in (".weight", ".bias", ".running_mean", ".running_var")), dtype=self.dtype,
                    stem=c.name.endswith("stem.conv1"))
        if not hasattr(self, "_pack"):
            self._build_pack_tables()
        for bucket in (self.exchange.ranges if buckets is None else buckets):
            for form in forms:
                pk = self._pack[form]
                tb = pk["tables"].get(bucket)
                if tb is not None:
                    ops._lib.call("osd_pack_multi", ops._ptr(tb["table"]), ops._ptr(tb["blocks"]), tb["n"], ops._ptr(self.flat_w),
                                  ops._ptr(self._flat_scale), ops._ptr(pk["flat"]), form, ops._dt(pk["flat"]), ops._stream())
            for c in self._padded_bias.get(bucket, ()):
                c.pc.bias[:c.cout] = c.b

    def _build_sgd_table(self, weights, biases):
        """Per gradient bucket the table of osd_sgd_momentum_pack_multi (one launch updates every tensor of the bucket AND writes
        the forward-form packed weights of its conv tensors; OSD_NO_FUSED_REPACK=1: osd_sgd_momentum_multi + the two-form repack)."""
        import numpy as np
        base = self.flat_w.data_ptr()
        conv_of = {c.w.data_ptr(): c for c in self.convs.values() if c.trainable}
        rows = {name: [] for name in self.exchange.ranges}
        for group, lr_mult, wd in ((weights, 1.0, self.weight_decay), (biases, 2.0, 0.0)):
            for t in group:
                rows[self._bucket_of(t)].append(((t.data_ptr() - base) // 4, t.numel(), lr_mult, wd, conv_of.get(t.data_ptr())))
        fuse = os.environ.get("OSD_NO_FUSED_REPACK", "0") == "0"
        tables = {}
        for name, rs in rows.items():
            if not rs:
                continue
            fused = fuse
            tab = np.zeros((len(rs), 8 if fused else 4), dtype=np.int64)         # 64 / 32 bytes per entry
            blocks = []
            for i, (off, n, lm, wd, c) in enumerate(rs):
                nb = max(1, min(64, (n + 256 * 16 - 1) // (256 * 16)))
                tab[i, 0], tab[i, 1] = off, n
                tab[i, 2] = np.frombuffer(np.array([lm, wd], dtype=np.float32).tobytes(), dtype=np.int64)[0]
                tab[i, 3] = np.frombuffer(np.array([len(blocks), nb], dtype=np.int32).tobytes(), dtype=np.int64)[0]
                if fused:
                    tab[i, 4] = tab[i, 5] = -1
                    if c is not None:
                        e = self._pack_fwd_entry[id(c)]
                        # the kernel trusts this layout (backward.hip: SgdPackEntry): a change in _build_pack_tables must fail here,
                        # not write out of range there
                        cout = n // (c.cin * c.r * c.s)
                        assert e["kpad"] >= c.cin and cout * c.cin * c.r * c.s == n, (name, i, e["kpad"], c.cin, n)
                        assert 0 <= e["dst"] and e["dst"] + cout * c.r * c.s * e["kpad"] <= self._pack[0]["flat"].numel(), (name, i)
                        tab[i, 4], tab[i, 5] = e["dst"], e["scale"]
                        tab[i, 6:8] = np.frombuffer(np.array([c.cin, c.r * c.s, e["kpad"], 0], dtype=np.int32).tobytes(), dtype=np.int64)
                blocks += [i] * nb
            tables[name] = dict(table=torch.from_numpy(tab).to(self.device), fused=fused,
                                blocks=torch.tensor(blocks, dtype=torch.int32, device=self.device), n=len(blocks))
        self._sgd = dict(tables=tables, buf=torch.zeros_like(self.flat_w), steps=0)

    def _update_bucket(self, name):
        """SGD(momentum) on the bucket's masters, then its repack, on the current stream."""
        sg = self._sgd
        tb = sg["tables"].get(name)
        fused = tb is not None and tb["fused"]
        if fused:
            pk = self._pack[0]["flat"]
            ops._lib.call("osd_sgd_momentum_pack_multi", ops._ptr(tb["table"]), ops._ptr(tb["blocks"]), tb["n"],
                          ops._ptr(self.flat_w), ops._ptr(self.flat_g), ops._ptr(sg["buf"]), ops._ptr(self._flat_scale), ops._ptr(pk),
                          ops._dt(pk), float(self.lr), float(self.momentum), int(sg["steps"] == 0), int(self.consume_grads), ops._stream())
            if self.consume_grads:
                self._zeroed.add(name)
        elif tb is not None:
            ops._lib.call("osd_sgd_momentum_multi", ops._ptr(tb["table"]), ops._ptr(tb["blocks"]), tb["n"],
                          ops._ptr(self.flat_w), ops._ptr(self.flat_g), ops._ptr(sg["buf"]), float(self.lr),
                          float(self.momentum), int(sg["steps"] == 0), ops._stream())
        self.repack([name], forms=(1,) if fused else (0, 1))      # (the padded copies of the 2 / 4 prediction biases ride along)
        self._updated.add(name)

    def optimizer_step(self):
        """Apply the update to every bucket train_step has not already updated behind the backward pass."""
        if self.opt is not None:
            self.opt.step()
            self.repack()
            return
        if self._overlap:              # (never inside a captured graph: capture() turns the overlap off)
            main = streams.current()
            main.wait_stream(self.ustream)
            if self.exchange.comm is not None:
                main.wait_stream(self.exchange.comm)
        for name in self.exchange.ranges:
            if name not in self._updated:
                self._update_bucket(name)
        self._end_of_update()
        self._sgd["steps"] += 1

    def _end_of_update(self):
        self._updated = set()
        self._grads_clean = bool(self.consume_grads and self._zeroed >= set(self._sgd["tables"]))      # every bucket consumed
        self._zeroed = set()
