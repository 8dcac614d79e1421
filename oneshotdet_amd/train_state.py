"""Reference-named state (TrainEngine mixin): the flat fp32 master / gradient / momentum buffers <-> the reference's state_dict
names and OIHW shapes (utils/checkpoint.py:35-52), including the tensors this build fuses or splits."""
import math

import torch

from . import spec


class State(object):
    def _codecs(self):
        """master tensor name -> (reference keys, import(list of reference tensors) -> master-shaped tensor, export(master-
        shaped tensor) -> {reference key or key#part: tensor}).  Conv weights live in [cout][r][s][cin] order; the fused /
        split tensors of this build are assembled from / taken apart into the reference's entries here, in ONE place, for
        state_dict(), named_grads() and the optimizer state alike."""
        h, b = "rpn.head.", "roi_heads.box."
        c, mid, p = spec.FPN_OUT, spec.FPN_OUT // 2, spec.BOX_POOL
        nc = spec.BOX_NUM_CLASSES
        out = {}
        for name, shape in self._plan:
            base, leaf = name.rsplit(".", 1)
            if name == h + "scales":
                keys = ["%sscales.%d.scale" % (h, i) for i in range(5)]
                out[name] = (keys, lambda ts: torch.cat([t.reshape(1) for t in ts]),
                             lambda v, keys=keys: {k: v[i:i + 1] for i, k in enumerate(keys)})
            elif base == h + "cls_ctr":
                keys = [h + "cls_logits." + leaf, h + "centerness." + leaf]
                if leaf == "weight":
                    out[name] = (keys, lambda ts: torch.cat(ts, 0).permute(0, 2, 3, 1),
                                 lambda v, keys=keys: {keys[0]: v[0:1].permute(0, 3, 1, 2), keys[1]: v[1:2].permute(0, 3, 1, 2)})
                else:
                    out[name] = (keys, lambda ts: torch.cat(ts, 0), lambda v, keys=keys: {keys[0]: v[0:1], keys[1]: v[1:2]})
            elif base in (b + "compress_dim_conv.0x", b + "compress_dim_conv.0q"):
                key = b + "compress_dim_conv.0." + leaf
                if leaf == "bias":
                    out[name] = ([key], lambda ts: ts[0], lambda v, key=key: {key: v})
                else:
                    lo = 0 if base.endswith("0x") else c
                    part = "#0" if base.endswith("0x") else "#1"          # merged along the input channels on export
                    out[name] = ([key], lambda ts, lo=lo: ts[0][:, lo:lo + c].permute(0, 2, 3, 1),
                                 lambda v, key=key, part=part: {key + part: v.permute(0, 3, 1, 2)})
            elif base == b + "fc6" and leaf == "weight":
                # Linear over x.view(N, -1) of NCHW maps (box_head.py:151): columns (c, h, w) -> this build's (h, w, c)
                out[name] = ([name], lambda ts: ts[0].view(-1, mid, p, p).permute(0, 2, 3, 1).reshape(-1, 1, 1, mid * p * p),
                             lambda v, name=name: {name: v.reshape(-1, p, p, mid).permute(0, 3, 1, 2).reshape(-1, mid * p * p)})
            elif base == b + "fc7" and leaf == "weight":
                out[name] = ([name], lambda ts: ts[0][:, None, None, :], lambda v, name=name: {name: v.reshape(v.shape[0], -1)})
            elif base == b + "pred":
                keys = [b + "predictor.cls_score." + leaf, b + "predictor.bbox_pred." + leaf]
                if leaf == "weight":
                    out[name] = (keys, lambda ts: torch.cat(ts, 0)[:, None, None, :],
                                 lambda v, keys=keys: {keys[0]: v[:nc].reshape(nc, -1), keys[1]: v[nc:].reshape(4 * nc, -1)})
                else:
                    out[name] = (keys, lambda ts: torch.cat(ts, 0), lambda v, keys=keys: {keys[0]: v[:nc], keys[1]: v[nc:]})
            elif base in self.convs and leaf == "weight":
                out[name] = ([name], lambda ts: ts[0].permute(0, 2, 3, 1), lambda v, name=name: {name: v.permute(0, 3, 1, 2)})
            else:                                       # conv bias, GroupNorm affine
                out[name] = ([name], lambda ts: ts[0], lambda v, name=name: {name: v})
        return out

    def _import_flat(self, flat, ref):
        """Fill a buffer laid out like the masters from reference-named tensors (weights, momentum, ...)."""
        off = 0
        for name, shape in self._plan:
            n = int(math.prod(shape))
            keys, imp, _ = self._codec[name]
            flat[off:off + n].view(shape).copy_(imp([torch.as_tensor(ref[k]).to(self.device, torch.float32) for k in keys]))
            off += n

    def _export_flat(self, flat):
        """Reference-named, reference-shaped copies of a buffer laid out like the masters."""
        out, parts, off = {}, {}, 0
        for name, shape in self._plan:
            n = int(math.prod(shape))
            _, _, exp = self._codec[name]
            for k, v in exp(flat[off:off + n].view(shape)).items():
                v = v.clone(memory_format=torch.contiguous_format)
                if "#" in k:
                    parts.setdefault(k.split("#")[0], {})[int(k.split("#")[1])] = v
                else:
                    out[k] = v
            off += n
        for k, ps in parts.items():
            out[k] = torch.cat([ps[i] for i in sorted(ps)], 1)
        return out

    def state_dict(self):
        """The reference's state_dict (same names, OIHW shapes) with the current fp32 master weights: what
        `DetectronCheckpointer.save` (utils/checkpoint.py:35-52) would write for the hot-path modules.  Frozen tensors
        (stem, layer1, every FrozenBN buffer) are returned unchanged."""
        self.join()
        # a checkpoint is where a training loop synchronises anyway: weights updated from a one-pass GroupNorm launch whose hand-off
        # timed out (NaN outputs + error word, groupnorm_onepass.hip) must not be written out as if they were valid
        from . import ops
        ops.gn_onepass_check("TrainEngine.state_dict")
        out = {k: v.clone() for k, v in self._frozen_sd.items()}
        out.update(self._export_flat(self.flat_w))
        return out

    def named_grads(self):
        """Reference-named gradients (OIHW) for parity tests."""
        return self._export_flat(self.flat_g)

    def optimizer_state_dict(self):
        """What the reference checkpoint stores under 'optimizer' (utils/checkpoint.py:42-46: torch.optim.SGD.state_dict()),
        keyed by reference parameter NAME instead of torch's positional ids: momentum buffers (OIHW, reference names), the
        number of steps taken (the first step initialises the buffer with the gradient, torch.optim.SGD semantics) and
        the hyper-parameters.  load_optimizer_state_dict() restores it, so a resumed run continues with its momentum."""
        self.join()
        if self.opt is not None:
            raise NotImplementedError("optimizer='torch' (A/B mode): use self.opt.state_dict()")
        return {"momentum_buffer": self._export_flat(self._sgd["buf"]), "steps": int(self._sgd["steps"]),
                "lr": float(self.lr), "momentum": float(self.momentum), "weight_decay": float(self.weight_decay)}

    def load_optimizer_state_dict(self, state):
        self.join()
        if self.opt is not None:
            raise NotImplementedError("optimizer='torch' (A/B mode): use self.opt.load_state_dict()")
        if float(state.get("weight_decay", self.weight_decay)) != self.weight_decay:
            raise ValueError("weight decay is baked into the update tables: construct the engine with weight_decay=%r"
                             % state["weight_decay"])
        self._import_flat(self._sgd["buf"], state["momentum_buffer"])
        self._sgd["steps"] = int(state["steps"])
        self.lr = float(state.get("lr", self.lr))
        self.momentum = float(state.get("momentum", self.momentum))
