"""Host-side mirror of reference vision_mtl/models/cross_stitch_model.py on the HIP kernels.

CrossStitchLayer keeps the reference's parameter shapes ((T,T,C) or (T,T), U(0,1) init) and its
exact arithmetic: the einsum "aac,abcij->abcij" only ever touches the DIAGONAL w[a,a,(c)], i.e.
a per-task (per-channel) scale with zero gradient for off-diagonal entries.

CSNet follows the reference's forward literally: walk every named module of the task networks in
registration order, apply only leaves, save / pad-concat skips at encoder / decoder block entries,
nearest-x2 at the last decoder block, stitch at block entries.  Each leaf type is dispatched to the
corresponding HIP kernel; the walk is compiled once into a flat program.
"""
from __future__ import annotations

import contextlib
import os
import re
import typing as t

import torch
from torch import nn

from .. import layers as L
from .. import ops
from ..utils.model_utils import get_module_by_name


def get_joint_layer_names_before_stitch_for_unet(joint_layer_names: t.List[str]) -> t.List[str]:
    """reference utils/model_utils.py:100-115: encoder stage entries `*.encoder.*` with 5 name parts and
    index != 0, and decoder block entries (4 name parts)."""
    out = []
    for name in joint_layer_names:
        parts = name.split(".")
        if ("encoder" in parts and len(parts) == 5 and int(parts[-1]) != 0) or ("decoder" in parts and len(parts) == 4):
            out.append(name)
    return out


class CrossStitchLayer(nn.Module):
    """reference models/cross_stitch_model.py:15-37."""

    def __init__(self, num_tasks: int, num_channels: t.Optional[int] = None):
        super().__init__()
        self.num_tasks = num_tasks
        self.channel_wise_stitching = num_channels is not None
        shape = (num_tasks, num_tasks, num_channels) if self.channel_wise_stitching else (num_tasks, num_tasks)
        self.weights = nn.Parameter(torch.Tensor(*shape))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.uniform_(self.weights)

    def run(self, acts: t.List[L.Act]) -> t.List[L.Act]:
        return [self.run_task(i, a) for i, a in enumerate(acts)]

    def run_task(self, i: int, a: L.Act) -> L.Act:
        return L.Act(ops.stitch(a.t, self.weights, i, a.C), a.C)

    def forward(self, mt_activations: torch.Tensor) -> torch.Tensor:
        """(T, B, C, H, W) -> (T, B, C, H, W), as the reference's einsum."""
        outs = self.run([L.from_nchw(mt_activations[i]) for i in range(self.num_tasks)])
        return torch.stack([L.to_nchw(o) for o in outs], dim=0)


_ACT_OF = {nn.ReLU: ops.ACT_RELU, nn.Hardswish: ops.ACT_HSWISH, nn.Hardsigmoid: ops.ACT_HSIGMOID,
           nn.Sigmoid: ops.ACT_SIGMOID}


class CSNet(nn.Module):
    """reference models/cross_stitch_model.py:40-201."""

    def __init__(self, models: dict, channel_wise_stitching: bool = False):
        super().__init__()
        self.encoder_block_regex = r"0.encoder.model.blocks.(\d+)$"
        self.decoder_block_regex = r"0.decoder.blocks.(\d+)$"
        self.num_tasks = len(models)
        self.model_names = list(models.keys())
        self.models = nn.ModuleDict(models)
        first = self.models[self.model_names[0]]
        self.joint_layer_names = [n for n, _ in list(first.named_modules())[1:]]
        self.joint_layer_names_before_stitch = get_joint_layer_names_before_stitch_for_unet(self.joint_layer_names)
        self.num_encoder_layers = len(list(get_module_by_name(first, "0.encoder.model.blocks").named_children()))
        self.num_decoder_layers = len(list(get_module_by_name(first, "0.decoder.blocks").named_children()))
        self.valid_cross_stitch_layer_names = [n.replace(".", "_") for n in self.joint_layer_names_before_stitch]
        self.true_cross_stitch_layer_names = list(self.joint_layer_names_before_stitch)
        if channel_wise_stitching:
            self.stitch_channels = self.get_stitch_channels(first, self.joint_layer_names_before_stitch)
            layers = {n: CrossStitchLayer(self.num_tasks, self.stitch_channels[i])
                      for i, n in enumerate(self.valid_cross_stitch_layer_names)}
        else:
            layers = {n: CrossStitchLayer(self.num_tasks) for n in self.valid_cross_stitch_layer_names}
        self.cross_stitch_layers = nn.ModuleDict(layers)
        self._program = None
        self.debug_acts = None  # set to a list to capture (op, arg, task, NCHW tensor) after every op (diagnostics)

    # ---- reference :159-201
    def consider_encoder_layer_at_idx(self, layer_idx: int) -> bool:
        return layer_idx not in (0, self.num_encoder_layers - 1, self.num_decoder_layers - 1)

    def consider_decoder_layer_at_idx(self, layer_idx: int) -> bool:
        return layer_idx != self.num_decoder_layers - 1

    def get_stitch_channels(self, random_model: nn.Module, names: t.List[str]) -> t.List[int]:
        mods = list(random_model.named_modules())[1:]
        index = {n: i for i, (n, _) in enumerate(mods)}
        stitch_channels, encoder_channels = [], []
        for name in names:
            j = index[name] - 1
            while not isinstance(mods[j][1], nn.Conv2d):  # last conv in front of the stitch site
                j -= 1
            ch = mods[j][1].out_channels
            if "encoder" in name:
                if self.consider_encoder_layer_at_idx(int(re.match(self.encoder_block_regex, name).group(1))):
                    encoder_channels.append(ch)
            if "decoder" in name:
                idx = int(re.match(self.decoder_block_regex, name).group(1))
                if self.consider_decoder_layer_at_idx(idx):
                    ch += encoder_channels[-idx - 1]
            stitch_channels.append(ch)
        return stitch_channels

    # ---- the walk of reference :102-157, compiled once
    def _compile(self):
        prog = []
        stitch_sites = set(self.joint_layer_names_before_stitch)
        first = self.models[self.model_names[0]]
        for name in self.joint_layer_names:
            layer = get_module_by_name(first, name)
            m = re.match(self.encoder_block_regex, name)
            if m and self.consider_encoder_layer_at_idx(int(m.group(1))):
                prog.append(("save", None))
            m = re.match(self.decoder_block_regex, name)
            if m:
                idx = int(m.group(1))
                prog.append(("merge", idx) if self.consider_decoder_layer_at_idx(idx) else ("up", None))
            if not any(True for _ in layer.named_children()) and not isinstance(layer, nn.Identity):
                prog.append(("leaf", name))
            if name in stitch_sites:
                prog.append(("stitch", name.replace(".", "_")))
        # peephole: conv -> plain BatchNorm2d -> ReLU leaves (decoder Conv2dReLU) become one fused call
        fused, i = [], 0
        while i < len(prog):
            if (i + 2 < len(prog) and all(p[0] == "leaf" for p in prog[i:i + 3])):
                a, b, c = (get_module_by_name(first, p[1]) for p in prog[i:i + 3])
                if (isinstance(a, nn.Conv2d) and a.groups == 1 and type(b) is nn.BatchNorm2d and isinstance(c, nn.ReLU)):
                    fused.append(("conv_bn_relu", (prog[i][1], prog[i + 1][1])))
                    i += 3
                    continue
            fused.append(prog[i])
            i += 1
        # peephole 2: a stitch site directly in front of a dense conv (every site of the reference's walk is: the next
        # leaf is a block's conv_pw / conv, or a decoder block's conv1) is FOLDED into that conv - the scale rides on the
        # packed weights, its gradient comes out of the conv's weight-gradient slabs (ops._Conv2d): no stitch launch at all
        folded, i = [], 0
        fold_ok = os.environ.get("VMTL_STITCH_FOLD", "1") != "0"
        while i < len(fused):
            op, arg = fused[i]
            if fold_ok and op == "stitch" and i + 1 < len(fused):
                nop, narg = fused[i + 1]
                conv_name = narg if nop == "leaf" else narg[0] if nop == "conv_bn_relu" else None
                layer = get_module_by_name(first, conv_name) if conv_name else None
                if isinstance(layer, nn.Conv2d) and layer.groups == 1:
                    folded.append(("st_" + nop, (arg, narg)))
                    i += 2
                    continue
            folded.append((op, arg))
            i += 1
        self._program = folded

    @staticmethod
    def _apply_leaf(layer: nn.Module, x: L.Act) -> L.Act:
        if isinstance(layer, nn.Conv2d):
            return L.dwconv(x, layer) if layer.groups != 1 else L.conv(x, layer)
        if type(layer) is nn.BatchNorm2d:
            return L.bn_act(x, layer, ops.ACT_NONE)
        for cls, code in _ACT_OF.items():
            if isinstance(layer, cls):
                return L.activation(x, code)
        raise NotImplementedError(f"CSNet: no HIP kernel registered for leaf module {type(layer).__name__}")

    def forward(self, x: torch.Tensor) -> dict:
        if self._program is None:
            self._compile()
        # task streams only with a gradient arena (or no gradients at all): torch's AccumulateGrad nodes would otherwise
        # run on a stream other than the one they were created on (extra syncs, a warning, trouble under capture)
        task_par = (x.is_cuda and ops.side.enabled and ops.side.task_parallel and self.num_tasks == 2
                    and self.debug_acts is None
                    and (not torch.is_grad_enabled()
                         or getattr(next(self.parameters()), "_vmtl_gslot", None) is not None))
        ops.packs.refresh(task_mode=task_par)  # one batched weight-packing launch for the whole step
        x0 = L.from_nchw(x)
        feats = {task: x0 for task in self.model_names}
        skips = {task: [] for task in self.model_names}
        # Two task networks that never exchange data (the stitch only scales a task's own features): the
        # second one runs on its own stream - a parallel branch of the captured graph - and is joined at the end.
        main = torch.cuda.current_stream() if x.is_cuda else None
        streams = {task: None for task in self.model_names}
        if task_par:
            s1 = ops.side.task_stream(x.device)
            s1.wait_stream(main)
            x0.t.record_stream(s1)
            streams[self.model_names[1]] = s1
        for op, arg in self._program:
            for ti, task in enumerate(self.model_names):
                with (torch.cuda.stream(streams[task]) if streams[task] is not None else contextlib.nullcontext()):
                    net, f = self.models[task], feats[task]
                    if op == "stitch":
                        feats[task] = self.cross_stitch_layers[arg].run_task(ti, f)
                    elif op == "save":  # two consumers (the next leaf and a decoder merge): gradients summed by our kernel
                        feats[task], keep = L.fork(f)
                        skips[task].append(keep)
                    elif op == "merge":
                        feats[task] = L.pad_cat(f, skips[task][-arg - 1])
                    elif op == "up":
                        feats[task] = L.up2_cat(f, None)
                    elif op == "leaf":
                        feats[task] = self._apply_leaf(get_module_by_name(net, arg), f)
                    elif op == "st_leaf":  # stitch scale folded into the conv that follows it
                        feats[task] = L.conv(f, get_module_by_name(net, arg[1]), stitch=(self.cross_stitch_layers[arg[0]].weights, ti))
                    elif op == "st_conv_bn_relu":
                        c_, b_ = arg[1]
                        feats[task] = L.conv_bn_act(f, get_module_by_name(net, c_), get_module_by_name(net, b_), ops.ACT_RELU,
                                                    stitch=(self.cross_stitch_layers[arg[0]].weights, ti))
                    else:  # conv_bn_relu
                        feats[task] = L.conv_bn_act(f, get_module_by_name(net, arg[0]), get_module_by_name(net, arg[1]),
                                                    ops.ACT_RELU)
                if self.debug_acts is not None and op in ("merge", "up", "conv_bn_relu", "st_conv_bn_relu"):
                    rec = [op, arg, task, L.to_nchw(feats[task]).detach().cpu(), None]
                    if feats[task].t.requires_grad:
                        C = feats[task].C
                        feats[task].t.register_hook(
                            lambda g, rec=rec, C=C: rec.__setitem__(4, g[..., :C].permute(0, 3, 1, 2).detach().cpu()))
                    self.debug_acts.append(rec)
        out = {}
        for task in self.model_names:
            if streams[task] is not None:
                with torch.cuda.stream(streams[task]):
                    out[task] = L.to_nchw(feats[task])
                main.wait_stream(streams[task])
                out[task].record_stream(main)
            else:
                out[task] = L.to_nchw(feats[task])
        return out
