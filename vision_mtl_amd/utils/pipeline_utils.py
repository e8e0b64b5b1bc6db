"""Host-side mirror of the construction API, reference vision_mtl/utils/pipeline_utils.py:22-30,80-136."""
from __future__ import annotations

import argparse
import typing as t

import torch

from ..lit_module import MTLModule
from ..models.basic_model import BasicMTLModel
from ..models.cross_stitch_model import CSNet
from ..models.mtan_model import MTANMiniUnet
from ..models.unet_mobilenetv3 import get_model_with_dense_preds


class DataConfig(t.Protocol):
    num_classes: int


def build_model(args: argparse.Namespace, data_cfg: DataConfig) -> t.Union[BasicMTLModel, MTANMiniUnet, CSNet]:
    """reference utils/pipeline_utils.py:80-136: dispatch on args.model_name in {basic, mtan, csnet}."""
    encoder_weights = getattr(args, "backbone_weights", "imagenet")
    if args.model_name == "basic":
        return BasicMTLModel(segm_classes=data_cfg.num_classes, decoder_first_channel=540, num_decoder_layers=5,
                             encoder_weights=encoder_weights)
    if args.model_name == "mtan":
        return MTANMiniUnet(in_channels=3, map_tasks_to_num_channels={"depth": 1, "segm": data_cfg.num_classes},
                            task_subnets_hidden_channels=128, encoder_first_channel=32, encoder_num_channels=4)
    if args.model_name == "csnet":
        backbone_params = dict(encoder_name="timm-mobilenetv3_large_100", encoder_weights=encoder_weights,
                               decoder_first_channel=256, num_decoder_layers=5)
        models = {
            "depth": get_model_with_dense_preds(segm_classes=1, activation=None, backbone_params=backbone_params),
            "segm": get_model_with_dense_preds(segm_classes=data_cfg.num_classes, activation=None,
                                               backbone_params=backbone_params),
        }
        return CSNet(models, channel_wise_stitching=getattr(args, "channel_wise_stitching", True))
    raise NotImplementedError(f"Unknown model name: {args.model_name}")


def init_model(args: argparse.Namespace, data_cfg: DataConfig) -> MTLModule:
    """reference utils/pipeline_utils.py:22-30."""
    model = build_model(args, data_cfg)
    module = MTLModule(model=model, num_classes=data_cfg.num_classes, lr=getattr(args, "lr", None),
                       device=getattr(args, "device", "cuda"))
    if getattr(args, "ckpt_dir", None):
        from .ckpt import load_ckpt_model

        module.load_state_dict(load_ckpt_model(args.ckpt_dir)["model"])
    return module


def fetch_data_cfg(dataset_name: str):
    """reference utils/pipeline_utils.py:288-294 with the constants of cfg.py:63-155."""
    if dataset_name == "cityscapes":
        return argparse.Namespace(num_classes=19, height=128, width=256, batch_size=8)
    if dataset_name == "nyuv2":
        return argparse.Namespace(num_classes=14, height=256, width=256, batch_size=4)
    raise ValueError(f"Unknown dataset name: {dataset_name}")
