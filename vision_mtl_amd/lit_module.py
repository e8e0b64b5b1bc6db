"""Host-side mirror of reference vision_mtl/lit_module.py: the step surface run_pipe()/predict()
call (training_lit.py:81-98,115-168,186-216).  A plain nn.Module (the reference's LightningModule is
used without a Trainer), same method names, attributes and step_outputs layout.

Losses, post-processing and the per-step metrics are HIP kernels; nothing on the step path
synchronises with the host.
"""
from __future__ import annotations

import typing as t
from typing import Any

import torch
from torch import nn

from . import metrics as M
from . import ops
from .losses import CrossEntropyLoss, SILogLoss
from .utils.loss_utils import summarize_epoch_metrics


class MTLModule(nn.Module):
    def __init__(self, model: nn.Module, num_classes: int, optim_dict: t.Optional[dict] = None,
                 lr: t.Optional[float] = None, device: str = "cuda", loss_segm_weight: float = 1.0,
                 loss_depth_weight: float = 1.0):
        super().__init__()
        self.hparams = {"num_classes": num_classes, "optim_dict": optim_dict, "lr": lr, "device": device,
                        "loss_segm_weight": loss_segm_weight, "loss_depth_weight": loss_depth_weight}
        self.num_classes = num_classes
        self.model = model
        self.segm_criterion = CrossEntropyLoss()
        self.depth_criterion = SILogLoss()
        self.optim_dict = optim_dict
        self.loss_segm_weight = loss_segm_weight
        self.loss_depth_weight = loss_depth_weight
        self.step_outputs = {k: {"loss": [], "accuracy": [], "jaccard_index": [], "fbeta_score": [], "mae": []}
                             for k in ["train", "val", "test", "predict"]}
        # reference lit_module.py:48-69 (torchmetrics 0.7.3 Accuracy-micro / FBeta-weighted / Jaccard / MAE)
        self.metrics = {"accuracy": M.Accuracy(num_classes), "fbeta_score": M.FBetaScore(num_classes, beta=1.0),
                        "jaccard_index": M.JaccardIndex(num_classes), "mae": M.MeanAbsoluteError()}
        self.automatic_optimization = False
        self.compute_metrics = True  # bench.py turns this off to time exactly fwd + losses + bwd
        self._nan = {}  # device -> the NaN placeholder of the skipped metrics (built once, not filled every step)
        self.dp_arena = None  # a dp.FlatArena: training_step's loss then averages gradients over ranks at end of backward

    def forward(self, x: torch.Tensor) -> dict:
        return self.model(x)

    # ---- reference lit_module.py:75-95
    def shared_step(self, batch: dict, stage: str) -> torch.Tensor:
        img, gt_mask, gt_depth = batch["img"], batch["mask"], batch["depth"]
        raw_out = self(img)
        # the segmentation prediction comes out of the cross-entropy pass (calc_losses) when the criterion can emit it
        out = self.postprocess_raw_out(raw_out, defer_segm_predictions=hasattr(self.segm_criterion, "forward_with_predictions"))
        all_losses = self.calc_losses(gt_mask, gt_depth, out)
        all_metrics = self.calc_metrics(gt_mask, gt_depth, out)
        self.update_step_stats(stage, all_losses, all_metrics)
        if stage == "train" and self.dp_arena is not None and torch.is_grad_enabled():
            return self.dp_arena.sync_loss(all_losses["loss"])
        return all_losses["loss"]

    def update_step_stats(self, stage: str, all_losses: dict, all_metrics: dict) -> None:
        so = self.step_outputs[stage]
        so["loss"].append(all_losses["loss"].detach())
        for k in ("accuracy", "jaccard_index", "fbeta_score", "mae"):
            so[k].append(all_metrics[k])

    def calc_metrics(self, gt_mask, gt_depth, out: dict) -> dict:
        if not self.compute_metrics:
            nan = self._nan.get(gt_mask.device)
            if nan is None:
                nan = self._nan[gt_mask.device] = torch.full((), float("nan"), device=gt_mask.device)
            return {"accuracy": nan, "jaccard_index": nan, "fbeta_score": nan, "mae": nan}
        cm = M.confusion_matrix(out["segm_predictions"], gt_mask, self.num_classes)
        return {"accuracy": self.metrics["accuracy"].from_confusion(cm),
                "jaccard_index": self.metrics["jaccard_index"].from_confusion(cm),
                "fbeta_score": self.metrics["fbeta_score"].from_confusion(cm),
                "mae": self.metrics["mae"](out["depth_predictions"].detach(), gt_depth)}

    def calc_losses(self, gt_mask, gt_depth, out: dict) -> dict:  # reference lit_module.py:120-131
        if out.get("segm_predictions") is None and hasattr(self.segm_criterion, "forward_with_predictions"):
            loss_segm, out["segm_predictions"] = self.segm_criterion.forward_with_predictions(out["segm_logits"], gt_mask)
        else:
            loss_segm = self.segm_criterion(out["segm_logits"], gt_mask)
        loss_depth = self.depth_criterion(out["depth_predictions"], gt_depth)
        if loss_segm.is_cuda and loss_segm.dim() == 0 and loss_depth.dim() == 0:
            loss = ops.add_losses(loss_segm, loss_depth, self.loss_segm_weight, self.loss_depth_weight)
        else:
            loss = self.loss_segm_weight * loss_segm + self.loss_depth_weight * loss_depth
        return {"loss": loss, "loss_segm": loss_segm, "loss_depth": loss_depth}

    def postprocess_raw_out(self, out: dict, defer_segm_predictions: bool = False) -> dict:  # reference lit_module.py:133-144
        segm_logits, depth_logits = out["segm"], out["depth"]
        return {"segm_logits": segm_logits,
                # argmax(softmax(z)) == argmax(z); deferred: calc_losses fills it from the cross-entropy pass
                "segm_predictions": None if defer_segm_predictions else ops.argmax_channels(segm_logits),
                "depth_predictions": ops.sigmoid(depth_logits).permute(0, 2, 3, 1)}

    def training_step(self, batch: dict, batch_idx: Any = 0):
        return self.shared_step(batch=batch, stage="train")

    def validation_step(self, batch: dict, batch_idx: Any = 0):
        return self.shared_step(batch=batch, stage="val")

    def test_step(self, batch: dict, batch_idx: Any = 0):
        return self.shared_step(batch=batch, stage="test")

    def predict_step(self, batch: dict, batch_idx: int = 0, dataloader_idx: int = 0):
        out = self.postprocess_raw_out(self(batch["img"]))
        if "mask" in batch and "depth" in batch:
            gt_mask, gt_depth = batch["mask"], batch["depth"]
            self.update_step_stats("predict", self.calc_losses(gt_mask, gt_depth, out),
                                   self.calc_metrics(gt_mask, gt_depth, out))
        return {"segm": out["segm_predictions"], "depth": out["depth_predictions"]}

    def shared_epoch_end(self, stage: Any):
        return summarize_epoch_metrics(self.step_outputs[stage], metric_name_prefix=stage)

    def on_train_epoch_end(self):
        return self.shared_epoch_end("train")

    def on_validation_epoch_end(self):
        return self.shared_epoch_end("val")

    def on_test_epoch_end(self):
        return self.shared_epoch_end("test")

    def on_predict_epoch_end(self):
        return self.shared_epoch_end("predict")

    def configure_optimizers(self):  # reference lit_module.py:193-209 (unused by run_pipe)
        if self.optim_dict:
            return self.optim_dict
        optimizer = torch.optim.Adam(params=self.parameters(), lr=self.hparams["lr"])
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer=optimizer, patience=5, factor=0.95)
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": scheduler, "interval": "epoch", "monitor": "train_loss"}}

    def transfer_batch_to_device(self, batch: dict, device, dataloader_idx: int = 0):
        """reference lit_module.py:211-219.  Pinned host tensors (DataLoader(pin_memory=True), data.collate) go up
        asynchronously on the current stream; an image batch kept in dataset sample layout (B,H,W,3) is re-laid on
        the device by one kernel straight into the model's input storage (data.upload_batch)."""
        from .data import upload_batch

        if isinstance(batch, dict):
            up = upload_batch(batch, device)
            for key in batch.keys():
                batch[key] = up[key]
            return batch
        return batch.to(device, non_blocking=batch.device.type == "cpu" and batch.is_pinned())

    def parameters(self, recurse: bool = True):  # reference lit_module.py:232-234
        for p in self.model.parameters():
            yield p
