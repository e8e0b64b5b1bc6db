"""Host-side helpers with the reference's names: model_utils (Backbone, DoubleConv, concat helper),
pipeline_utils (build_model), loss_utils (calc_loss), ckpt (checkpoint wire format)."""
