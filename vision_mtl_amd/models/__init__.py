"""Model families of the reference's build_model(): basic (MobileNetV3-Large U-Net, two heads), csnet
(two U-Nets with cross-stitch scaling), mtan (attention mini U-Net).  Parameters live in torch modules
with the reference's state_dict keys; all arithmetic goes through vision_mtl_amd.ops."""
