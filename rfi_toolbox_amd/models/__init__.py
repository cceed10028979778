"""Drop-in for ``rfi_toolbox.models`` (reference: rfi_toolbox/models/__init__.py:12-14 exports
``UNet``; ``UNetBigger`` is models/unet.py:79-118) running on MI355X through librfi_hip.so."""
from .backbone import ResNet50FPN, resnet50_fpn_entries
from .box_head import BoxHead
from .mask_head import MaskHead, mask_head_entries
from .mask_rcnn import MaskRCNN
from .resnet_unet import UNetResNet18, resnet_unet_entries
from .rpn_head import RPNHead, rpn_head_entries
from .simple_cnn import SimpleCNN, simple_cnn_entries
from .unet import (HipSegmenter, UNet, UNetBigger, UNetDifferentActivation, UNetOverfit, default_init_state,
                   unet_entries)

__all__ = ["UNet", "UNetBigger", "UNetOverfit", "UNetDifferentActivation", "SimpleCNN", "UNetResNet18", "resnet_unet_entries", "MaskHead", "mask_head_entries", "MaskRCNN", "ResNet50FPN", "resnet50_fpn_entries", "BoxHead", "RPNHead", "rpn_head_entries", "HipSegmenter", "default_init_state", "unet_entries",
           "simple_cnn_entries"]
