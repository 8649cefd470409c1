"""Encoder registry (same names as reference src/encoder/__init__.py:11-20)."""
from . import pointnet
from ..layers import Resnet18, Resnet34, TactileUNet

encoder_dict = {
    'pointnet_local_pool': pointnet.LocalPoolPointnet,
    'UNet': TactileUNet,
    'Resnet18': Resnet18,
    'Resnet34': Resnet34,
}
