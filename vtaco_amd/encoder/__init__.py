"""Encoder registry (same names as reference src/encoder/__init__.py:11-20)."""
from . import pointnet
from ..layers import TactileUNet

encoder_dict = {
    'pointnet_local_pool': pointnet.LocalPoolPointnet,
    'UNet': TactileUNet,
}
