"""Model container (drop-in for reference src/conv_onet/models/__init__.py:15-197)."""
from __future__ import annotations

import os

import torch
from torch import distributions as dist
from torch import nn

from . import decoder

# same registry names as the reference (models/__init__.py:7-12); the crop / PointConv
# baselines are out of scope
# A/B knob: "0" keeps the reference's per-scene loop over the tactile feature encoder (one pass over all scenes' images otherwise)
_SCENE_BATCH = os.environ.get("VTACO_TACTILE_SCENE_BATCH", "1") != "0"

decoder_dict = {
    'simple_local': decoder.LocalDecoder,
    'attention_local': decoder.AttentionDecoder,
}


class ConvolutionalOccupancyNetwork(nn.Module):
    """Holds decoder / encoder / encoder_hand / encoder_img / encoder_t2d and exposes
    encode_* / decode* exactly as the reference does."""

    def __init__(self, decoder, encoder=None, encoder_hand=None, encoder_img=None, encoder_t2d=None, device=None):
        super().__init__()
        put = lambda m: None if m is None else m.to(device)
        self.decoder = put(decoder)
        self.encoder = put(encoder)
        self.encoder_hand = put(encoder_hand)
        self.encoder_img = put(encoder_img)
        self.encoder_t2d = put(encoder_t2d)
        self._device = device

    def forward(self, p, inputs, imgs=None, sample=True, **kwargs):
        return self.decode(p, self.encode_inputs(inputs), **kwargs)

    def _encode_with(self, enc, x):
        return enc(x) if enc is not None else torch.empty(x.size(0), 0)

    def encode_inputs(self, inputs):
        return self._encode_with(self.encoder, inputs)

    def encode_hand_inputs(self, inputs):
        return self._encode_with(self.encoder_hand, inputs)

    def encode_hand_mano(self, inputs):
        """MANO layer alone on given pose parameters (models/__init__.py:104-112)."""
        return self.encoder_hand.forward_mano(inputs)

    def encode_img_inputs(self, imgs):
        """Per-scene loop over the 5 tactile images (models/__init__.py:115-136): keeps
        train-mode BatchNorm statistics per scene, as the reference does."""
        if self.encoder_img is None:
            return torch.empty(imgs.size(0), 0)
        B, Fn = imgs.shape[:2]
        if B > 1 and _SCENE_BATCH and hasattr(self.encoder_img, "forward_scenes"):
            return self.encoder_img.forward_scenes(imgs)             # the same values from one pass over the B * Fn images
        return torch.cat([self.encoder_img(imgs[b]).reshape(1, Fn, -1) for b in range(B)], dim=0)

    def encode_t2d(self, inputs, imgs):
        return self.encoder_t2d.encode_img_inputs(imgs), self.encoder_t2d.encode_hand_inputs(inputs)

    # validate_args=False: the constructor's argument check reads the logits back to the host -- a device synchronisation
    # (8 ms of a training step) that the reference's callers, who only read .logits / .probs, have no use for
    def decode(self, p, c, **kwargs):
        return dist.Bernoulli(logits=self.decoder(p, c, **kwargs), validate_args=False)

    def decode_img(self, p, c, c_img=None, **kwargs):
        return dist.Bernoulli(logits=self.decoder.forward_img(p, c, c_img, **kwargs), validate_args=False)

    def decode_contact(self, p, c, **kwargs):
        logits, contact = self.decoder.forward_contact(p, c, **kwargs)
        return dist.Bernoulli(logits=logits, validate_args=False), contact

    def to(self, device):
        model = super().to(device)
        model._device = device
        return model
