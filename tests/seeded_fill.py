"""Seeded parameter fill shared by the golden generators (run against the real reference in the build container) and the
tests (run against vtaco_amd's mirror modules): two modules with the same state_dict keys and shapes get the same numbers,
so a fixture of a multi-million-parameter network stores inputs and outputs only.  Test infrastructure."""
import torch


def seeded_fill(module, seed, gain=0.6):
    """Overwrite every parameter / buffer in state_dict order: matrices and conv kernels ~ N(0, gain^2 / fan_in) (activations
    a little above the default init's 0.577: logits stay O(1) like a freshly initialised model's), norm scales 1 + 0.1 r, running variances 0.5 + |r|, every other vector 0.1 r."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if "num_batches_tracked" in name:
                continue
            r = torch.randn(t.shape, generator=g)
            if name.endswith("running_var"):
                t.copy_(0.5 + r.abs())
            elif t.dim() >= 2:
                t.copy_(r * gain * (1.0 / max(1, t[0].numel())) ** 0.5)
            elif name.endswith("norm.weight") or name.endswith("groupnorm.weight") or ".bn" in name and name.endswith(".weight"):
                t.copy_(1.0 + 0.1 * r)
            else:
                t.copy_(0.1 * r)
    return module


def keys_of(module):
    return [f"{k}:{tuple(v.shape)}" for k, v in module.state_dict().items() if "num_batches_tracked" not in k]
