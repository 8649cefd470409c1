"""vtaco_amd -- MI355X (gfx950) implementation of VTacO's occupancy hot path.

Host side mirrors the reference's module interface (same registry names,
constructor kwargs, state_dict keys); the arithmetic runs in hand-written HIP
kernels behind the C ABI of libvtaco_hip.so (include/vtaco_hip.h).
"""
__version__ = "0.1.0"
