"""pointcept_amd -- MI355X (gfx950) engine for Pointcept's SparseUNet voxel convolution and PTv3
serialized-attention hot paths.  Host code is Python/PyTorch-ROCm; all hot-path compute goes
through the C-ABI of libptcore.so (include/ptcore.h), hand-written HIP for CDNA4.
"""
__version__ = "0.1.0"
