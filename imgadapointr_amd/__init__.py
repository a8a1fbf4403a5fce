"""MI355X-native AdaPoinTr completion hot path.

Hand-written HIP kernels for gfx950 (imgadapointr_amd/csrc -> libptops.so, C ABI in
include/ptops.h) behind the reference's own operator / registry / runner interface.
"""
__version__ = "0.1.0"
