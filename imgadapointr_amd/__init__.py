"""MI355X-native AdaPoinTr completion hot path.

Hand-written HIP kernels for gfx950 (imgadapointr_amd/csrc -> libptops.so, C ABI in
include/ptops.h) behind the reference's own operator / registry / runner interface.
"""
__version__ = "0.1.0"

import os as _os

# HIP graphs on this stack (ROCm 7.2 / PyTorch 2.10+rocm7.0): with the runtime's AQL-packet capture of graph nodes on (its
# default), a MEMSET node stops taking effect on replays that follow a host synchronisation preceded by other work
# (benchmarks/graph_memset_env.py -> profiles/round4/graph_memset_env_v1.log: explicit hipMemsetAsync and ATen's multi-block
# reductions replay stale results; with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 every flow is correct).  A whole train step of
# the image / segmentation models contains memsets this build does not issue (ATen reductions, rocPRIM, MIOpen), so the
# flag is set here - the runtime reads it at its first call, which is why this package should be imported before the first
# GPU call.  utils/graph_runner.py probes the behaviour and refuses to replay graphs where memset nodes are unreliable.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
