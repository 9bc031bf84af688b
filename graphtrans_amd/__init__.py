"""graphtrans_amd — MI355X-native GraphTrans forward/backward hot path.

HIP/CDNA4 kernels (graphtrans_amd/csrc, C-ABI in include/graphtrans_hip.h) behind the
reference's own nn.Module surface (graphtrans_amd.modules / graphtrans_amd.models mirror
/root/reference/modules and /root/reference/models for the hot path only).
"""
__version__ = "0.1.0"
