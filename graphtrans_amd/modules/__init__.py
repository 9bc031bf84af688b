"""Mirror of /root/reference/modules for the hot path (same class names, ctor/forward signatures
and state_dict keys), computing through the HIP kernels in graphtrans_amd/csrc."""
