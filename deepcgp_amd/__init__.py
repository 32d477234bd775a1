"""deepcgp_amd -- MI355X-native (gfx950) conv-GP forward / ELBO hot path of DeepCGP.

Host code is plain Python + NumPy calling hand-written HIP kernels through the C-ABI declared in
``include/dcgp.h`` (``deepcgp_amd/csrc`` -> ``libdcgp.so``) via ``ctypes``.  There is no CPU fallback:
anything that computes raises ``deepcgp_amd.device.LibraryMissing`` when the library is absent.
"""
__version__ = "0.1.0"
