"""sradsgan_amd -- MI355X-native implementation of SRADSGAN's generator/discriminator training step.

Layout: csrc/ (HIP kernels + C ABI, built into lib/libsradsgan_hip.so), _hip.py (ctypes binding),
ops.py (autograd wrappers), model/ (drop-in mirror of the reference's model/sradsgan.py surface).
"""
__version__ = '0.1.0'
