"""MI355X-native batched Gumbel-MuZero self-play search (the hot path of enpasos/muzero).

The product is muzero_amd/libmuzero_hip.so (HIP kernels behind the C ABI of include/muzero_hip.h);
this package only holds its ctypes binding and the weight-blob description.
"""
from . import netspec  # noqa: F401
