"""flatnav_amd -- MI355X-native batched k-NN search for flat navigable-small-world indexes.

Drop-in for the search path of BlaiseMuhirwa/flatnav: the device side is the C ABI of
include/flatnav_hip.h (hand-written gfx950 HIP kernels, libflatnav_hip.so); `flatnav_amd.hip`
binds it with ctypes.
"""
__version__ = "0.1.0"
