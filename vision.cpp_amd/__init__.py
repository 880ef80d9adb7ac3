"""MI355X-native backend for the visp:: Depth-Anything hot path.

The directory name contains a dot, so import it through `__graft_entry__.load_package()`
(registers it as module `visioncpp_amd`). Contents:
  csrc/      HIP kernels + C++ host code + the C ABI (built into lib/libvisioncpp.so)
  _lib.py    ctypes declarations of the C ABI (mirror of the reference's
             bindings/python/visioncpp/_lib.py)
  vision.py  Device / Model classes with the reference's Python API plus the batched extension
  gguf.py    GGUF v3 writer/reader (numpy)
  synth.py   synthetic checkpoints / inputs
"""
__all__ = ["gguf", "synth"]
