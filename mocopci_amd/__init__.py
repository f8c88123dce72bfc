"""mocopci_amd -- MI355X-native point-set hot path of icdm-adteam/MoCoPCI.

Submodules:
  _lib             ctypes binding of libmocopci_hip.so (C ABI: include/mocopci_hip.h)
  pointnet2_cuda   the reference extension's nine entry points (pointnet2/src/pointnet2_api.cpp)
  pointnet2_utils  the reference's autograd operator API (pointnet2/pointnet2_utils.py)
  ops              fused channel-last layer operators (knn, group_rows, interp3, chamfer, ...)
  model            the MoCoPCI inference graph on those operators (state-dict compatible)
  synth            synthetic inputs + deterministic by-name weights
"""
__version__ = "0.1.0"
