"""pointvs_amd: MI355X-native (gfx950 HIP) implementation of the PointVS EGNN hot path.

The package mirrors the reference's class surface for this path
(`EGNNLayer`, `SartorrasEGNN`, `MultitaskSatorrasEGNN`, `PygLinearPass`, `PNNGeometricBase`,
`PointNeuralNetworkBase`) and routes every layer body through `libpvs_egnn.so`
(`include/pvs_egnn.h`). There is no CPU fallback: importing the kernels without the built
library, or running them on a non-HIP tensor, raises.
"""
__version__ = '0.1.0'
