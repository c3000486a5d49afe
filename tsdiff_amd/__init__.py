"""tsdiff_amd -- MI355X-native (gfx950) score-network / Langevin-sampling hot path of TSDiff.

Public surface mirrors the reference modules on the path:
    tsdiff_amd.epsnet.get_model            <- models/epsnet/__init__.py
    tsdiff_amd.epsnet.condensenc.CondenseEncoderEpsNetwork
    tsdiff_amd.sampler.EnsembleSampler     <- models/sampler.py
    tsdiff_amd.geometry.eq_transform       <- models/geometry.py
All compute runs in tsdiff_amd/libtsdiff_hip.so (C ABI: include/tsdiff_hip.h); there is no
CPU fallback -- importing is cheap, the first call fails loudly if the library is not built.
"""
__version__ = "0.1.0"
