"""MI355X-native Real-ESRGAN (RRDBNet) inference path for NESR.

Drop-in objects for the two third-party classes the reference constructs
(nesr/nesr.py:161-162,216-229; standalone/direct_esrgan.py:92-93,104,118-127):

    from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer

or, with ``<repo>/dropin`` on PYTHONPATH, the reference's own import lines
(``from basicsr.archs.rrdbnet_arch import RRDBNet``; ``from realesrgan import RealESRGANer``)
resolve to these classes unchanged.  The network forward runs in hand-written HIP kernels
(libnesr_hip.so, C ABI in include/nesr_hip.h); there is no CPU fallback.
"""
from .rrdbnet import RRDBNet, conv3x3, rrdbnet_state_dict_spec  # noqa: F401
from .realesrganer import RealESRGANer  # noqa: F401

__all__ = ["RRDBNet", "RealESRGANer", "conv3x3", "rrdbnet_state_dict_spec"]
__version__ = "0.1.0"
