"""Shim package: ``from realesrgan import RealESRGANer`` (nesr/nesr.py:162,
standalone/direct_esrgan.py:93) resolves to the MI355X-native wrapper."""
from neural_enhanced_super_resolution_amd.realesrganer import RealESRGANer  # noqa: F401
__version__ = "0.0-nesr-hip-shim"
