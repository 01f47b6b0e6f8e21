"""Shim package: lets the reference's unmodified import line
``from basicsr.archs.rrdbnet_arch import RRDBNet`` (nesr/nesr.py:161,
standalone/direct_esrgan.py:92) and its ``importlib.util.find_spec("basicsr")`` probe
(nesr/nesr.py:153) resolve to the MI355X-native implementation.  Put ``<repo>/dropin`` and
``<repo>`` on PYTHONPATH; nothing of upstream basicsr is included."""
__version__ = "0.0-nesr-hip-shim"
