"""CPU oracle for the Real-ESRGAN (RRDBNet) inference path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic of this path lives in two third-party packages that
are absent from /root/reference and from this image (``basicsr>=1.4.2``,
``realesrgan>=0.3.0``; reference requirements.txt:9-10, setup.py:37-38), and the
reference ships no tests, golden vectors or fixtures for it (SURVEY.md section 4,
section 8(c)).  The restatement below follows the published basicsr 1.4.2
``archs/rrdbnet_arch.py`` and realesrgan 0.3.0 ``utils.py`` algorithms and is anchored on
the reference's own call sites (nesr/nesr.py:216-229, 845-986;
standalone/direct_esrgan.py:104-148) and on structural cross-checks that the
reference does pin (parameter count vs. the 67,010,191-byte checkpoint size recorded
at nesr/utils/downloader.py:25).  The primitive ops (conv2d, leaky_relu, nearest
interpolate, pixel_unshuffle, reflect pad) are torch CPU ops, which are present here
and are the ground truth for those primitives.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker.  The product package
``neural_enhanced_super_resolution_amd`` never imports it.
"""
