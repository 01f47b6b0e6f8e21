from neural_enhanced_super_resolution_amd.rrdbnet import RRDBNet  # noqa: F401
