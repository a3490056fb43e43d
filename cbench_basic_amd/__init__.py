"""cbench_basic_amd -- MI355X-native encode/decode hot path behind cbench's plugin surface.

Sub-modules mirror the reference package layout for the hot path only:
  ans, rans                      <- cbench.ans / cbench.rans (csrc/ans, csrc/rans)
  utils.bytes_ops                <- cbench/utils/bytes_ops.py
  codecs, modules, nn            <- the codec / entropy-coder / prior-coder / transform classes
All compute goes through libbasic_hip.so (include/basic_hip.h); nothing falls back to CPU.
"""
__version__ = "0.1.0"
