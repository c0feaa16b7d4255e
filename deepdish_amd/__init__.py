"""deepdish_amd -- MI355X-native detect -> encode -> track hot path of AdaptiveCity/deepdish.

Sub-packages mirror the reference's plugin surface: `deep_sort.*`, `tools.*`.
The arithmetic lives in libdeepdish_hip.so (include/deepdish_hip.h); build it with
`python -m deepdish_amd.build`.
"""
__version__ = '0.1.0'
