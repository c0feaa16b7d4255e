"""MI355X-native counterparts of the reference's deep_sort package (same module and symbol names).

Every numeric routine here calls libdeepdish_hip.so; nothing falls back to numpy.
"""
