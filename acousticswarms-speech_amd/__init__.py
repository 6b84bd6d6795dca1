"""MI355X-native Spotforming localization-by-separation search (hot path only).

Sub-modules are imported lazily by their users; importing the package itself
touches neither torch nor the HIP library.
"""
__all__ = ["config", "weights", "native", "spot", "patch", "geometry", "search",
           "mic_array", "srp", "joint", "scenes", "shard"]
__version__ = "0.1.0"
