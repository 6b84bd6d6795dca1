"""MI355X-native Spotforming localization-by-separation search (hot path only).

Sub-modules are imported lazily by their users; importing the package itself
touches neither torch nor the HIP library.
"""
__all__ = ["config", "weights", "native", "ops", "spot", "sep", "patch", "search", "hostdsp", "dense_grid",
           "mic_array", "srp", "joint", "scenes", "shard", "batching", "evalkit", "experiment", "flops"]
__version__ = "0.3.0"
