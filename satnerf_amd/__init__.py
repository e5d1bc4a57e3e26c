"""satnerf_amd -- MI355X-native drop-in for Sat-NeRF's volumetric-rendering hot path."""
__version__ = "0.1.0"
