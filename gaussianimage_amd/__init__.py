"""MI355X-native accumulated-summation 2D Gaussian rasterizer (HIP, gfx950)."""
__version__ = "0.1.0"
