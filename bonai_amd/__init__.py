"""bonai_amd -- MI355X-native LOFT/FOA detector hot path."""
__version__ = '0.1.0'
