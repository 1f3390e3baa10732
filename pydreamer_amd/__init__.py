"""pydreamer_amd — MI355X-native DreamerV2 gradient step (hand-written HIP behind pydreamer's module API)."""
__version__ = '0.1.0'
