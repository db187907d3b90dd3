"""MI355X-native rational Bloom-filter residual coder (host surface; HIP library loaded on first use)."""
__version__ = "0.1.0"
