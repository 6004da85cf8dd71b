"""semireward_amd: MI355X-native SemiReward hot path.  Compute = libsrhip.so only."""
__version__ = "0.1.0"
