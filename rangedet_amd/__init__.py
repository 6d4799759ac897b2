"""rangedet_amd -- MI355X-native RangeDet inference hot path (hand-written HIP kernels behind the reference's
Python symbol/config surface).  See DESIGN.md."""
__version__ = "0.1.0"
