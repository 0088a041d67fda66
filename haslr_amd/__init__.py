"""haslr_amd — MI355X-native backbone + consensus stage of HASLR's haslr_assemble.

  haslr_amd.host  host pipeline (ingest, serial graph cleaning, stitching, writers)  -> libhaslr_host.so
  haslr_amd.hip   the C-ABI of include/haslr_hip.h (HIP kernels for gfx950)          -> libhaslr_hip.so
  haslr_amd/bin/haslr_assemble   drop-in CLI (same flags as the reference's binary)
"""
__version__ = "0.1.0"
