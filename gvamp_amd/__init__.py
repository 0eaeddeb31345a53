"""gvamp_amd -- MI355X-native engine for gVAMP's linear-model hot path (data::Ax / data::ATx inside the
CG-LMMSE VAMP loop).  The product is libgvamp.so (HIP kernels behind the C ABI of include/gvamp.h) plus the
host-side C++ mirror of the reference's `data` / `vamp` / `Options` classes under gvamp_amd/csrc/host.
`gvamp_amd.capi` is the ctypes plumbing that tests/ and bench.py use."""
__version__ = "0.1.0"
