"""Stand-in for the slice of `xformers` that vLLM v0.1.0 imports; see
../README.md.  Only block-diagonal causal attention without dropout exists
here, computed by the MI355X prompt attention kernel."""
from . import ops  # noqa: F401

__version__ = "0.0.19+mi355x.shim"
