"""xsqueezeit_amd — MI355X-native genotype-block codec for xSqueezeIt's .xsi format.

The product is the C-ABI shared library ``libxsi_hip.so`` (see ``include/xsi_hip.h``), built
from the hand-written gfx950 HIP kernels under ``csrc/``.  This package is the thin Python
harness around it: a ctypes signature table (``binding``), block sharding and the RCCL / gloo
gather of compressed block streams (``dist``), the synthetic workload generator (``synth``) and a
GT-only VCF reader for the reference's micro fixtures (``vcf_lite``).  The file-level writer /
accessor mirrors of the reference's ``XsiFactoryInterface`` / ``Accessor`` live in the library
itself (``csrc/xsi_host.hip``, ``xsi_writer_*`` / ``xsi_accessor_*``).  There is no CPU fallback: importing
``binding`` without the built library, or creating a context without a GPU, raises.
"""

__version__ = "0.1.0"
