"""matfree_extensions -- MI355X-native build of the Lanczos/Arnoldi-with-adjoint hot path.

Drop-in for the modules of pnkraemer/experiments-lanczos-adjoints that sit on the SLQ
log-determinant path: ``lanczos``, ``arnoldi``, ``hutchinson`` and the SLQ/kernel helpers of
``util.gp_util``; tensors are ``torch`` tensors on a ROCm device and the work is done by the
hand-written HIP kernels of ``libmfx.so`` (C-ABI in ``include/mfx.h``).
"""

from . import arnoldi, hutchinson, lanczos, operators  # noqa: F401
