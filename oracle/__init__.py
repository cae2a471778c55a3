"""CPU oracle for the cross-modal attention hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32/fp64)
restatement of the reference algorithm (OpenViVQA ``models/modules`` + the
pointer scorers).  It exists to (1) be checked against golden vectors produced
by the real reference (``tests/golden/make_golden.py``), (2) act as the parity
checker for the HIP path in ``tests/`` and ``__graft_entry__.smoke()``, and
(3) be timed as the ``cpu_baseline`` leg of ``bench.py``.

Nothing under ``openvivqa_amd/`` imports it; the product path has no CPU
fallback and fails loudly when the HIP library is missing.

Parity status: PINNED -- every function here is compared against outputs of the
reference itself (fixtures under ``tests/golden/*.npz``, generated in the build
container by importing ``/root/reference``; the reference has no tests or
golden vectors of its own).
"""
from .restatement import *  # noqa: F401,F403
