r"""CPU oracle for the azula sampling hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (plain ``torch`` fp32 eager ops on CPU tensors, in the
reference's op order) of the one path of probabilists/azula that ``azula_amd`` accelerates:

    Sampler.__call__ -> per step { VPSchedule alpha/sigma -> Karras / ADM preconditioning
                                   -> backbone forward -> DDPM / DDIM transition }

It is *not* part of the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.  The product
(``azula_amd``) never imports it and has no CPU fallback for device tensors.

Every function cites the reference ``file:line`` (relative to ``/root/reference``) it follows.
The arithmetic itself lives in the third-party dependency PyTorch ATen (container pin:
``torch 2.10.0+rocm7.0``; reference requires ``torch>=2.0.0``, ``pyproject.toml:21``), so the
oracle calls the same ATen CPU ops at the same call sites.

Pinning: ``oracle/make_golden.py`` imports the real reference from ``/root/reference`` (this
container only), checks every oracle function against it (bit-exact on CPU) and writes the
golden vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks the oracle
against those committed vectors on every run.  The reference's own tests pin only the
VE-reschedule invariance (``tests/test_denoise.py:135-143``), which is restated as G7.
"""

from . import sampling  # noqa: F401
from . import nets  # noqa: F401
