"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the flowmse hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / reported baseline.  The product path
(``flowmse_amd``) never imports this package and fails loudly when the HIP
library is missing.

Parity pinning: the reference (seongq/flowmse) ships no tests, fixtures or
known-answer vectors for this path.  The oracle is pinned
against outputs of the reference itself, imported unmodified in the build
container by ``oracle/gen_golden.py`` and committed as ``tests/golden/*.npz``.
"""
