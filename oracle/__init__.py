"""CPU oracle for the NMF hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``nmf_amd``) never
imports this package and fails loudly when the HIP library is missing.

Parity status: PINNED.  ``oracle/nmf_ref.py`` is checked against vectors
produced by importing the reference itself in the build container
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.
"""
