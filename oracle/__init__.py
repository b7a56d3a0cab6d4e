"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
only as the checker / the timed CPU baseline.  The product package
(``zeroshotvideoclassification_amd``) never imports it and has no CPU fallback.

Parity status: the reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md section 4 / 8c), so the oracle is pinned against the reference's own
``resnet.py`` / ``network.py`` imported and run in the build container
(``oracle/reference_import.py`` + ``oracle/make_golden.py``); the resulting vectors are
committed under ``tests/golden/``.  The arithmetic itself lives in PyTorch ATen
(version unpinned by the reference; torch 2.10 CPU here).
"""
