"""CPU oracle: TEST INFRASTRUCTURE ONLY.

A numpy restatement of the reference's algorithms for the hot path. Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker. Nothing under
``recommendation_amd/`` imports it: the product path runs on the HIP library
and fails loudly when that library is missing.

Parity status
-------------
The reference computes through TensorFlow 1.x, which is not vendored in the
reference tree and is not installable here, and the reference ships no tests
or golden vectors. ``colour_model`` therefore follows the reference's model
file op for op and is pinned by the hand-derived known-answer tests listed in
``tests/test_oracle_kat.py`` (the single known-answer snippet in the
reference's own tree, ``recommendation/test.py:19-28``, is one of them).
``fm``, ``dcn`` and ``din`` restate published formulas that the reference does
not contain at all: PARITY UNPINNED versus the reference for those.
"""
