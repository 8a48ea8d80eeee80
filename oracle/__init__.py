"""CPU oracle for the DSNT contour-regression hot path.  TEST INFRASTRUCTURE ONLY.

This package is a PyTorch-CPU (fp32) restatement of the reference's arithmetic for
the path named in BASELINE.json (`north_star`), written from the reference's text
and pinned against outputs of the reference's own importable leaf modules
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from here; the product path
(``contouring-uncertainty_amd/``) never does and fails loudly when the HIP
extension is missing.

Parity status: the reference ships no golden vectors / known-answer tests for this
path.  The oracle is pinned by golden vectors generated
*here* from the imported reference modules (torch 2.10 CPU semantics); the
generating script is ``oracle/make_golden.py``.
"""
