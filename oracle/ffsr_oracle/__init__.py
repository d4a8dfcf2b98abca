"""CPU oracle for the FreqFusionSR x4 inference hot path (TEST INFRASTRUCTURE ONLY).

A plain PyTorch-CPU fp32 restatement of the reference algorithm, written functionally over
reference-compatible ``state_dict``s (same keys, same shapes).  It is the checker for the HIP
engine: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product package (``image-super-resolution_amd``) never does.

Pinning: the reference ships no golden vectors for this path, so the oracle
is pinned against outputs of the reference itself, imported in the build container through
``oracle/ref_harness.py`` (``oracle/make_golden.py`` wrote ``tests/golden/*.pt``;
``tests/test_oracle_vs_reference.py`` re-checks live when /root/reference is present).
Exception: the Mamba selective scan lives in the un-vendored ``mamba-ssm`` 2.3.0 wheel, so
``scan.selective_scan_ref`` restates the published recurrence and is "parity unpinned".
"""

