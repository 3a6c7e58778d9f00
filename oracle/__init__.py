"""CPU oracle for the BLP scoring / ranking hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``blp_amd/`` imports it; the product path fails loudly when the HIP
library is missing instead of falling back to this code.
"""
