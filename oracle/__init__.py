"""oracle/ -- TEST INFRASTRUCTURE, not product code.

CPU restatement (torch fp32/fp64 functional ops on the host) of the ClimateGAN generator/discriminator hot
path, plus the container-only importer of the real reference (``ref_shim``) and the fixture generator
(``make_golden``).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the *checker* / reported CPU baseline -- never as the thing shipped.
The product path (``climategan_amd/``) never imports it and fails loudly if its HIP library is missing.

Parity pin: ``oracle.cpu_ref`` is validated against (a) the real reference imported in the dev container
(``tests/test_oracle_vs_reference.py``, skipped where /root/reference is absent) and (b) the committed golden
vectors under ``tests/golden/`` that were produced by the real reference (``oracle/make_golden.py``).
"""
