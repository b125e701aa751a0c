"""ORACLE — test infrastructure only.

CPU (torch fp32, NCHW) restatement of the reference's CNN hot path, used as
the parity checker by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``syke-pic_amd/`` may
import it: the product path has no CPU fallback and raises when the HIP
library is missing.  Pinned by the golden vectors in ``tests/golden/`` that
were produced by the reference's own code (see ``refnet.py`` header).
"""
