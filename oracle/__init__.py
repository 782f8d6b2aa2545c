"""CPU oracle for the SINDy hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and only as the checker / reported CPU baseline, never as the thing measured or shipped.
The product package (``symmetry-ode-discovery_amd``) never imports this package and
fails loudly when its HIP library is missing.
"""
