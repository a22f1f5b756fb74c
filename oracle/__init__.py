"""CPU oracle of the captioning hot path: TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package, and only as the checker / timed baseline.  ``image_captioning_amd`` never imports it.
"""
