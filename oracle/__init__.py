"""CPU oracle for the BoT-SORT per-frame update path -- TEST INFRASTRUCTURE ONLY.

Nothing in ``boxmot_amd`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker / the reported CPU baseline -- never as the thing shipped.

Parity status (see DESIGN.md "Oracle"):
  * Kalman / association / bookkeeping: pinned -- checked bit-for-bit against the
    reference's own Python classes (imported from /root/reference in the build
    container by ``tests/golden/make_golden.py``; fixtures in ``tests/golden``).
  * ``lap.lapjv`` (lapx 0.9.4) and ``cv2.resize`` (opencv-python 4.11): PARITY
    UNPINNED -- neither package exists offline; both are restated from their
    published algorithms (``oracle/lapjv.c``, ``oracle/crops.py``).
"""
