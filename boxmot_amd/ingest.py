"""Pinned-host frame ingest ring ("real frame ingest").

The reference hands every tracker a numpy frame per call (basetracker.py:120-147) and its native binding copies it again
(native/trackers/botsort.py:200-230).  On a PCIe-attached GPU the per-frame upload (6.2 MB at 1080p) from pageable memory is
the larger part of a host-API step, so the ring gives the decoder page-locked buffers to write into and moves them with
asynchronous DMAs on a copy stream of their own:

    ring = FrameRing(n_slots=3, n_streams=S, rows=1080, cols=1920)
    ring.host_view(0)[s] = first frame of stream s ; ring.submit(0)
    for t in range(T):
        k, k1 = t % 3, (t + 1) % 3
        ring.host_view(k1)[...] = frames of t + 1            # decode straight into pinned memory (host_done(k1) first if reused)
        ring.submit(k1)                                      # its upload overlaps the kernels of frame t
        rows = tracker.update_batch(dets[t], ring=ring, slot=k)   # MultiStreamBotSort: waits for slot k on the device, no host wait

Everything is a thin ctypes wrapper over boxmot_hip_ingest_* (include/boxmot_hip.h).
"""
from __future__ import annotations

import ctypes
import sys

import numpy as np

from boxmot_amd import _lib


class FrameRing:
    def __init__(self, n_slots: int, n_streams: int, rows: int, cols: int):
        self._lib = _lib.load()
        self.n_slots, self.n_streams, self.rows, self.cols = int(n_slots), int(n_streams), int(rows), int(cols)
        self._handle = self._lib.boxmot_hip_ingest_create(self.n_slots, self.n_streams, self.rows, self.cols)
        if not self._handle:
            raise RuntimeError(_lib.last_error())
        self._views = {}
        self._roots = {}

    def host_view(self, slot: int) -> np.ndarray:
        """(n_streams, rows, cols, 3) uint8 view of the slot's page-locked host memory."""
        if slot not in self._views:
            p = self._lib.boxmot_hip_ingest_host_ptr(self._handle, int(slot), 0)
            if not p:
                raise RuntimeError(_lib.last_error())
            n = self.n_streams * self.rows * self.cols * 3
            buf = (ctypes.c_uint8 * n).from_address(p)
            root = np.frombuffer(buf, dtype=np.uint8)        # numpy collapses view chains onto this array: every slice a caller
            self._roots[slot] = root                         # keeps holds a reference to IT (close() counts them)
            self._views[slot] = root.reshape(self.n_streams, self.rows, self.cols, 3)
        return self._views[slot]

    def submit(self, slot: int, n_streams: int | None = None) -> None:
        _lib.check(self._lib.boxmot_hip_ingest_submit(self._handle, int(slot), int(n_streams or self.n_streams)))

    def wait(self, slot: int, consumer_stream: int) -> None:
        _lib.check(self._lib.boxmot_hip_ingest_wait(self._handle, int(slot), ctypes.c_void_p(consumer_stream)))

    def release(self, slot: int, consumer_stream: int) -> None:
        _lib.check(self._lib.boxmot_hip_ingest_release(self._handle, int(slot), ctypes.c_void_p(consumer_stream)))

    def host_done(self, slot: int) -> None:
        _lib.check(self._lib.boxmot_hip_ingest_host_done(self._handle, int(slot)))

    def device_frames(self, slot: int) -> int:
        """Device address of the slot's table of per-stream frame pointers (the ``d_frames`` of ``step_device``)."""
        p = self._lib.boxmot_hip_ingest_device_frames(self._handle, int(slot))
        if not p:
            raise RuntimeError(_lib.last_error())
        return int(p)

    def close(self, force: bool = False) -> None:
        """Free the ring.  The arrays ``host_view`` handed out alias the page-locked memory this frees: while the caller still
        holds one (or a slice of one) ``close`` refuses, unless ``force`` (interpreter shutdown / ``__del__``)."""
        h = getattr(self, "_handle", None)
        if h:
            if not force:
                # a slot's root array is referenced by: _roots, the cached 4-D view's base, getrefcount's argument; and the 4-D
                # view by: _views, getrefcount's argument
                held = [s for s in self._views if sys.getrefcount(self._roots[s]) > 3 or sys.getrefcount(self._views[s]) > 2]
                if held:
                    raise RuntimeError(f"FrameRing.close(): host views of slot(s) {held} are still referenced; drop them first")
            self._views.clear()
            self._roots.clear()
            self._lib.boxmot_hip_ingest_destroy(h)
            self._handle = None

    def __del__(self):
        try:
            self.close(force=True)
        except Exception:
            pass
