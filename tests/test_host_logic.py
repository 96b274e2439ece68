"""Host-side logic that needs no GPU: result view, factory config flattening, weight packing
layout, scenario determinism, input validation of the plugin surface."""
import numpy as np
import pytest

from boxmot_amd.basetracker import BaseTracker
from boxmot_amd.scenario import Scenario, stress_frames
from boxmot_amd.track_results import TrackResults
from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS, flatten_yaml_config


class _Probe(BaseTracker):
    """Records what _update_impl receives (no device)."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.calls = []

    def _update_impl(self, dets, img, embs=None, masks=None, class_list=0):
        self.check_inputs(dets, img, embs)
        self.calls.append((dets.copy(), None if embs is None else embs.copy(), class_list, self.frame_count))
        self.frame_count += 1
        out = np.zeros((len(dets), 8), dtype=np.float32)
        out[:, :4] = dets[:, :4]
        out[:, 6] = dets[:, 5]
        return out


IMG = np.zeros((64, 64, 3), dtype=np.uint8)


def test_track_results_view():
    r = TrackResults(np.array([[1, 2, 11, 22, 7, 0.9, 3, 5]], dtype=np.float64))
    assert r.dtype == np.float32 and r.shape == (1, 8)
    assert r.id.tolist() == [7] and r.cls.tolist() == [3] and r.det_ind.tolist() == [5]
    assert np.allclose(r.xywh, [[6, 12, 10, 20]])
    assert TrackResults(np.empty((0, 8))).shape == (0, 8)
    assert TrackResults(np.array([])).shape == (0, 0)          # DeepOCSORT-style empty (track_results.py:24-26)
    assert r.to_mot_lines(3)[0].startswith("3,7,1.00,2.00,10.00,20.00,")


def test_empty_and_none_inputs_become_empty_dets():
    t = _Probe()
    for dets in (None, np.array([]), np.empty((0, 6))):
        out = t.update(dets, IMG)
        assert out.shape == (0, 8)
    assert all(c[0].shape == (0, 6) for c in t.calls)


def test_input_validation_matches_reference_errors():
    t = _Probe()
    with pytest.raises(AssertionError):
        t.update(np.zeros((2, 5), dtype=np.float32), IMG)             # bad width (test_trackers.py:577-591)
    with pytest.raises(AssertionError):
        t.update(np.zeros((2, 6), dtype=np.float32), IMG, np.zeros((3, 4)))   # embs mismatch (:560-574)
    with pytest.raises(AssertionError, match="OBB"):
        _Probe().update(np.zeros((2, 7), dtype=np.float32), IMG)


def test_float64_dets_are_cast_to_float32_like_the_reference():
    t = _Probe()
    t.update(np.array([[0, 0, 10, 10, 0.9, 1]], dtype=np.float64), IMG)
    assert t.calls[0][0].dtype == np.float32


def test_per_class_fanout():
    t = _Probe(per_class=True, nr_classes=3)
    dets = np.array([[0, 0, 5, 5, .9, 0], [1, 1, 6, 6, .8, 2], [2, 2, 7, 7, .7, 0]], dtype=np.float32)
    embs = np.arange(12, dtype=np.float32).reshape(3, 4)
    out = t.update(dets, IMG, embs)
    assert [c[2] for c in t.calls] == [0, 1, 2]
    assert [len(c[0]) for c in t.calls] == [2, 0, 1]
    assert [c[3] for c in t.calls] == [0, 0, 0]                  # every class sees the same frame count
    assert t.frame_count == 1
    assert out.shape == (3, 8) and out[:, 6].tolist() == [0, 0, 2]
    assert t.calls[2][1].tolist() == [[4, 5, 6, 7]]


def test_yaml_flattening_and_defaults():
    cfg = {"track_high_thresh": {"type": "uniform", "default": 0.6, "range": [0.3, 0.7]},
           "with_reid": {"type": "choice", "default": True, "activates": {"proximity_thresh": {"default": 0.61}}}}
    assert flatten_yaml_config(cfg) == {"track_high_thresh": 0.6, "with_reid": True, "proximity_thresh": 0.61}
    assert BOTSORT_YAML_DEFAULTS["track_buffer"] == 40 and BOTSORT_YAML_DEFAULTS["removed_stracks_buffer"] == 329


def test_weight_blob_layout_and_folding():
    import torch

    from boxmot_amd.reid_weights import ARCH_CHANNELS, HEADER_INTS, MAGIC, pack_osnet, random_osnet_state_dict

    sd = random_osnet_state_dict("osnet_x0_25", seed=3, calib_batch=1)
    blob = pack_osnet(sd)
    hdr = blob[:HEADER_INTS].view(np.int32)
    assert hdr[0] == MAGIC and tuple(hdr[1:5]) == ARCH_CHANNELS["osnet_x0_25"] and hdr[5] == 512
    assert hdr[6] == blob.size - HEADER_INTS
    n_params = sum(v.numel() for k, v in sd.items() if k.endswith("weight") and v.dim() > 1)
    assert blob.size - HEADER_INTS > n_params      # folded biases add to the conv/linear weights
    # folded stem conv reproduces conv+BN on a probe input
    w = blob[HEADER_INTS:HEADER_INTS + 16 * 147].reshape(16, 7, 7, 3)
    b = blob[HEADER_INTS + 16 * 147:HEADER_INTS + 16 * 147 + 16]
    x = torch.randn(1, 3, 16, 16, generator=torch.Generator().manual_seed(0))
    want = torch.nn.functional.batch_norm(
        torch.nn.functional.conv2d(x, sd["conv1.conv.weight"], None, 2, 3), sd["conv1.bn.running_mean"],
        sd["conv1.bn.running_var"], sd["conv1.bn.weight"], sd["conv1.bn.bias"], False, 0.0, 1e-5)
    got = torch.nn.functional.conv2d(x, torch.from_numpy(np.transpose(w, (0, 3, 1, 2)).copy()), torch.from_numpy(b.copy()), 2, 3)
    assert torch.allclose(got, want, atol=1e-5)


def test_scenario_is_deterministic_and_keeps_the_pool_full():
    a, b = Scenario(64, 256, random_image=False), Scenario(64, 256, random_image=False)
    fa, fb = a.frames(20), b.frames(20)
    assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(fa, fb))
    assert [len(f[0]) for f in fa[:4]] == [256, 256, 256, 64]
    seen = set()
    for t in range(3, 3 + a.n_groups):
        seen |= set(a.visible(t).tolist())
    assert len(seen) == 256                       # every object is re-seen within one rotation
    assert a.n_groups < 30                        # ... well inside track_buffer
    s = stress_frames(50)
    assert any(len(d) == 0 for d, _ in s) and max(len(d) for d, _ in s) <= 25


def test_cmc_provider_surface_without_a_device():
    """boxmot_amd.cmc: the factory names what exists, argument checks are host-side, and nothing estimates on the CPU."""
    from boxmot_amd.cmc import HipECC, HipSOF, get_cmc_method
    assert get_cmc_method("ecc") is HipECC and get_cmc_method("sof") is HipSOF
    with pytest.raises(NotImplementedError, match="orb"):
        get_cmc_method("orb")
    with pytest.raises(NotImplementedError):
        HipECC(warp_mode=1)                          # MOTION_EUCLIDEAN: only the reference's default translation model is built
    with pytest.raises(NotImplementedError):
        HipECC(align=True)
    e = HipECC()                                     # no handle until the first frame
    with pytest.raises(ValueError):
        e.apply(np.zeros((10, 10), np.uint8))
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):            # no CPU fallback: the estimator fails loudly without a HIP device
            e.apply(np.zeros((64, 64, 3), np.uint8))
        from boxmot_amd.ingest import FrameRing
        with pytest.raises(RuntimeError):
            FrameRing(2, 1, 64, 64)


def test_every_module_of_the_package_and_tools_compiles():
    """A syntax error in a lazily imported module must not wait for the GPU box to be found."""
    import py_compile
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    files = sorted((root / "boxmot_amd").glob("*.py")) + sorted((root / "tools").glob("*.py")) + sorted((root / "oracle").glob("*.py")) \
        + [root / "bench.py", root / "__graft_entry__.py"]
    for f in files:
        py_compile.compile(str(f), doraise=True)
    import importlib
    for f in sorted((root / "boxmot_amd").glob("*.py")):
        if f.stem != "__init__":
            importlib.import_module(f"boxmot_amd.{f.stem}")


def test_frame_ring_close_refuses_while_host_views_are_held(monkeypatch):
    """FrameRing.host_view hands out arrays that alias the ring's page-locked memory; close() frees it.  With a stand-in library
    (plain host memory, no device) the wrapper's own logic is checked: a held view or slice blocks close(), dropping it unblocks."""
    import ctypes

    from boxmot_amd import ingest

    class FakeLib:
        def __init__(self):
            self.bufs, self.destroyed = {}, 0

        def boxmot_hip_ingest_create(self, n_slots, n_streams, rows, cols):
            self.n = n_streams * rows * cols * 3
            return 1

        def boxmot_hip_ingest_host_ptr(self, h, slot, stream):
            self.bufs.setdefault(slot, (ctypes.c_uint8 * self.n)())
            return ctypes.addressof(self.bufs[slot])

        def boxmot_hip_ingest_destroy(self, h):
            self.destroyed += 1

    fake = FakeLib()
    monkeypatch.setattr(ingest._lib, "load", lambda: fake)
    ring = ingest.FrameRing(2, 3, 4, 5)
    ring.host_view(0)[...] = 7                      # temporaries do not count
    assert ring.host_view(0).shape == (3, 4, 5, 3) and fake.bufs[0][0] == 7
    held = ring.host_view(1)[2]                     # a slice: numpy hangs it on the slot's root array
    with pytest.raises(RuntimeError, match="still referenced"):
        ring.close()
    assert fake.destroyed == 0
    del held
    whole = ring.host_view(0)
    with pytest.raises(RuntimeError, match="still referenced"):
        ring.close()
    del whole
    ring.close()
    assert fake.destroyed == 1
    ring.close()                                    # idempotent
    assert fake.destroyed == 1


def test_multi_stream_handle_over_the_emulated_abi_equals_per_stream_oracles(monkeypatch):
    """boxmot_amd.streams.MultiStreamBotSort (the bench's tracker object) on the build container's CPU: update_batch over the emulated
    ABI (tests/emu_lib.py: one emulated step per stream) -- every stream equals its own oracle, empty streams included."""
    from boxmot_amd import _lib
    from boxmot_amd.scenario import stress_frames
    from boxmot_amd.streams import MultiStreamBotSort
    from emu_lib import EmuHipLib
    from oracle.botsort import BotSortOracle
    lib = EmuHipLib()
    monkeypatch.setattr(_lib, "load", lambda: lib)
    monkeypatch.setattr(_lib, "last_error", lambda: lib.boxmot_hip_last_error().decode())
    S, n = 3, 40
    ms = MultiStreamBotSort(S, max_tracks=128, max_dets=64, emb_dim=32)
    orcs = [BotSortOracle() for _ in range(S)]
    frames = [list(stress_frames(n, seed=11 + s)) for s in range(S)]
    for t in range(n):
        got = ms.update_batch([frames[s][t][0] for s in range(S)], None, [frames[s][t][1] for s in range(S)])
        for s in range(S):
            want = np.asarray(orcs[s].update(frames[s][t][0].copy(), None, frames[s][t][1].copy()), dtype=np.float32).reshape(-1, 8)
            g = np.asarray(got[s]).reshape(-1, 8)
            assert g.shape == want.shape and np.array_equal(g[:, 4:], want[:, 4:]), (t, s)
            assert np.allclose(g[:, :4], want[:, :4], rtol=0, atol=1e-3), (t, s)
    ms.close()
