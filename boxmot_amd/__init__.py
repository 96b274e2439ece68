"""boxmot_amd -- MI355X-native tracker update path (BoT-SORT, ByteTrack, DeepOCSORT, OC-SORT, StrongSORT; HIP kernels behind a C ABI).

Public surface (mirrors the reference's for this path):
  BotSort, ByteTrack, DeepOcSort, OcSort, StrongSort, HipReID, TrackResults, create_tracker, MultiStreamBotSort.
"""
__version__ = "0.1.0"

__all__ = ["BotSort", "ByteTrack", "DeepOcSort", "OcSort", "StrongSort", "HipReID", "TrackResults", "create_tracker", "MultiStreamBotSort"]


def __getattr__(name):
    if name == "BotSort":
        from boxmot_amd.botsort import BotSort
        return BotSort
    if name == "ByteTrack":
        from boxmot_amd.bytetrack import ByteTrack
        return ByteTrack
    if name == "DeepOcSort":
        from boxmot_amd.deepocsort import DeepOcSort
        return DeepOcSort
    if name == "OcSort":
        from boxmot_amd.deepocsort import OcSort
        return OcSort
    if name == "StrongSort":
        from boxmot_amd.strongsort import StrongSort
        return StrongSort
    if name == "HipReID":
        from boxmot_amd.reid import HipReID
        return HipReID
    if name == "TrackResults":
        from boxmot_amd.track_results import TrackResults
        return TrackResults
    if name == "create_tracker":
        from boxmot_amd.tracker_zoo import create_tracker
        return create_tracker
    if name == "MultiStreamBotSort":
        from boxmot_amd.streams import MultiStreamBotSort
        return MultiStreamBotSort
    raise AttributeError(name)
