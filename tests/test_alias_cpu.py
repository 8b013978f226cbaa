"""The reference's import root: ``vdetlib.*`` resolves to the build's modules (no GPU needed to import)."""
import importlib

import pytest


@pytest.mark.parametrize("mod", ["utils.protocol", "utils.common", "utils.cython_nms", "utils.timer", "utils.log", "vdet.track",
                                 "vdet.video_det", "vdet.image_det", "vdet.tubelet_cls", "vdet.dataset",
                                 "tools.imagenet_annotation_processor", "tools.gen_vid_proto_file"])
def test_reference_import_names_resolve(mod):
    a = importlib.import_module("vdetlib." + mod)
    b = importlib.import_module("vdetlib_amd." + mod)
    assert a is b


def test_reference_style_from_imports():
    # the reference's own import lines (vdet/track.py:13, vdet/video_det.py:11, vdet/image_det.py:9 and T-CNN's scripts)
    from vdetlib.utils.cython_nms import nms, vid_nms, track_det_nms                    # noqa: F401
    from vdetlib.utils.protocol import proto_load, proto_dump, track_proto_from_annot_proto  # noqa: F401
    from vdetlib.utils.common import iou, options                                      # noqa: F401
    from vdetlib.vdet.track import greedily_track_from_raw_dets, greedily_track_from_det    # noqa: F401
    from vdetlib.vdet.video_det import apply_vid_nms, fast_rcnn_det_vid                # noqa: F401
    from vdetlib.vdet.tubelet_cls import score_proto_temporal_maxpool, raw_dets_spatial_max_pooling  # noqa: F401
    import vdetlib
    from vdetlib import utils
    assert vdetlib.vdet.track.greedily_track_from_raw_dets is greedily_track_from_raw_dets
    assert utils.protocol.proto_load is proto_load
    with pytest.raises(ImportError):
        importlib.import_module("vdetlib.no_such_module")


def test_monkey_patching_is_seen_through_both_names():
    import vdetlib.vdet.video_det as a
    import vdetlib_amd.vdet.video_det as b
    old = b.imread
    try:
        a.imread = lambda p: "patched"
        assert b.imread("x") == "patched"
    finally:
        b.imread = old


def test_real_modules_keep_their_own_spec():
    """importing through the alias must not rewrite the real module's __spec__ (reload / relative imports / package paths)"""
    import vdetlib.utils.timer as a
    import vdetlib_amd.utils.timer as b
    import vdetlib_amd.utils as pkg
    import vdetlib.utils                                                               # noqa: F401
    assert a is b and b.__spec__.name == "vdetlib_amd.utils.timer" and b.__package__ == b.__spec__.parent
    assert pkg.__spec__.name == "vdetlib_amd.utils" and list(pkg.__spec__.submodule_search_locations)
    assert importlib.reload(b) is b
    import vdetlib
    with pytest.raises(AttributeError):
        vdetlib.no_such_submodule
