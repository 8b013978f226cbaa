"""vdetlib_amd -- MI355X (gfx950) implementation of vdetlib's per-frame scoring and tubelet
post-processing hot path, behind the reference's own python API.

Layout mirrors the reference tree (``/root/reference``): ``utils/`` (protocol dicts, ``cython_nms``
drop-in, ``common.iou``) and ``vdet/`` (``video_det``, ``track``, ``tubelet_cls``, ``image_det``);
``ops`` holds the device-resident array forms; ``csrc/`` the HIP kernels + C-ABI
(``include/vdet_hip.h``).  There is NO CPU fallback: every numeric entry point raises if
``libvdet_hip.so`` or the GPU is missing.
"""
__version__ = "0.1"
