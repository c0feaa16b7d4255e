"""ctypes binding of libdeepdish_hip.so (the C ABI declared in include/deepdish_hip.h).

There is deliberately no fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import c_int, c_int64, c_double, c_float, c_void_p, c_char_p, POINTER

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DD_LIB') or os.path.join(HERE, 'libdeepdish_hip.so')      # DD_LIB: another build of the same ABI (same-box A/B runs)

P = c_void_p


class DeepDishHipError(RuntimeError):
    pass


# name -> argtypes (every function returns int unless listed in _RESTYPE)
SIGNATURES = {
    'dd_last_error': [],
    'dd_version': [],
    'dd_ctx_create': [c_int, POINTER(P)],
    'dd_ctx_destroy': [P],
    'dd_ctx_stream': [P, POINTER(P)],
    'dd_ctx_sync': [P],
    'dd_kf_initiate': [P, P, P, P, P, c_int, P],
    'dd_kf_predict': [P, P, P, P, c_int, P],
    'dd_kf_project': [P, P, P, P, c_int, P, P, P],
    'dd_kf_update': [P, P, P, P, P, c_int, P],
    'dd_kf_gate': [P, P, P, P, c_int, P, c_int, c_int, P, P],
    'dd_iou_cost': [P, P, P, c_int, P, c_int, P, P],
    'dd_cosine_nn_cost': [P, P, P, c_int, P, c_int, P, P],
    'dd_nms': [P, P, P, c_int, c_double, P, P, P],
    'dd_nms_ssd': [P, P, P, c_int, c_double, P, P, P],
    'dd_lsap_host': [P, c_int, c_int, P, P],
    'dd_pyset_difference_order_host': [P, c_int, P, c_int, P, POINTER(c_int)],
    'dd_tracker_create': [P, c_double, c_double, c_int, c_int, c_int, c_int, c_int, POINTER(P)],
    'dd_tracker_destroy': [P],
    'dd_tracker_predict': [P],
    'dd_tracker_update': [P, P, P, c_int, c_int],
    'dd_tracker_count': [P, c_int, POINTER(c_int)],
    'dd_tracker_read': [P, c_int, P, P, P],
    'dd_tracker_track_update': [P, c_int64, P, P, c_int],
    'dd_tracker_track_predict': [P, c_int64],
    'dd_tracker_track_set': [P, c_int64, c_int, c_int],
    'dd_tracker_remove': [P, P, c_int],
    'dd_tracker_next_id': [P, POINTER(c_int64)],
    'dd_tracker_last_cost': [P, P, P, c_int, POINTER(c_int), POINTER(c_int)],
    'dd_tracker_last_matches': [P, P, c_int, POINTER(c_int)],
    'dd_crop_resize': [P, P, c_int, c_int, P, c_int, c_int, c_int, P, P, P],
    'dd_crop_resize_f64': [P, P, c_int, c_int, P, c_int, c_int, c_int, P, P, P],
    'dd_fake_encode': [P, P, c_int, c_int, P, P],
    'dd_resize_lanczos': [P, P, c_int, c_int, c_int, c_int, P, c_int, c_int, P],
    'dd_resize_lanczos_batch': [P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P],
    'dd_resize_bilinear': [P, P, c_int, c_int, c_int, P, c_int, c_int, P],
    'dd_ingest_create': [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    'dd_ingest_destroy': [P],
    'dd_ingest_host_slot': [P, c_int, P, P],
    'dd_ingest_submit': [P, c_int],
    'dd_ingest_wait_uploaded': [P, c_int],
    'dd_ingest_acquire': [P, c_int, P, P],
    'dd_ingest_release': [P, c_int, P],
    'dd_mog2_create': [P, c_int, c_int, c_int, c_int, c_double, c_int, POINTER(P)],
    'dd_mog2_destroy': [P],
    'dd_mog2_state': [P, c_int, P, P, P, P],
    'dd_mog2_apply': [P, P, c_double, P, P, P],
    'dd_mask_box_count': [P, P, c_int, c_int, c_int, P, P, c_int, P, P],
    'dd_net_create': [P, P, c_int, P, c_int64, c_int, POINTER(P)],
    'dd_net_create_shared': [P, P, c_int, P, c_int64, c_int, POINTER(P)],
    'dd_net_activation_bytes': [P, POINTER(c_int64)],
    'dd_net_destroy': [P],
    'dd_net_forward': [P, P, c_int, P],
    'dd_net_output': [P, c_int, POINTER(P), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                      POINTER(c_int)],
    'dd_net_use_graph': [P, c_int],
    'dd_net_input_size': [P, POINTER(c_int), POINTER(c_int)],
    'dd_net_max_batch': [P, POINTER(c_int)],
    'dd_net_last_batch': [P, POINTER(c_int)],
    'dd_net_read': [P, c_int, c_int, P, c_int, P],
    'dd_net_profile': [P, c_int],
    'dd_net_profile_read': [P, P, c_int, POINTER(c_int)],
    'dd_net_op_launches': [P, P, c_int, POINTER(c_int)],
    'dd_ssd_postprocess': [P, P, P, c_int, c_int, c_int, c_float, c_float, P, P, P, P, P],
    'dd_ssd_decode': [P, P, P, c_int, c_int, c_float, P, P, P, P, c_int, P],
    'dd_ssd_postprocess_decoded': [P, P, P, P, P, c_int, c_int, c_float, c_float, P, P, P, P, c_int, P],
    'dd_net_ssd_decode': [P, P, c_int, c_float, c_int],
    'dd_net_ssd_decoded': [P, POINTER(P), POINTER(P), POINTER(P), POINTER(P)],
    'dd_net_ssd_decoded_read': [P, c_int, P, P, P, P],
    'dd_net_yolo_decode': [P, c_int],
    'dd_net_yolo_decoded': [P, P, P, P, P],
    'dd_net_yolo_decoded_read': [P, c_int, P, P, P],
    'dd_ssd_detections': [P, P, P, P, c_int, c_int, c_double, c_double, c_double, c_double, P, P, P, P, P],
    'dd_yolov5_decode': [P, P, c_int, c_int, c_float, c_float, c_float, P, P, P, c_int, P, P],
    'dd_pipeline_create': [P, c_int, c_int, c_int, P, P, c_int, c_int, P, c_char_p, c_char_p, c_double, c_double,
                           c_double, c_int, c_int, P, c_int, c_int, POINTER(P)],
    'dd_pipeline_destroy': [P],
    'dd_pipeline_detector_adaptor': [P, c_int],
    'dd_pipeline_ssd_options': [P, c_int, c_float, c_float],
    'dd_pipeline_step': [P, P, P, P, P, P],
    'dd_pipeline_step2': [P, P, P, P, P, P, P],
    'dd_pipeline_background_subtraction': [P, c_double, c_int],
    'dd_pipeline_motion_mask': [P, P, c_int, POINTER(ctypes.c_longlong)],
    'dd_pipeline_counts': [P, P],
    'dd_pipeline_tracker': [P, c_int, POINTER(P)],
    'dd_pipeline_stage_seconds': [P, P, POINTER(ctypes.c_longlong)],
    'dd_pipeline_stage_gpu_ms': [P, P, POINTER(ctypes.c_longlong)],
    'dd_pipeline_detector_stream': [P, POINTER(P)],
    'dd_pipeline_detections': [P, c_int, P, P, P, c_int, POINTER(c_int)],
    'dd_counts_accumulate': [P, P, P, c_int, P],
}
_RESTYPE = {'dd_last_error': c_char_p}

_lib = None
MISSING = []


def lib():
    """Load the shared library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeepDishHipError(
            'libdeepdish_hip.so is missing (%s): build it with `python -m deepdish_amd.build`; '
            'there is no CPU fallback' % LIB_PATH)
    # torch first: its wheel carries its own libamdhip64, and a process that loads /opt/rocm's copy through this library BEFORE torch
    # ends up with two HIP runtimes (seen: build() then smoke() in one process -> "no ROCm-capable device is detected")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    l = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:         # header and library out of sync; tests/test_abi.py fails on this
            MISSING.append(name)
            continue
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, c_int)
    _lib = l
    return l


def check(rc, what=''):
    if rc != 0:
        msg = lib().dd_last_error()
        raise DeepDishHipError('%s failed (%d): %s' % (what or 'libdeepdish_hip call', rc,
                                                      msg.decode() if msg else '?'))
