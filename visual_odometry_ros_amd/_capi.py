"""ctypes binding of libvo_hip.so (C ABI in include/vo_hip.h).

The library is the product; there is no Python or CPU fallback. Importing this
module raises if the shared object has not been built, and every compute call
raises VoError when no gfx950 device is usable.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvo_hip.so")


class VoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvo_hip error {code}: {msg}")
        self.code = code


class VoConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("max_width", C.c_int), ("max_height", C.c_int),
                ("max_points", C.c_int), ("n_slots", C.c_int), ("max_level", C.c_int)]


class GnInfo(C.Structure):
    _fields_ = [("iterations", C.c_int), ("err", C.c_float), ("delta_err", C.c_float),
                ("delta_norm", C.c_float), ("cnt_invalid", C.c_int), ("is_nan", C.c_int)]


class StereoParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("win", C.c_int), ("max_level", C.c_int),
                ("thres_err", C.c_float), ("thres_bidirection", C.c_float),
                ("thres_poseba", C.c_float), ("Kl", C.c_float * 4), ("Kr", C.c_float * 4),
                ("T_lr", C.c_float * 16), ("thres_sampson", C.c_float)]


class FrameCounts(C.Structure):
    _fields_ = [("n_l0l1", C.c_int), ("n_refine", C.c_int), ("n_l1r1", C.c_int),
                ("n_inlier", C.c_int), ("n_new_ok", C.c_int), ("gn_iterations", C.c_int),
                ("n_replayed", C.c_int), ("n_ba", C.c_int)]


class MonoParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("win", C.c_int), ("max_level", C.c_int),
                ("thres_err", C.c_float), ("thres_bidirection", C.c_float),
                ("thres_poseba", C.c_int), ("thres_sampson", C.c_float), ("K", C.c_float * 4)]


class MonoCounts(C.Structure):
    _fields_ = [("n_klt", C.c_int), ("n_refine", C.c_int), ("n_ba", C.c_int),
                ("n_motion", C.c_int), ("n_final", C.c_int), ("gn_iterations", C.c_int),
                ("need_five_point", C.c_int), ("n_replayed", C.c_int)]


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_double), ("n_levels", C.c_int),
                ("edge_threshold", C.c_int), ("fast_threshold", C.c_int)]


class BinParams(C.Structure):
    _fields_ = [("n_bins_u", C.c_int), ("n_bins_v", C.c_int), ("u_step", C.c_int), ("v_step", C.c_int),
                ("inv_u_step", C.c_float), ("inv_v_step", C.c_float), ("orb", OrbParams)]


class SbaProblem(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("n_opt", C.c_int), ("n_points", C.c_int), ("n_obs", C.c_int),
                ("stereo", C.c_int), ("max_iter", C.c_int), ("Kl", C.c_double * 4), ("Kr", C.c_double * 4),
                ("T_lr", C.c_double * 16), ("thres_huber", C.c_double)]


class SvoParams(C.Structure):
    _fields_ = [("frame", StereoParams), ("bins", BinParams), ("kf_overlap_ratio", C.c_float),
                ("kf_rotation_deg", C.c_float), ("kf_translation", C.c_float), ("kf_window", C.c_int),
                ("strict_border", C.c_int), ("local_ba", C.c_int), ("rectify", C.c_int)]



class SvoFrameInfo(C.Structure):
    _fields_ = [("frame_id", C.c_int), ("is_first", C.c_int), ("is_keyframe", C.c_int), ("lba_ran", C.c_int),
                ("n_tracks_in", C.c_int), ("n_final", C.c_int), ("n_new", C.c_int), ("n_tracks_out", C.c_int),
                ("n_kf_tracked", C.c_int), ("n_new_candidates", C.c_int), ("counts", FrameCounts), ("gn", GnInfo),
                ("dT", C.c_float * 16), ("T_wc", C.c_float * 16), ("lba_err_first", C.c_double),
                ("lba_err_last", C.c_double), ("lba_landmarks", C.c_int), ("lba_observations", C.c_int)]


FIVE_POINT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float),
                            C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint8))


class MvoParams(C.Structure):
    _fields_ = [("frame", MonoParams), ("bins", BinParams), ("kf_overlap_ratio", C.c_float), ("kf_rotation_deg", C.c_float),
                ("kf_translation", C.c_float), ("kf_window", C.c_int), ("thres_parallax_deg", C.c_float),
                ("strict_border", C.c_int), ("local_ba", C.c_int), ("rectify", C.c_int), ("five_point", FIVE_POINT_FN),
                ("five_point_user", C.c_void_p)]


class MvoFrameInfo(C.Structure):
    _fields_ = [("frame_id", C.c_int), ("is_first", C.c_int), ("is_init", C.c_int), ("is_keyframe", C.c_int), ("lba_ran", C.c_int),
                ("used_five_point", C.c_int), ("n_tracks_in", C.c_int), ("n_final", C.c_int), ("n_new", C.c_int),
                ("n_tracks_out", C.c_int), ("n_kf_tracked", C.c_int), ("n_reconstructed", C.c_int), ("counts", MonoCounts),
                ("gn", GnInfo), ("dT01", C.c_float * 16), ("T_wc", C.c_float * 16), ("lba_err_first", C.c_double),
                ("lba_err_last", C.c_double), ("lba_landmarks", C.c_int), ("lba_observations", C.c_int)]


# every symbol include/vo_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "vo_abi_version", "vo_device_count", "vo_create", "vo_destroy", "vo_last_error", "vo_stream",
    "vo_synchronize", "vo_set_image", "vo_set_image_device", "vo_swap_slots", "vo_pyramid_levels",
    "vo_get_level", "vo_klt_track", "vo_track", "vo_track_bidirection",
    "vo_track_bidirection_with_prior", "vo_track_with_prior", "vo_calc_prior",
    "vo_sampson_distance", "vo_symmetric_epipolar_distance", "vo_weight_bin_update", "vo_bucket_argmax",
    "vo_track_with_scale", "vo_gn_pose_mono", "vo_gn_pose_stereo", "vo_orb_hamming",
    "vo_orb_match", "vo_compact_indices", "vo_stereo_frame_set_strict_border",
    "vo_stereo_frame_enqueue", "vo_stereo_frame_result",
    "vo_mono_frame_enqueue", "vo_mono_frame_result", "vo_mono_frame_enqueue_closed", "vo_mono_frame_new_points",
    "vo_sba_solve", "vo_orb_detect", "vo_orb_get_level", "vo_extract_orb_with_binning",
    "vo_extract_orb_with_binning_enqueue", "vo_extract_orb_with_binning_result", "vo_rectify_init_mono", "vo_rectify_init_stereo", "vo_rectify_set_maps", "vo_rectify_get_maps",
    "vo_set_image_rectified", "vo_set_image_rectified_device", "vo_set_stereo_pair_rectified_device",
    "vo_profile_enable", "vo_profile_reset", "vo_profile_get", "vo_profile_set_classes",
    "vo_set_stereo_pair_device", "vo_set_pyramid_window_hint",
    "vo_set_ingest_side_stream", "vo_set_stereo_pair_host_async", "vo_new_point_candidates_enqueue",
    "vo_new_point_candidates_get", "vo_stereo_frame_enqueue_closed", "vo_stereo_frame_new_points",
    "vo_stereo_frame_enqueue_closed_world", "vo_stereo_frame_recoveries", "vo_svo_create", "vo_svo_destroy", "vo_svo_track", "vo_svo_run", "vo_svo_enqueue",
    "vo_svo_prefetch", "vo_svo_result", "vo_svo_get_tracks", "vo_svo_get_new_points", "vo_svo_keyframe_count", "vo_svo_get_keyframe", "vo_svo_get_keyframes", "vo_triangulate_dlt", "vo_batch_create", "vo_batch_destroy",
    "vo_mvo_create", "vo_mvo_destroy", "vo_mvo_track", "vo_mvo_run", "vo_mvo_enqueue", "vo_mvo_prefetch", "vo_mvo_result", "vo_mvo_get_tracks",
    "vo_mvo_keyframe_count", "vo_mvo_get_keyframes",
    "vo_batch_last_error", "vo_batch_run", "vo_debug_set", "vo_batch_debug_set", "vo_batch_strict_border", "vo_debug_allocation_count", "vo_svo_device_bytes",
    "vo_se3_exp", "vo_ids_reset", "vo_ids_peek", "vo_ids_new_frames", "vo_ids_new_landmarks", "vo_compact_tracks",
]

_lib = None


def load():
    """Load libvo_hip.so; raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). visual_odometry_ros_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.vo_last_error.restype = C.c_char_p
    lib.vo_last_error.argtypes = [C.c_void_p]
    lib.vo_stream.restype = C.c_void_p
    lib.vo_stream.argtypes = [C.c_void_p]
    lib.vo_create.argtypes = [C.POINTER(VoConfig), C.POINTER(C.c_void_p)]
    lib.vo_destroy.argtypes = [C.c_void_p]
    lib.vo_destroy.restype = None
    # the two per-frame calls carry declared argument types: no per-call inference, plain ints as device pointers
    vp, ci = C.c_void_p, C.c_int
    lib.vo_stereo_frame_enqueue.argtypes = [vp, C.POINTER(StereoParams), ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, ci, ci]
    lib.vo_stereo_frame_result.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(FrameCounts), C.POINTER(GnInfo)]
    lib.vo_set_stereo_pair_device.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci]
    lib.vo_set_stereo_pair_host_async.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci]
    lib.vo_stereo_frame_enqueue_closed.argtypes = [vp, C.POINTER(StereoParams), ci, ci, ci, vp, vp, vp, vp, ci, vp,
                                                   C.POINTER(BinParams), ci, ci]
    lib.vo_new_point_candidates_enqueue.argtypes = [vp, ci, C.POINTER(BinParams), ci]
    lib.vo_stereo_frame_new_points.argtypes = [vp, vp, vp]
    lib.vo_mono_frame_enqueue_closed.argtypes = [vp, C.POINTER(MonoParams), ci, ci, vp, vp, vp, ci, vp, vp, vp,
                                                 C.POINTER(BinParams), ci, ci]
    lib.vo_mono_frame_new_points.argtypes = [vp, vp, vp, vp, vp]
    lib.vo_stereo_frame_enqueue_closed_world.argtypes = [vp, C.POINTER(StereoParams), ci, ci, ci, vp, vp, vp, vp, ci, vp, vp,
                                                         vp, C.POINTER(BinParams), ci, ci]
    lib.vo_svo_create.argtypes = [vp, C.POINTER(SvoParams), C.POINTER(C.c_void_p)]
    lib.vo_svo_destroy.argtypes = [vp]
    lib.vo_svo_destroy.restype = None
    lib.vo_svo_track.argtypes = [vp, vp, vp, ci, ci, C.c_double, C.POINTER(SvoFrameInfo)]
    lib.vo_svo_run.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp, vp]
    lib.vo_svo_enqueue.argtypes = [vp, vp, vp, ci, ci, C.c_double]
    lib.vo_svo_prefetch.argtypes = [vp, vp, vp, ci, ci]
    lib.vo_svo_result.argtypes = [vp, C.POINTER(SvoFrameInfo)]
    lib.vo_svo_get_tracks.argtypes = [vp, vp, vp, vp, vp, vp, ci, vp]
    lib.vo_svo_get_new_points.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.vo_svo_keyframe_count.argtypes = [vp, vp]
    lib.vo_svo_get_keyframe.argtypes = [vp, ci, vp, vp, ci, vp]
    lib.vo_svo_get_keyframes.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    lib.vo_svo_device_bytes.argtypes = [vp, vp]
    lib.vo_mvo_create.argtypes = [vp, C.POINTER(MvoParams), C.POINTER(C.c_void_p)]
    lib.vo_mvo_destroy.argtypes = [vp]
    lib.vo_mvo_destroy.restype = None
    lib.vo_mvo_track.argtypes = [vp, vp, ci, ci, C.c_double, C.POINTER(MvoFrameInfo)]
    lib.vo_mvo_run.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp]
    lib.vo_mvo_enqueue.argtypes = [vp, vp, ci, ci, C.c_double]
    lib.vo_mvo_prefetch.argtypes = [vp, vp, ci, ci]
    lib.vo_mvo_result.argtypes = [vp, C.POINTER(MvoFrameInfo)]
    lib.vo_mvo_get_tracks.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, vp]
    lib.vo_mvo_keyframe_count.argtypes = [vp, vp]
    lib.vo_mvo_get_keyframes.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    lib.vo_debug_set.argtypes = [vp, ci, ci]
    lib.vo_batch_debug_set.argtypes = [vp, ci, ci]
    lib.vo_batch_strict_border.argtypes = [vp]
    lib.vo_debug_allocation_count.argtypes = [vp, vp]
    lib.vo_triangulate_dlt.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp]
    lib.vo_batch_create.argtypes = [C.POINTER(VoConfig), C.POINTER(SvoParams), ci, C.POINTER(C.c_void_p)]
    lib.vo_batch_destroy.argtypes = [vp]
    lib.vo_batch_destroy.restype = None
    lib.vo_batch_last_error.argtypes = [vp]
    lib.vo_batch_last_error.restype = C.c_char_p
    lib.vo_batch_run.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp]
    _lib = lib
    return lib
