"""ctypes binding of liborbhip.so (include/orbhip.h) — the host-side mirror of the reference's two classes.

`ORBextractor` and `ORBmatcher` below keep the names, argument meaning and outputs of ORB_SLAM2::ORBextractor
(ORBextractor.h:45-111) and ORB_SLAM2::ORBmatcher (ORBmatcher.h:37-102) for the hot path this repository
replaces; they call the C ABI only.  There is no CPU fallback: if the HIP library is missing or no GPU is usable
the constructors raise `OrbHipError`.

The optional `library=` constructor argument names another build of the same C ABI; only the CPU test-suite uses it
(the fiber-emulation build of the same kernel sources, tests/emu/liborbhip_emu.so, where no GPU exists).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

# every symbol include/orbhip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "orbhip_version", "orbhip_thread_release", "orbhip_thread_api_ms", "orbhip_device_count", "orbhip_last_error", "orbhip_create", "orbhip_destroy", "orbhip_keypoint_capacity",
    "orbhip_get_scale_tables", "orbhip_level_size", "orbhip_extract", "orbhip_extract_batch", "orbhip_pyramid_level",
    "orbhip_extract_device", "orbhip_sync", "orbhip_fetch", "orbhip_fetch_matches", "orbhip_descriptor_distance",
    "orbhip_hamming_nn", "orbhip_hamming_nn_device", "orbhip_nn_expanded_size", "orbhip_nn_expand_device", "orbhip_hamming_nn_device_expanded", "orbhip_search_for_initialization", "orbhip_profile_enable",
    "orbhip_profile_num_kernels", "orbhip_profile_get", "orbhip_profile_reset", "orbhip_pyramid_cascade_tiles", "orbhip_algorithmic_bytes_per_frame",
    "orbhip_algorithmic_bytes_per_frame_kernel", "orbhip_debug_blurred_level", "orbhip_debug_candidates",
    "orbhip_compute_stereo_matches", "orbhip_extract_stereo", "orbhip_search_by_projection", "orbhip_extract_batch_color",
    "orbhip_extract_device_color", "orbhip_voc_load_text", "orbhip_voc_destroy", "orbhip_voc_info", "orbhip_voc_transform_features",
    "orbhip_voc_transform", "orbhip_compute_bow", "orbhip_fetch_bow", "orbhip_voc_score", "orbhip_search_by_bow", "orbhip_search_for_triangulation", "orbhip_search_best_in_window",
    "orbhip_search_by_bow_batch", "orbhip_search_for_triangulation_batch", "orbhip_search_best_in_window_batch",
    "orbhip_undistort_points", "orbhip_image_bounds", "orbhip_set_camera", "orbhip_get_bounds", "orbhip_fetch_undistorted",
    "orbhip_search_for_initialization_bounds", "orbhip_search_by_projection_bounds", "orbhip_search_best_in_window_bounds",
    "orbhip_set_rectification", "orbhip_extract_batch_rectify", "orbhip_extract_device_rectify", "orbhip_compute_stereo_from_rgbd", "orbhip_set_stereo_columns",
    "orbhip_search_by_projection_frame", "orbhip_search_best_in_window_frame", "orbhip_search_by_projection_batch",
    "orbhip_pyramid_fetch_all", "orbhip_set_blur_rounding", "orbhip_set_fp_contract", "orbhip_submit", "orbhip_collect", "orbhip_ring_depth", "orbhip_host_alloc", "orbhip_host_free",
    "orbhip_pool_create", "orbhip_pool_destroy", "orbhip_pool_num_devices", "orbhip_pool_device_of", "orbhip_pool_keypoint_capacity", "orbhip_pool_numa_node",
    "orbhip_pool_extract", "orbhip_pool_submit", "orbhip_pool_collect", "orbhip_pool_db_load", "orbhip_pool_db_shard", "orbhip_pool_db_query",
    "orbhip_reloc_candidates", "orbhip_runtime_info", "orbhip_device_alloc", "orbhip_device_free", "orbhip_device_upload", "orbhip_device_download",
    "orbhip_device_synchronize", "orbhip_submit_to",
    "orbhip_predict_scale_table", "orbhip_project_search_bounds", "orbhip_project_search_frame", "orbhip_project_best_in_window_bounds", "orbhip_project_best_in_window_batch", "orbhip_project_best_in_window_shared", "orbhip_project_best_in_window_held",
]


class OrbHipError(RuntimeError):
    pass


class Camera(C.Structure):          # orbhip_camera: mK (Tracking.cc:60-68) + mDistCoef (Tracking.cc:70-82)
    _fields_ = [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]

    @staticmethod
    def of(camera):
        """(fx, fy, cx, cy, k1, k2, p1, p2[, k3]) or a Camera"""
        if isinstance(camera, Camera):
            return camera
        c = [float(v) for v in camera]
        assert len(c) in (8, 9), "camera = (fx, fy, cx, cy, k1, k2, p1, p2[, k3])"
        return Camera(*(c + [0.0])[:9])


class Bounds(C.Structure):          # orbhip_bounds: Frame::mnMinX, mnMinY, mnMaxX, mnMaxY
    _fields_ = [(n, C.c_float) for n in ("min_x", "min_y", "max_x", "max_y")]

    @staticmethod
    def of(bounds, im_w=None, im_h=None):
        if isinstance(bounds, Bounds):
            return bounds
        if bounds is None:
            return Bounds(0.0, 0.0, float(im_w), float(im_h))
        return Bounds(*[float(v) for v in bounds])

    def array(self):
        return np.array([self.min_x, self.min_y, self.max_x, self.max_y], np.float32)


MAX_PROJ_LEVELS = 16
LEVEL_OF = C.CFUNCTYPE(C.c_int, C.c_float, C.c_void_p)       # int level_of(float ratio, void* user): the caller's own PredictScale expression
PROJ_LAST_FRAME, PROJ_FRAME_KF, PROJ_KF_SIM3, PROJ_FUSE, PROJ_FUSE_SIM3, PROJ_SIM3 = range(6)        # orbhip_projection_kind


class Projection(C.Structure):      # orbhip_projection: what one call of a pose-guided ORBmatcher member holds fixed
    _fields_ = [("kind", C.c_int32), ("gemm_mode", C.c_int32), ("R", C.c_float * 9), ("t", C.c_float * 3), ("R2", C.c_float * 9), ("t2", C.c_float * 3), ("Ow", C.c_float * 3),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float), ("th", C.c_float),
                ("forward", C.c_int32), ("backward", C.c_int32), ("nlevels", C.c_int32),
                ("scale_factors", C.c_float * MAX_PROJ_LEVELS), ("level_ratio", C.c_float * MAX_PROJ_LEVELS)]


# orbhip_map_point: what a member reads of one map point
MAP_POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("cam_x", "<f4"), ("cam_y", "<f4"), ("cam_z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
                            ("min_dist", "<f4"), ("max_dist", "<f4"), ("scale_dist", "<f4"), ("level", "<i4"), ("blocks", "<i4"), ("angle", "<f4")])
assert MAP_POINT_DTYPE.itemsize == 60


class Config(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32), ("ini_th_fast", C.c_int32),
                ("min_th_fast", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("max_batch", C.c_int32),
                ("device", C.c_int32), ("stream", C.c_void_p), ("blur_round_mode", C.c_int32), ("num_streams", C.c_int32)]


def library_path():
    """The in-tree liborbhip.so.  ORBHIP_LIBRARY (measurement aid) names another build of the same C ABI, e.g. an older revision built by
    tools/build_ref_lib.sh for an A/B run inside one GPU call."""
    return os.environ.get("ORBHIP_LIBRARY") or os.path.join(_HERE, "liborbhip.so")


_libs = {}


def lib(path=None):
    """Load (once per path) and return the ctypes handle; path=None -> the in-tree liborbhip.so (hipcc, gfx950)."""
    path = os.path.abspath(path or library_path())
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise OrbHipError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(path)
    vp, i32p, ip = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int)
    L.orbhip_version.restype = C.c_char_p
    L.orbhip_runtime_info.argtypes = [C.c_char_p, C.c_int]
    L.orbhip_device_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    L.orbhip_device_free.argtypes = [C.c_int, C.c_void_p]
    L.orbhip_device_upload.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.orbhip_device_download.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.orbhip_device_synchronize.argtypes = [C.c_int]
    L.orbhip_last_error.restype = C.c_char_p
    L.orbhip_create.argtypes = [C.POINTER(vp), C.POINTER(Config)]
    L.orbhip_destroy.argtypes = [vp]
    L.orbhip_destroy.restype = None
    L.orbhip_keypoint_capacity.argtypes = [vp]
    L.orbhip_get_scale_tables.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orbhip_level_size.argtypes = [vp, C.c_int, ip, ip]
    L.orbhip_extract.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, ip]
    L.orbhip_extract_batch.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp]
    L.orbhip_submit_to.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, ip]
    L.orbhip_pyramid_level.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    L.orbhip_extract_device.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    L.orbhip_extract_batch_color.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]
    L.orbhip_extract_device_color.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    L.orbhip_voc_load_text.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int]
    L.orbhip_voc_destroy.argtypes = [vp]
    L.orbhip_voc_destroy.restype = None
    L.orbhip_voc_info.argtypes = [vp, ip, ip, ip, ip, ip, ip]
    L.orbhip_voc_transform_features.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.orbhip_voc_transform.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, ip, vp, vp, vp, ip]
    L.orbhip_compute_bow.argtypes = [vp, vp, C.c_int, C.c_int]
    L.orbhip_fetch_bow.argtypes = [vp, vp, C.c_int, vp, vp, ip, vp, vp, vp, ip]
    L.orbhip_voc_score.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int]
    L.orbhip_voc_score.restype = C.c_double
    L.orbhip_search_by_bow.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp, ip]
    L.orbhip_search_for_triangulation.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                                  vp, C.c_float, C.c_float, vp, vp, C.c_int, C.c_int, C.c_int, vp, ip]
    L.orbhip_search_best_in_window.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]
    L.orbhip_thread_api_ms.argtypes = [C.c_int]
    L.orbhip_thread_api_ms.restype = C.c_double
    L.orbhip_search_by_bow_batch.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_float, C.c_int]
    L.orbhip_search_for_triangulation_batch.argtypes = [C.c_int, vp, C.c_int, vp, C.c_int, C.c_int]
    L.orbhip_search_best_in_window_batch.argtypes = [C.c_int, C.c_int, vp, C.c_int]
    L.orbhip_undistort_points.argtypes = [C.c_int, C.POINTER(Camera), vp, C.c_int, vp]
    L.orbhip_image_bounds.argtypes = [C.c_int, C.POINTER(Camera), C.c_int, C.c_int, C.POINTER(Bounds)]
    L.orbhip_set_camera.argtypes = [vp, C.POINTER(Camera)]
    L.orbhip_get_bounds.argtypes = [vp, C.POINTER(Bounds)]
    L.orbhip_fetch_undistorted.argtypes = [vp, C.c_int, vp, C.c_int]
    L.orbhip_search_for_initialization_bounds.argtypes = [C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(Bounds), vp, vp, C.c_int, C.c_float, C.c_int, ip]
    L.orbhip_search_by_projection_bounds.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int, C.POINTER(Bounds), vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp, ip]
    L.orbhip_search_best_in_window_bounds.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.POINTER(Bounds), vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]
    L.orbhip_set_rectification.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.orbhip_extract_batch_rectify.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp]
    L.orbhip_extract_device_rectify.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    L.orbhip_compute_stereo_from_rgbd.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp, C.c_int]
    L.orbhip_set_stereo_columns.argtypes = [vp, C.c_int, vp, C.c_int]
    L.orbhip_search_by_projection_batch.argtypes = [C.c_int, C.c_int, vp, C.POINTER(Bounds), C.c_int, C.c_float, C.c_int, C.c_int]
    L.orbhip_search_by_projection_frame.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp, ip]
    L.orbhip_search_best_in_window_frame.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]
    L.orbhip_sync.argtypes = [vp]
    L.orbhip_fetch.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp]
    L.orbhip_fetch_matches.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
    L.orbhip_descriptor_distance.argtypes = [vp, vp]
    L.orbhip_hamming_nn.argtypes = [C.c_int, vp, C.c_int, vp, C.c_int64, C.c_int64, vp, vp, vp]
    L.orbhip_hamming_nn_device.argtypes = [vp, vp, C.c_int, vp, C.c_int64, C.c_int64, vp, vp, vp]
    L.orbhip_nn_expanded_size.argtypes = [C.c_int64]; L.orbhip_nn_expanded_size.restype = C.c_size_t
    L.orbhip_nn_expand_device.argtypes = [vp, vp, C.c_int64, vp]
    L.orbhip_hamming_nn_device_expanded.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int64, C.c_int64, vp, vp, vp]
    L.orbhip_search_for_initialization.argtypes = [C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp,
                                                   C.c_int, C.c_float, C.c_int, ip]
    L.orbhip_search_by_projection.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp, ip]
    L.orbhip_predict_scale_table.argtypes = [LEVEL_OF, vp, C.c_int, vp]
    L.orbhip_project_search_bounds.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int, C.POINTER(Bounds), C.POINTER(Projection), vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp, ip, vp]
    L.orbhip_project_search_frame.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(Projection), vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp, ip, vp]
    L.orbhip_project_best_in_window_bounds.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.POINTER(Bounds), vp, C.c_int, C.POINTER(Projection), vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.orbhip_project_best_in_window_batch.argtypes = [C.c_int, C.c_int, vp, C.c_int]
    L.orbhip_project_best_in_window_shared.argtypes = [C.c_int, C.c_int, vp, vp, C.c_int]
    L.orbhip_project_best_in_window_held.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp]
    L.orbhip_compute_stereo_matches.argtypes = [vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, C.c_int]
    L.orbhip_extract_stereo.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_float, C.c_float, vp, vp]
    L.orbhip_profile_enable.argtypes = [vp, C.c_int]
    L.orbhip_profile_num_kernels.argtypes = [vp]
    L.orbhip_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.orbhip_profile_reset.argtypes = [vp]
    L.orbhip_pyramid_cascade_tiles.argtypes = [vp]; L.orbhip_pyramid_cascade_tiles.restype = C.c_int
    L.orbhip_algorithmic_bytes_per_frame.argtypes = [vp]
    L.orbhip_algorithmic_bytes_per_frame.restype = C.c_int64
    L.orbhip_algorithmic_bytes_per_frame_kernel.argtypes = [vp, C.c_int]
    L.orbhip_algorithmic_bytes_per_frame_kernel.restype = C.c_int64
    L.orbhip_debug_blurred_level.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    L.orbhip_debug_candidates.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, ip]
    L.orbhip_pyramid_fetch_all.argtypes = [vp, C.c_int, vp, vp]
    L.orbhip_set_blur_rounding.argtypes = [vp, C.c_int]
    L.orbhip_set_fp_contract.argtypes = [vp, C.c_int]
    L.orbhip_submit.argtypes = [vp, C.c_int, vp, C.c_int, ip]
    L.orbhip_collect.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp]
    L.orbhip_host_alloc.argtypes = [C.c_size_t]
    L.orbhip_host_alloc.restype = vp
    L.orbhip_host_free.argtypes = [vp]
    L.orbhip_host_free.restype = None
    L.orbhip_pool_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.POINTER(Config), C.c_int]
    L.orbhip_pool_destroy.argtypes = [vp]
    L.orbhip_pool_destroy.restype = None
    L.orbhip_pool_num_devices.argtypes = [vp]
    L.orbhip_pool_device_of.argtypes = [vp, C.c_int]
    L.orbhip_pool_keypoint_capacity.argtypes = [vp]
    L.orbhip_pool_numa_node.argtypes = [vp, C.c_int, ip]
    L.orbhip_pool_extract.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp]
    L.orbhip_pool_submit.argtypes = [vp, vp, C.c_int, ip]
    L.orbhip_pool_collect.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp]
    L.orbhip_pool_db_load.argtypes = [vp, vp, C.c_int64]
    L.orbhip_pool_db_shard.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.orbhip_pool_db_shard.restype = None
    L.orbhip_pool_db_query.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.orbhip_reloc_candidates.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, ip]
    _libs[path] = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


OK, ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_UNSUPPORTED = 0, 1, 2, 3, 4      # orbhip_status (include/orbhip.h)


def _check(st, what, L=None):
    if st != 0:
        raise OrbHipError(f"{what} failed (status {st}): {(L or lib()).orbhip_last_error().decode()}")


class ORBextractor:
    """ORB_SLAM2::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) on one MI355X.

    The reference extractor accepts any image size per call; here the HBM layout is fixed at construction
    (width, height, max_batch) — construct one per camera geometry.  `__call__(image)` is operator().
    """
    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, max_batch=1, device=0,
                 stream=None, blur_round_mode=0, library=None, num_streams=1):
        self.L = lib(library)
        self.cfg = Config(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, max_batch, device,
                          C.c_void_p(stream) if stream else None, blur_round_mode, num_streams)
        self.h = C.c_void_p()
        _check(self.L.orbhip_create(C.byref(self.h), C.byref(self.cfg)), "orbhip_create", self.L)
        self.width, self.height, self.nlevels, self.max_batch = width, height, nlevels, max_batch
        self.capacity = self.L.orbhip_keypoint_capacity(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.orbhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- accessors (ORBextractor.h:63-84)
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return float(np.float32(self.cfg.scale_factor))

    def _tables(self):
        n = self.nlevels
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl = np.zeros(n, np.int32)
        _check(self.L.orbhip_get_scale_tables(self.h, _p(sf), _p(isf), _p(s2), _p(is2), _p(fpl)), "orbhip_get_scale_tables", self.L)
        return sf, isf, s2, is2, fpl

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        return self._tables()[4]

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        _check(self.L.orbhip_level_size(self.h, level, C.byref(w), C.byref(h)), "orbhip_level_size", self.L)
        return w.value, h.value

    def SetBlurRounding(self, mode):
        """cv::GaussianBlur's last rounding: 0 = OpenCV generic C++, 1 = the SSE2 column filter of x86-64 builds"""
        _check(self.L.orbhip_set_blur_rounding(self.h, int(mode)), "orbhip_set_blur_rounding", self.L)

    def SetFpContract(self, mode):
        """Pattern rotation of computeOrbDescriptor: 0 = two roundings (-ffp-contract=off), 1 = gcc's fused forms under the reference's own
        flags (-O3 -march=native), DESIGN.md H3"""
        _check(self.L.orbhip_set_fp_contract(self.h, int(mode)), "orbhip_set_fp_contract", self.L)

    # ---- operator()
    def __call__(self, image, mask=None):
        """Returns (keypoints[KEYPOINT_DTYPE], descriptors[N,32] uint8).  `mask` is ignored like in the reference."""
        if image is None or image.size == 0:
            return np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        k, d = self.extract_batch([image])
        return k[0], d[0]

    def extract_batch(self, images):
        n = len(images)
        assert 1 <= n <= self.max_batch
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        for im in imgs:
            assert im.shape == (self.height, self.width), (im.shape, (self.height, self.width))
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        cap = self.capacity
        kps = np.zeros((n, cap), KEYPOINT_DTYPE)
        desc = np.zeros((n, cap, 32), np.uint8)
        nout = np.zeros(n, np.int32)
        _check(self.L.orbhip_extract_batch(self.h, n, ptrs, self.width, _p(kps), _p(desc), cap, _p(nout)), "orbhip_extract_batch", self.L)
        return [kps[f, :nout[f]].copy() for f in range(n)], [desc[f, :nout[f]].copy() for f in range(n)]

    # ---- pipelined host path: up to ring_depth() batches in flight (orbhip_submit / orbhip_collect)
    def submit(self, images, out=None):
        """Stage + upload `images` and enqueue their extraction; returns a ticket for collect().  Pageable images are consumed when this
        returns; pinned ones (pinned_array) are read by DMA until the ticket is collected.  out = (kps, desc, nout) names the result
        buffers up front (orbhip_submit_to): pinned ones are filled by DMA directly and collect() of the ticket returns nout."""
        n = len(images)
        assert 1 <= n <= self.max_batch
        if isinstance(images, np.ndarray) and images.ndim == 3 and images.dtype == np.uint8 and images.shape[1:] == (self.height, self.width) \
                and images.strides[2] == 1 and images.strides[1] == self.width and images.strides[0] >= self.height * self.width:
            # one array of frames: the pointer table is arithmetic (no per-image Python work: at 100 k frames/s a batch of 256 is 2.5 ms)
            imgs = images
            table = (images.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(images.strides[0])).astype(np.uint64)
            ptrs = table.ctypes.data_as(C.c_void_p)
        else:
            imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
            for im in imgs:
                assert im.shape == (self.height, self.width), (im.shape, (self.height, self.width))
            table = None
            ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        t = C.c_int(-1)
        if out is not None:
            # pinned result buffers are written by DMA, n * capacity records each: anything smaller or strided would be a silent host-memory overrun
            k_, d_ = out[0], out[1]
            if not (isinstance(k_, np.ndarray) and k_.dtype == KEYPOINT_DTYPE and k_.ndim == 2 and k_.shape[0] >= n and k_.shape[1] == self.capacity and k_.flags["C_CONTIGUOUS"]):
                raise ValueError(f"submit(out=...): key point buffer must be a C-contiguous [>= {n}, {self.capacity}] array of KEYPOINT_DTYPE")
            if not (isinstance(d_, np.ndarray) and d_.dtype == np.uint8 and d_.ndim == 3 and d_.shape[0] >= n and d_.shape[1:] == (self.capacity, 32) and d_.flags["C_CONTIGUOUS"]):
                raise ValueError(f"submit(out=...): descriptor buffer must be a C-contiguous [>= {n}, {self.capacity}, 32] uint8 array")
            if len(out) > 2 and not (isinstance(out[2], np.ndarray) and out[2].dtype == np.int32 and out[2].size >= n and out[2].flags["C_CONTIGUOUS"]):
                raise ValueError(f"submit(out=...): the count buffer must be a C-contiguous int32 array of >= {n} entries")
            _check(self.L.orbhip_submit_to(self.h, n, ptrs, self.width, _p(out[0]), _p(out[1]), self.capacity, C.byref(t)), "orbhip_submit_to", self.L)
        else:
            _check(self.L.orbhip_submit(self.h, n, ptrs, self.width, C.byref(t)), "orbhip_submit", self.L)
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[t.value] = (n, (imgs, table, out))         # keeps the (possibly pinned) sources and named result buffers alive until collect
        return t.value

    def collect(self, ticket, out=None):
        """Results of a submitted batch (tickets in submission order): ([keypoints], [descriptors]).  out = (kps, desc, nout) buffers to reuse."""
        n, held = self._inflight[ticket]
        cap = self.capacity
        if out is None and held[2] is not None:
            out = held[2]                                          # the buffers named at submit
        kps, desc, nout = out if out is not None else (np.zeros((n, cap), KEYPOINT_DTYPE), np.zeros((n, cap, 32), np.uint8), np.zeros(n, np.int32))
        st = self.L.orbhip_collect(self.h, ticket, _p(kps), _p(desc), cap, _p(nout))
        if st != ERR_INVALID:                                      # every other status retired the ticket on the C side; INVALID (wrong ticket / wrong buffers) left it collectable
            self._inflight.pop(ticket, None)
        _check(st, "orbhip_collect", self.L)
        if out is not None:
            return nout
        return [kps[f, :nout[f]].copy() for f in range(n)], [desc[f, :nout[f]].copy() for f in range(n)]

    def extract_batch_color(self, images, rgb=True):
        """Interleaved 8-bit colour frames [H,W,3|4]; `rgb` is the reference's mbRGB (Tracking.cc:82): True = R first,
        False = B first.  The cvtColor of Tracking::GrabImage* (Tracking.cc:172-198) runs on the device."""
        n = len(images)
        assert 1 <= n <= self.max_batch
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        ch = imgs[0].shape[2]
        for im in imgs:
            assert im.shape == (self.height, self.width, ch), (im.shape, (self.height, self.width, ch))
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        cap = self.capacity
        kps = np.zeros((n, cap), KEYPOINT_DTYPE)
        desc = np.zeros((n, cap, 32), np.uint8)
        nout = np.zeros(n, np.int32)
        _check(self.L.orbhip_extract_batch_color(self.h, n, ptrs, self.width * ch, ch, int(rgb), _p(kps), _p(desc), cap, _p(nout)),
               "orbhip_extract_batch_color", self.L)
        return [kps[f, :nout[f]].copy() for f in range(n)], [desc[f, :nout[f]].copy() for f in range(n)]

    # ---- distorted cameras / rectification
    def set_camera(self, camera):
        """camera = (fx, fy, cx, cy, k1, k2, p1, p2[, k3]) or None.  With k1 != 0 every extraction also produces mvKeysUn
        (Frame::UndistortKeyPoints) and the pipeline matcher works on it inside the undistorted image bounds."""
        cam = None if camera is None else Camera.of(camera)
        _check(self.L.orbhip_set_camera(self.h, None if cam is None else C.byref(cam)), "orbhip_set_camera", self.L)

    def bounds(self):
        """(mnMinX, mnMinY, mnMaxX, mnMaxY) of Frame::ComputeImageBounds for the attached camera"""
        b = Bounds()
        _check(self.L.orbhip_get_bounds(self.h, C.byref(b)), "orbhip_get_bounds", self.L)
        return b.array()

    def fetch_undistorted(self, nimg, counts):
        """mvKeysUn of the last call's first nimg frames (counts = the key point counts orbhip_fetch reported)"""
        cap = self.capacity
        un = np.zeros((nimg, cap), KEYPOINT_DTYPE)
        _check(self.L.orbhip_fetch_undistorted(self.h, nimg, _p(un), cap), "orbhip_fetch_undistorted", self.L)
        return [un[f, :counts[f]].copy() for f in range(nimg)]

    def set_rectification(self, map_x, map_y, src_w, src_h):
        """CV_32FC1 maps [height, width] of cv::initUndistortRectifyMap (stereo_euroc.cc:97-98); None removes them"""
        if map_x is None:
            _check(self.L.orbhip_set_rectification(self.h, None, None, 0, 0), "orbhip_set_rectification", self.L)
            self.src_w = self.src_h = 0
            return
        mx = np.ascontiguousarray(map_x, np.float32); my = np.ascontiguousarray(map_y, np.float32)
        assert mx.shape == (self.height, self.width) and my.shape == mx.shape
        _check(self.L.orbhip_set_rectification(self.h, _p(mx), _p(my), src_w, src_h), "orbhip_set_rectification", self.L)
        self.src_w, self.src_h = src_w, src_h

    def extract_batch_rectify(self, raw_images):
        """RAW (unrectified) gray frames [src_h, src_w]; cv::remap of stereo_euroc.cc:136-137 runs on the device"""
        n = len(raw_images)
        assert 1 <= n <= self.max_batch
        imgs = [np.ascontiguousarray(im, np.uint8) for im in raw_images]
        for im in imgs:
            assert im.shape == (self.src_h, self.src_w), (im.shape, (self.src_h, self.src_w))
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        cap = self.capacity
        kps = np.zeros((n, cap), KEYPOINT_DTYPE); desc = np.zeros((n, cap, 32), np.uint8); nout = np.zeros(n, np.int32)
        _check(self.L.orbhip_extract_batch_rectify(self.h, n, ptrs, self.src_w, _p(kps), _p(desc), cap, _p(nout)), "orbhip_extract_batch_rectify", self.L)
        return [kps[f, :nout[f]].copy() for f in range(n)], [desc[f, :nout[f]].copy() for f in range(n)]

    def extract_device_rectify(self, dptr, nimg, frame_stride, row_stride, match_prev=False, window=100, nnratio=0.9, check_ori=True):
        _check(self.L.orbhip_extract_device_rectify(self.h, nimg, C.c_void_p(dptr), frame_stride, row_stride, int(match_prev), window,
                                                    nnratio, int(check_ori)), "orbhip_extract_device_rectify", self.L)

    # ---- windowed searches on a frame that is still on the device
    def search_by_projection(self, frame, n, queries, query_desc, mode, nnratio=0.8, th_high=100, check_ori=True, use_u_right=False, blocked=None):
        """orbhip_search_by_projection_frame: like the module-level search_by_projection, on frame `frame` of the last call"""
        queries = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE); query_desc = np.ascontiguousarray(query_desc, np.uint8)
        bl = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
        assert bl is None or len(bl) == n
        fq = np.full(max(n, 1), -1, np.int32); nm = C.c_int()
        _check(self.L.orbhip_search_by_projection_frame(self.h, frame, n, int(use_u_right), None if bl is None else _p(bl), _p(queries), _p(query_desc), len(queries),
                                                        mode, nnratio, th_high, int(check_ori), _p(fq), C.byref(nm)), "orbhip_search_by_projection_frame", self.L)
        return nm.value, fq[:n]

    def search_best_in_window(self, frame, n, queries, query_desc, chi2_gate, use_u_right=False):
        queries = np.ascontiguousarray(queries, BEST_QUERY_DTYPE); query_desc = np.ascontiguousarray(query_desc, np.uint8)
        bi = np.full(len(queries), -1, np.int32); bd = np.full(len(queries), 256, np.int32)
        _check(self.L.orbhip_search_best_in_window_frame(self.h, frame, n, int(use_u_right), _p(queries), _p(query_desc), len(queries), int(chi2_gate), _p(bi), _p(bd)),
               "orbhip_search_best_in_window_frame", self.L)
        return bi, bd

    def ComputeStereoFromRGBD(self, depth_maps, depth_factor, mbf):
        """Frame::ComputeStereoFromRGBD for the frames of the last call; depth_maps: float32 or uint16 [H,W] arrays (the conversion of
        Tracking::GrabImageRGBD is folded in).  Returns (mvuRight[nimg, cap], mvDepth[nimg, cap])."""
        n = len(depth_maps)
        dms = [np.ascontiguousarray(d) for d in depth_maps]
        assert all(d.dtype == dms[0].dtype and d.shape == (self.height, self.width) for d in dms) and dms[0].dtype in (np.float32, np.uint16)
        ptrs = (C.c_void_p * n)(*[d.ctypes.data for d in dms])
        cap = self.capacity
        u = np.zeros((n, cap), np.float32); z = np.zeros((n, cap), np.float32)
        _check(self.L.orbhip_compute_stereo_from_rgbd(self.h, n, ptrs, dms[0].strides[0], int(dms[0].dtype == np.uint16), float(depth_factor), float(mbf),
                                                      _p(u), _p(z), cap), "orbhip_compute_stereo_from_rgbd", self.L)
        return u, z

    def set_stereo_columns(self, u_right, frame=0):
        """mvuRight computed on the host (the reference's Frame::ComputeStereoFromRGBD loop) for a frame of the last call: the resident searches read it in HBM."""
        u = np.ascontiguousarray(u_right, np.float32)
        _check(self.L.orbhip_set_stereo_columns(self.h, frame, _p(u), len(u)), "orbhip_set_stereo_columns", self.L)

    def mvImagePyramid(self, level, frame=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        _check(self.L.orbhip_pyramid_level(self.h, frame, level, _p(out), w), "orbhip_pyramid_level", self.L)
        return out

    # ---- device-resident pipeline
    def extract_device(self, dptr, nimg, frame_stride, row_stride, match_prev=False, window=100, nnratio=0.9, check_ori=True):
        _check(self.L.orbhip_extract_device(self.h, nimg, C.c_void_p(dptr), frame_stride, row_stride, int(match_prev), window,
                                            nnratio, int(check_ori)), "orbhip_extract_device", self.L)

    def extract_device_color(self, dptr, nimg, frame_stride, row_stride, channels, rgb=True, match_prev=False, window=100,
                             nnratio=0.9, check_ori=True):
        _check(self.L.orbhip_extract_device_color(self.h, nimg, C.c_void_p(dptr), frame_stride, row_stride, channels, int(rgb),
                                                  int(match_prev), window, nnratio, int(check_ori)), "orbhip_extract_device_color", self.L)

    def sync(self):
        _check(self.L.orbhip_sync(self.h), "orbhip_sync", self.L)

    def fetch(self, nimg):
        cap = self.capacity
        kps = np.zeros((nimg, cap), KEYPOINT_DTYPE)
        desc = np.zeros((nimg, cap, 32), np.uint8)
        nout = np.zeros(nimg, np.int32)
        _check(self.L.orbhip_fetch(self.h, nimg, _p(kps), _p(desc), cap, _p(nout)), "orbhip_fetch", self.L)
        return [kps[f, :nout[f]].copy() for f in range(nimg)], [desc[f, :nout[f]].copy() for f in range(nimg)]

    def fetch_matches(self, nimg):
        cap = self.capacity
        m12 = np.full((nimg, cap), -1, np.int32)
        n1 = np.zeros(nimg, np.int32)
        nm = np.zeros(nimg, np.int32)
        _check(self.L.orbhip_fetch_matches(self.h, nimg, _p(m12), cap, _p(n1), _p(nm)), "orbhip_fetch_matches", self.L)
        return [m12[f, :n1[f]].copy() for f in range(nimg)], nm

    # ---- Frame::ComputeStereoMatches (this extractor = left camera)
    def ComputeStereoMatches(self, right, mbf, mb, nimg=1):
        """mvuRight, mvDepth (float32 [N] per frame) for the frames both extractors processed in their last call."""
        cap = self.capacity
        u = np.zeros((nimg, cap), np.float32)
        d = np.zeros((nimg, cap), np.float32)
        _check(self.L.orbhip_compute_stereo_matches(self.h, right.h, nimg, mbf, mb, _p(u), _p(d), cap), "orbhip_compute_stereo_matches", self.L)
        return u, d

    def extract_stereo(self, left, right, mbf, mb):
        """orbhip_extract_stereo: the stereo pair as one call on this context (max_batch >= 2).
        -> (keys_left, desc_left, keys_right, desc_right, mvuRight[N_left], mvDepth[N_left])"""
        left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
        assert left.shape == (self.height, self.width) and right.shape == left.shape
        cap = self.capacity
        kps = np.zeros((2, cap), KEYPOINT_DTYPE); desc = np.zeros((2, cap, 32), np.uint8); n = np.zeros(2, np.int32)
        u = np.zeros(cap, np.float32); d = np.zeros(cap, np.float32)
        _check(self.L.orbhip_extract_stereo(self.h, _p(left), _p(right), left.strides[0], _p(kps), _p(desc), cap, _p(n), mbf, mb, _p(u), _p(d)), "orbhip_extract_stereo", self.L)
        return kps[0, :n[0]].copy(), desc[0, :n[0]].copy(), kps[1, :n[1]].copy(), desc[1, :n[1]].copy(), u[:n[0]].copy(), d[:n[0]].copy()

    # ---- measurement / stage dumps
    def pyramid_cascade_tiles(self):
        """tiles of the last level owned by k_pyramid_cascade's workgroups; 0 = the level-by-level kernels run (orbhip_pyramid_cascade_tiles)"""
        return int(self.L.orbhip_pyramid_cascade_tiles(self.h))

    def profile_enable(self, on=True):
        _check(self.L.orbhip_profile_enable(self.h, int(on)), "orbhip_profile_enable", self.L)

    def profile_reset(self):
        _check(self.L.orbhip_profile_reset(self.h), "orbhip_profile_reset", self.L)

    def profile(self):
        out = {}
        for k in range(self.L.orbhip_profile_num_kernels(self.h)):
            name, ms, n = C.c_char_p(), C.c_double(), C.c_int64()
            _check(self.L.orbhip_profile_get(self.h, k, C.byref(name), C.byref(ms), C.byref(n)), "orbhip_profile_get", self.L)
            out[name.value.decode()] = dict(index=k, total_ms=ms.value, launches=n.value,
                                            alg_bytes_per_frame=self.L.orbhip_algorithmic_bytes_per_frame_kernel(self.h, k))
        return out

    def algorithmic_bytes_per_frame(self):
        return self.L.orbhip_algorithmic_bytes_per_frame(self.h)

    def blurred_level(self, level, frame=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        _check(self.L.orbhip_debug_blurred_level(self.h, frame, level, _p(out), w), "orbhip_debug_blurred_level", self.L)
        return out

    def candidates(self, level, frame=0):
        n = C.c_int()
        _check(self.L.orbhip_debug_candidates(self.h, frame, level, None, 0, C.byref(n)), "orbhip_debug_candidates", self.L)
        out = np.zeros((max(n.value, 1), 3), np.int32)
        _check(self.L.orbhip_debug_candidates(self.h, frame, level, _p(out), n.value, C.byref(n)), "orbhip_debug_candidates", self.L)
        return out[:n.value].copy()


def device_count(library=None):
    """HIP devices visible to the library (0 = nothing can run; there is no CPU fallback)"""
    return lib(library).orbhip_device_count()


def runtime_info(library=None):
    """One line naming the HIP runtime the library is running on (versions, the file libamdhip64 was mapped from, device 0)."""
    L = lib(library)
    buf = C.create_string_buffer(1024)
    _check(L.orbhip_runtime_info(buf, 1024), "orbhip_runtime_info", L)
    return buf.value.decode()


def mapped_hip_runtimes():
    """Files named libamdhip64* that are mapped into this process (there must be exactly one: a process that mixes the system
    runtime with a framework's bundled copy hands raw device pointers across two runtimes)."""
    seen = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 6 and "libamdhip64" in parts[-1]:
                    seen.add(os.path.realpath(parts[-1]))
    except OSError:
        pass
    return sorted(seen)


class DeviceBuffer:
    """Frames resident in HBM without a framework: orbhip_device_alloc / upload / download / free on the library's own HIP runtime.
    `ptr` is what orbhip_extract_device* take."""

    def __init__(self, nbytes, device=0, library=None):
        self.L, self.device, self.nbytes = lib(library), int(device), int(nbytes)
        p = C.c_void_p()
        _check(self.L.orbhip_device_alloc(self.device, max(self.nbytes, 1), C.byref(p)), "orbhip_device_alloc", self.L)
        self.ptr = p.value

    @classmethod
    def from_array(cls, arr, device=0, library=None):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes, device, library)
        b.upload(arr)
        return b

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        _check(self.L.orbhip_device_upload(self.device, C.c_void_p(self.ptr + offset), _p(arr), arr.nbytes), "orbhip_device_upload", self.L)

    def download(self, shape, dtype=np.uint8, offset=0):
        out = np.empty(shape, dtype)
        assert offset + out.nbytes <= self.nbytes
        _check(self.L.orbhip_device_download(self.device, _p(out), C.c_void_p(self.ptr + offset), out.nbytes), "orbhip_device_download", self.L)
        return out

    def free(self):
        if self.ptr:
            self.L.orbhip_device_free(self.device, C.c_void_p(self.ptr))
            self.ptr = None

    close = free

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_synchronize(device=0, library=None):
    L = lib(library)
    _check(L.orbhip_device_synchronize(int(device)), "orbhip_device_synchronize", L)


def pinned_array(shape, dtype=np.uint8, library=None):
    """A numpy array in pinned host memory (orbhip_host_alloc): images handed over in such arrays and result buffers of this kind are
    moved by DMA directly, without the staging copy.  The memory is released when the array (and every view of it) is gone."""
    L = lib(library)
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dt.itemsize
    ptr = L.orbhip_host_alloc(max(nbytes, 1))
    if not ptr:
        raise OrbHipError("orbhip_host_alloc failed")

    class _Owner:
        def __del__(self, L=L, ptr=ptr):
            L.orbhip_host_free(ptr)
    buf = (C.c_uint8 * max(nbytes, 1)).from_address(ptr)
    arr = np.frombuffer(buf, dtype=np.uint8, count=nbytes).view(dt).reshape(shape)
    buf._owner = _Owner()                                         # the ctypes buffer is the array's base: the owner lives as long as it does
    return arr


class MultiGpuExtractor:
    """One node, G GPUs (orbhip_pool_*): camera c is served by devices[c mod G] — one context, one host thread and one pinned staging
    ring per device, no collective.  Also owns the row-sharded descriptor DB of the relocalisation query (BASELINE.json config 5).
    The reference's analogue is the pair of extractor threads of the stereo Frame constructor (Frame.cc:78-81)."""

    def __init__(self, devices, ncameras, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, blur_round_mode=0, library=None):
        self.L = lib(library)
        self.devices = [int(d) for d in devices]
        self.ncameras, self.width, self.height = ncameras, width, height
        cfg = Config(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, 1, 0, None, blur_round_mode, 1)
        devs = (C.c_int * len(self.devices))(*self.devices)
        self.h = C.c_void_p()
        _check(self.L.orbhip_pool_create(C.byref(self.h), devs, len(self.devices), C.byref(cfg), ncameras), "orbhip_pool_create", self.L)
        self.capacity = self.L.orbhip_pool_keypoint_capacity(self.h)
        self._inflight = {}

    def close(self):
        if getattr(self, "h", None):
            self.L.orbhip_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_of(self, camera):
        return self.L.orbhip_pool_device_of(self.h, camera)

    def numa_node(self, r):
        """(NUMA node of worker r's device or -1, whether the worker thread and its pinned ring were bound to that node)"""
        b = C.c_int(0)
        node = self.L.orbhip_pool_numa_node(self.h, r, C.byref(b))
        return node, bool(b.value)

    def _ptrs(self, images):
        assert len(images) == self.ncameras
        imgs = [None if im is None else np.ascontiguousarray(im, np.uint8) for im in images]
        for im in imgs:
            assert im is None or im.shape == (self.height, self.width)
        return imgs, (C.c_void_p * self.ncameras)(*[None if im is None else im.ctypes.data for im in imgs])

    def _outputs(self):
        return np.zeros((self.ncameras, self.capacity), KEYPOINT_DTYPE), np.zeros((self.ncameras, self.capacity, 32), np.uint8), np.zeros(self.ncameras, np.int32)

    def extract(self, images):
        """One frame per camera (None = no frame this round) -> ([keypoints], [descriptors]) per camera"""
        imgs, ptrs = self._ptrs(images)
        kps, desc, nout = self._outputs()
        _check(self.L.orbhip_pool_extract(self.h, ptrs, self.width, _p(kps), _p(desc), self.capacity, _p(nout)), "orbhip_pool_extract", self.L)
        return [kps[c, :nout[c]].copy() for c in range(self.ncameras)], [desc[c, :nout[c]].copy() for c in range(self.ncameras)]

    def submit(self, images):
        imgs, ptrs = self._ptrs(images)
        t = C.c_int(-1)
        _check(self.L.orbhip_pool_submit(self.h, ptrs, self.width, C.byref(t)), "orbhip_pool_submit", self.L)
        self._inflight[t.value] = imgs
        return t.value

    def collect(self, ticket, out=None):
        self._inflight.pop(ticket)
        kps, desc, nout = out if out is not None else self._outputs()
        _check(self.L.orbhip_pool_collect(self.h, ticket, _p(kps), _p(desc), self.capacity, _p(nout)), "orbhip_pool_collect", self.L)
        if out is not None:
            return nout
        return [kps[c, :nout[c]].copy() for c in range(self.ncameras)], [desc[c, :nout[c]].copy() for c in range(self.ncameras)]

    # ---- descriptor DB sharded by contiguous row ranges (config 5)
    def db_load(self, db):
        db = np.ascontiguousarray(db, np.uint8)
        assert db.ndim == 2 and db.shape[1] == 32
        _check(self.L.orbhip_pool_db_load(self.h, _p(db), len(db)), "orbhip_pool_db_load", self.L)
        self.ndb = len(db)

    def db_shard(self, r):
        lo, hi = C.c_int64(), C.c_int64()
        self.L.orbhip_pool_db_shard(self.h, r, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def db_query(self, q):
        q = np.ascontiguousarray(q, np.uint8)
        bi = np.zeros(len(q), np.int64); bd = np.zeros(len(q), np.int32); sd = np.zeros(len(q), np.int32)
        _check(self.L.orbhip_pool_db_query(self.h, _p(q), len(q), _p(bi), _p(bd), _p(sd)), "orbhip_pool_db_query", self.L)
        return bi, bd, sd


def reloc_candidates(best_idx, best_dist, second_dist, row_keyframe, nkf, th_dist=50, ratio=0.75, top_k=10, library=None):
    """Top-k key frames by accepted nearest-neighbour votes (orbhip_reloc_candidates): the relocalisation candidate source that stands
    where Tracking::Relocalization calls DetectRelocalizationCandidates (Tracking.cc:1344-1348).  -> (keyframe ids, votes)"""
    L = lib(library)
    bi = np.ascontiguousarray(best_idx, np.int64); bd = np.ascontiguousarray(best_dist, np.int32); sd = np.ascontiguousarray(second_dist, np.int32)
    rk = np.ascontiguousarray(row_keyframe, np.int32)
    kf = np.zeros(max(top_k, 1), np.int32); votes = np.zeros(max(top_k, 1), np.int32); n = C.c_int(0)
    _check(L.orbhip_reloc_candidates(_p(bi), _p(bd), _p(sd), len(bi), _p(rk), len(rk), nkf, th_dist, ratio, top_k, _p(kf), _p(votes), C.byref(n)),
           "orbhip_reloc_candidates", L)
    return kf[:n.value].copy(), votes[:n.value].copy()


class ORBmatcher:
    """The Frame-to-Frame / Hamming part of ORB_SLAM2::ORBmatcher(nnratio=0.6, checkOri=true) on one MI355X."""
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30

    def __init__(self, nnratio=0.6, checkOri=True, device=0, library=None):
        self.L = lib(library)
        self.nnratio, self.checkOri, self.device = float(nnratio), bool(checkOri), device

    @staticmethod
    def DescriptorDistance(a, b, library=None):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        assert a.size == 32 and b.size == 32
        return lib(library).orbhip_descriptor_distance(_p(a), _p(b))

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, im_w, im_h, vbPrevMatched=None, windowSize=10, bounds=None):
        """F1/F2 are passed as (mvKeysUn, mDescriptors) + the image bounds (bounds = (mnMinX, mnMinY, mnMaxX, mnMaxY) of a distorted
        camera, default the whole im_w x im_h image).  Returns (nmatches, vnMatches12, vbPrevMatched)."""
        kps1 = np.ascontiguousarray(kps1)
        kps2 = np.ascontiguousarray(kps2)
        desc1 = np.ascontiguousarray(desc1, np.uint8)
        desc2 = np.ascontiguousarray(desc2, np.uint8)
        if vbPrevMatched is None:
            vbPrevMatched = np.stack([kps1["x"], kps1["y"]], axis=1)
        prev = np.ascontiguousarray(vbPrevMatched, np.float32).copy()
        m12 = np.full(max(len(kps1), 1), -1, np.int32)
        nm = C.c_int()
        if bounds is None:
            _check(self.L.orbhip_search_for_initialization(self.device, _p(kps1), _p(desc1), len(kps1), _p(kps2), _p(desc2), len(kps2),
                                                           im_w, im_h, _p(prev), _p(m12), windowSize, self.nnratio, int(self.checkOri),
                                                           C.byref(nm)), "orbhip_search_for_initialization", self.L)
        else:
            b = Bounds.of(bounds)
            _check(self.L.orbhip_search_for_initialization_bounds(self.device, _p(kps1), _p(desc1), len(kps1), _p(kps2), _p(desc2), len(kps2),
                                                                  C.byref(b), _p(prev), _p(m12), windowSize, self.nnratio, int(self.checkOri),
                                                                  C.byref(nm)), "orbhip_search_for_initialization_bounds", self.L)
        return nm.value, m12[:len(kps1)], prev


PROJ_QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("radius", "<f4"), ("ur", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                             ("blocks", "<i4"), ("angle", "<f4")])          # orbhip_proj_query


def search_by_projection(kps, desc, im_w, im_h, queries, query_desc, mode, nnratio=0.8, th_high=100, check_ori=True, u_right=None,
                         blocked=None, device=0, library=None, bounds=None):
    """Search loop of ORBmatcher::SearchByProjection(Frame, MapPoints) (mode 0) / (Current, Last) (mode 1) on flat data.
    Returns (nmatches, feature_query int32[n])."""
    L = lib(library)
    kps = np.ascontiguousarray(kps)
    desc = np.ascontiguousarray(desc, np.uint8)
    queries = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE)
    query_desc = np.ascontiguousarray(query_desc, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    bl = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
    fq = np.full(max(len(kps), 1), -1, np.int32)
    nm = C.c_int()
    if bounds is None:
        _check(L.orbhip_search_by_projection(device, _p(kps), _p(desc), None if ur is None else _p(ur), None if bl is None else _p(bl), len(kps), im_w, im_h,
                                             _p(queries), _p(query_desc), len(queries), mode, nnratio, th_high, int(check_ori), _p(fq), C.byref(nm)),
               "orbhip_search_by_projection", L)
    else:
        b = Bounds.of(bounds)
        _check(L.orbhip_search_by_projection_bounds(device, _p(kps), _p(desc), None if ur is None else _p(ur), None if bl is None else _p(bl), len(kps), C.byref(b),
                                                    _p(queries), _p(query_desc), len(queries), mode, nnratio, th_high, int(check_ori), _p(fq), C.byref(nm)),
               "orbhip_search_by_projection_bounds", L)
    return nm.value, fq[:len(kps)]


class ProjSlot(C.Structure):        # orbhip_proj_slot
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("u_right", C.c_void_p), ("blocked", C.c_void_p), ("n", C.c_int32),
                ("queries", C.c_void_p), ("query_desc", C.c_void_p), ("nq", C.c_int32), ("feature_query", C.c_void_p), ("nmatches", C.c_int32)]


def search_by_projection_batch(frames, im_w, im_h, mode, nnratio=0.8, th_high=100, check_ori=True, device=0, library=None, bounds=None):
    """orbhip_search_by_projection_batch: frames = [(kps, desc, queries, query_desc[, u_right[, blocked]])] — one camera slot each, searched in
    one pass.  Returns [(nmatches, feature_query)] per slot, identical to per-slot search_by_projection calls."""
    L = lib(library)
    keep, slots = [], (ProjSlot * len(frames))()
    for s, fr in enumerate(frames):
        kps, desc, queries, qdesc = fr[:4]
        ur = fr[4] if len(fr) > 4 else None
        bl = fr[5] if len(fr) > 5 else None
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
        queries = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        ur = None if ur is None else np.ascontiguousarray(ur, np.float32)
        bl = None if bl is None else np.ascontiguousarray(bl, np.uint8)
        fq = np.full(max(len(kps), 1), -1, np.int32)
        keep.append((kps, desc, queries, qdesc, ur, bl, fq))
        slots[s] = ProjSlot(kps.ctypes.data, desc.ctypes.data, None if ur is None else ur.ctypes.data, None if bl is None else bl.ctypes.data, len(kps),
                            queries.ctypes.data, qdesc.ctypes.data, len(queries), fq.ctypes.data, 0)
    b = Bounds.of(bounds, im_w, im_h)
    _check(L.orbhip_search_by_projection_batch(device, len(frames), slots, C.byref(b), mode, nnratio, th_high, int(check_ori)), "orbhip_search_by_projection_batch", L)
    return [(slots[s].nmatches, keep[s][6][:len(keep[s][0])]) for s in range(len(frames))]


def hamming_nn(q, db, device=0, index_base=0, library=None):
    """Brute-force NN of every query descriptor over db: (best_idx int64, best_dist, second_dist)."""
    q = np.ascontiguousarray(q, np.uint8)
    db = np.ascontiguousarray(db, np.uint8)
    bi = np.zeros(len(q), np.int64)
    bd = np.zeros(len(q), np.int32)
    sd = np.zeros(len(q), np.int32)
    L = lib(library)
    _check(L.orbhip_hamming_nn(device, _p(q), len(q), _p(db), len(db), index_base, _p(bi), _p(bd), _p(sd)), "orbhip_hamming_nn", L)
    return bi, bd, sd


def nn_expanded_size(ndb, library=None):
    """bytes of the expanded form of an ndb-row descriptor database (orbhip_nn_expanded_size)"""
    return int(lib(library).orbhip_nn_expanded_size(int(ndb)))


def nn_expand_device(stream, d_db, ndb, d_expanded, library=None):
    L = lib(library)
    _check(L.orbhip_nn_expand_device(C.c_void_p(stream) if stream else None, C.c_void_p(d_db), int(ndb), C.c_void_p(d_expanded)), "orbhip_nn_expand_device", L)


def hamming_nn_device_expanded(stream, d_q, nq, d_db, d_expanded, ndb, d_best_idx, d_best_dist, d_second, index_base=0, library=None):
    L = lib(library)
    _check(L.orbhip_hamming_nn_device_expanded(C.c_void_p(stream) if stream else None, C.c_void_p(d_q), nq, C.c_void_p(d_db), C.c_void_p(d_expanded), ndb,
                                               index_base, C.c_void_p(d_best_idx), C.c_void_p(d_best_dist), C.c_void_p(d_second)),
           "orbhip_hamming_nn_device_expanded", L)


def hamming_nn_device(stream, d_q, nq, d_db, ndb, d_best_idx, d_best_dist, d_second, index_base=0, library=None):
    L = lib(library)
    _check(L.orbhip_hamming_nn_device(C.c_void_p(stream) if stream else None, C.c_void_p(d_q), nq, C.c_void_p(d_db), ndb,
                                      index_base, C.c_void_p(d_best_idx), C.c_void_p(d_best_dist), C.c_void_p(d_second)),
           "orbhip_hamming_nn_device", L)


class ORBVocabulary:
    """Mirror of ORB_SLAM2::ORBVocabulary (include/ORBVocabulary.h:31-32 = DBoW2::TemplatedVocabulary<FORB>): text loader,
    transform (Frame::ComputeBoW, Frame.cc:395-402) and score on the GPU library.  BowVector / FeatureVector come back as
    flat arrays in std::map order: (word ids, values) and (node ids, offsets, feature indices)."""

    def __init__(self, path=None, device=0, library=None):
        self.L_ = lib(library)
        self.h = C.c_void_p()
        self.device = device
        if path is not None and not self.loadFromTextFile(path):
            raise OrbHipError(self.L_.orbhip_last_error().decode())

    def loadFromTextFile(self, path):
        self.close()
        st = self.L_.orbhip_voc_load_text(C.byref(self.h), str(path).encode(), self.device)
        if st != 0:
            self.h = C.c_void_p()
            return False
        v = [C.c_int() for _ in range(6)]
        _check(self.L_.orbhip_voc_info(self.h, *[C.byref(x) for x in v]), "orbhip_voc_info", self.L_)
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = [x.value for x in v]
        return True

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L_.orbhip_voc_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return self.nwords

    def transform_features(self, desc, levelsup):
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        w = np.zeros(n, np.uint32); v = np.zeros(n, np.float64); nd = np.zeros(n, np.uint32)
        _check(self.L_.orbhip_voc_transform_features(self.h, _p(desc), n, levelsup, _p(w), _p(v), _p(nd)), "orbhip_voc_transform_features", self.L_)
        return w, v, nd

    @staticmethod
    def _outputs(n):
        return (np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64), np.zeros(max(n, 1), np.uint32),
                np.zeros(n + 2, np.int32), np.zeros(max(n, 1), np.uint32), C.c_int(0), C.c_int(0))

    @staticmethod
    def _trim(bid, bval, fnode, foff, ffeat, nb, nf):
        m, q = nb.value, nf.value
        return bid[:m].copy(), bval[:m].copy(), fnode[:q].copy(), foff[:q + 1].copy(), ffeat[:foff[q]].copy()

    def transform(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        bid, bval, fnode, foff, ffeat, nb, nf = self._outputs(n)
        _check(self.L_.orbhip_voc_transform(self.h, _p(desc), n, levelsup, _p(bid), _p(bval), C.byref(nb), _p(fnode), _p(foff), _p(ffeat), C.byref(nf)),
               "orbhip_voc_transform", self.L_)
        return self._trim(bid, bval, fnode, foff, ffeat, nb, nf)

    def compute_bow(self, extractor, nimg, levelsup=4):
        """Frame::ComputeBoW for the frames of `extractor`'s last call, descriptors read in place on the device."""
        _check(self.L_.orbhip_compute_bow(extractor.h, self.h, nimg, levelsup), "orbhip_compute_bow", self.L_)

    def fetch_bow(self, extractor, frame):
        n = extractor.capacity
        bid, bval, fnode, foff, ffeat, nb, nf = self._outputs(n)
        _check(self.L_.orbhip_fetch_bow(extractor.h, self.h, frame, _p(bid), _p(bval), C.byref(nb), _p(fnode), _p(foff), _p(ffeat), C.byref(nf)),
               "orbhip_fetch_bow", self.L_)
        return self._trim(bid, bval, fnode, foff, ffeat, nb, nf)

    def score(self, id1, val1, id2, val2):
        id1 = np.ascontiguousarray(id1, np.uint32); id2 = np.ascontiguousarray(id2, np.uint32)
        val1 = np.ascontiguousarray(val1, np.float64); val2 = np.ascontiguousarray(val2, np.float64)
        return self.L_.orbhip_voc_score(self.h, _p(id1), _p(val1), len(id1), _p(id2), _p(val2), len(id2))


def search_by_bow(mode, desc1, angle1, valid1, fv1, desc2, angle2, valid2, fv2, nnratio=0.7, check_ori=True, device=0, library=None):
    """ORBmatcher(nnratio, check_ori).SearchByBoW on flat data: mode 0 = (KeyFrame, Frame) ORBmatcher.cc:159-288, mode 1 =
    (KeyFrame, KeyFrame) :522-655.  fv = (node ids, offsets, feature indices).  -> (nmatches, match12[n1])"""
    L = lib(library)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    angle1 = np.ascontiguousarray(angle1, np.float32); angle2 = np.ascontiguousarray(angle2, np.float32)
    valid1 = np.ascontiguousarray(valid1, np.uint8)
    valid2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    f1 = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    f2 = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    m12 = np.full(len(desc1), -1, np.int32)
    nm = C.c_int(0)
    _check(L.orbhip_search_by_bow(device, mode, _p(desc1), _p(angle1), _p(valid1), len(desc1), _p(f1[0]), _p(f1[1]), _p(f1[2]), len(f1[0]),
                                  _p(desc2), _p(angle2), None if valid2 is None else _p(valid2), len(desc2), _p(f2[0]), _p(f2[1]), _p(f2[2]), len(f2[0]),
                                  nnratio, int(check_ori), _p(m12), C.byref(nm)), "orbhip_search_by_bow", L)
    return nm.value, m12


def search_for_triangulation(desc1, kps1, has_mp1, stereo1, fv1, desc2, kps2, has_mp2, stereo2, fv2, F12, ex, ey, scale_factors2, level_sigma2_2,
                             only_stereo=False, check_ori=True, device=0, library=None):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:657-823) on flat data; kps = KEYPOINT arrays (mvKeysUn).  -> (nmatches, match12[n1])"""
    L = lib(library)
    kp4 = lambda k: np.ascontiguousarray(np.stack([k["x"], k["y"], k["angle"], k["octave"].astype(np.float32)], axis=1), np.float32)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    k1, k2 = kp4(kps1), kp4(kps2)
    a = [np.ascontiguousarray(v, np.uint8) for v in (has_mp1, stereo1, has_mp2, stereo2)]
    f1 = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    f2 = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sc = np.ascontiguousarray(scale_factors2, np.float32); sg = np.ascontiguousarray(level_sigma2_2, np.float32)
    m12 = np.full(len(desc1), -1, np.int32)
    nm = C.c_int(0)
    _check(L.orbhip_search_for_triangulation(device, _p(desc1), _p(k1), _p(a[0]), _p(a[1]), len(desc1), _p(f1[0]), _p(f1[1]), _p(f1[2]), len(f1[0]),
                                             _p(desc2), _p(k2), _p(a[2]), _p(a[3]), len(desc2), _p(f2[0]), _p(f2[1]), _p(f2[2]), len(f2[0]),
                                             _p(F), float(ex), float(ey), _p(sc), _p(sg), len(sc), int(only_stereo), int(check_ori), _p(m12), C.byref(nm)),
           "orbhip_search_for_triangulation", L)
    return nm.value, m12


class BowSide(C.Structure):          # orbhip_bow_side
    _fields_ = [("desc", C.c_void_p), ("angle", C.c_void_p), ("valid", C.c_void_p), ("n", C.c_int32), ("fv_node", C.c_void_p), ("fv_off", C.c_void_p), ("fv_feat", C.c_void_p), ("nfv", C.c_int32)]


class BowPair(C.Structure):          # orbhip_bow_pair
    _fields_ = [("side1", C.POINTER(BowSide)), ("side2", C.POINTER(BowSide)), ("match12", C.c_void_p), ("nmatches", C.c_int32)]


class TriSide(C.Structure):          # orbhip_tri_side
    _fields_ = [("desc", C.c_void_p), ("kp", C.c_void_p), ("has_mp", C.c_void_p), ("stereo", C.c_void_p), ("n", C.c_int32), ("fv_node", C.c_void_p), ("fv_off", C.c_void_p), ("fv_feat", C.c_void_p),
                ("nfv", C.c_int32), ("scale_factors", C.c_void_p), ("level_sigma2", C.c_void_p), ("nlevels", C.c_int32)]


class TriPair(C.Structure):          # orbhip_tri_pair
    _fields_ = [("kf2", C.POINTER(TriSide)), ("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float), ("match12", C.c_void_p), ("nmatches", C.c_int32)]


class BestSlot(C.Structure):         # orbhip_best_slot
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("u_right", C.c_void_p), ("n", C.c_int32), ("bounds", Bounds), ("inv_level_sigma2", C.c_void_p), ("nlevels", C.c_int32),
                ("queries", C.c_void_p), ("query_desc", C.c_void_p), ("nq", C.c_int32), ("best_idx", C.c_void_p), ("best_dist", C.c_void_p)]


def _fv3(fv):
    return [np.ascontiguousarray(fv[0], np.uint32), np.ascontiguousarray(fv[1], np.int32), np.ascontiguousarray(fv[2], np.uint32)]


def search_by_bow_batch(mode, pairs, nnratio=0.7, check_ori=True, device=0, library=None):
    """orbhip_search_by_bow_batch.  pairs = [(side1, side2), ...], a side = dict(desc=, angle=, valid= (or None), fv=(nodes, offsets, features)); a side OBJECT named by several
    pairs (the current frame of Relocalization) is uploaded once.  -> [(nmatches, match12), ...]"""
    L = lib(library)
    keep, sides = [], {}

    def side(d):
        if id(d) not in sides:
            desc = np.ascontiguousarray(d["desc"], np.uint8); ang = np.ascontiguousarray(d["angle"], np.float32)
            val = None if d.get("valid") is None else np.ascontiguousarray(d["valid"], np.uint8)
            f = _fv3(d["fv"])
            keep.extend([desc, ang, val] + f)
            sides[id(d)] = BowSide(_p(desc).value if len(desc) else None, _p(ang).value if len(ang) else None, None if val is None else _p(val).value, len(desc), _p(f[0]).value, _p(f[1]).value, _p(f[2]).value, len(f[0]))
        return sides[id(d)]
    arr = (BowPair * max(len(pairs), 1))()
    outs = []
    for k, (a, b) in enumerate(pairs):
        sa, sb = side(a), side(b)
        m12 = np.full(sa.n, -1, np.int32); outs.append(m12)
        arr[k].side1 = C.pointer(sa); arr[k].side2 = C.pointer(sb); arr[k].match12 = _p(m12).value if len(m12) else None; arr[k].nmatches = 0
    _check(L.orbhip_search_by_bow_batch(device, mode, len(pairs), arr, nnratio, int(check_ori)), "orbhip_search_by_bow_batch", L)
    return [(int(arr[k].nmatches), outs[k]) for k in range(len(pairs))]


def search_for_triangulation_batch(kf1, neighbours, only_stereo=False, check_ori=False, device=0, library=None):
    """orbhip_search_for_triangulation_batch.  kf1 / a neighbour's "kf" = dict(desc=, kps= (KEYPOINT array, mvKeysUn), has_mp=, stereo=, fv=, scale_factors=, level_sigma2=);
    neighbours = [dict(kf=..., F12=, ex=, ey=), ...].  Every pair is searched with kf1's has_mp as given.  -> [(nmatches, match12[n1]), ...]"""
    L = lib(library)
    keep = []
    kp4 = lambda k: np.ascontiguousarray(np.stack([k["x"], k["y"], k["angle"], k["octave"].astype(np.float32)], axis=1), np.float32) if len(k) else np.zeros((0, 4), np.float32)

    def side(d):
        desc = np.ascontiguousarray(d["desc"], np.uint8); kp = kp4(d["kps"]); hm = np.ascontiguousarray(d["has_mp"], np.uint8); st = np.ascontiguousarray(d["stereo"], np.uint8)
        f = _fv3(d["fv"]); sc = np.ascontiguousarray(d["scale_factors"], np.float32); sg = np.ascontiguousarray(d["level_sigma2"], np.float32)
        keep.extend([desc, kp, hm, st, sc, sg] + f)
        return TriSide(_p(desc).value, _p(kp).value, _p(hm).value, _p(st).value, len(desc), _p(f[0]).value, _p(f[1]).value, _p(f[2]).value, len(f[0]), _p(sc).value, _p(sg).value, len(sc))
    s1 = side(kf1)
    arr = (TriPair * max(len(neighbours), 1))()
    outs, s2s = [], []
    for k, nb in enumerate(neighbours):
        s2 = side(nb["kf"]); s2s.append(s2)
        m12 = np.full(s1.n, -1, np.int32); outs.append(m12)
        arr[k].kf2 = C.pointer(s2)
        for i, v in enumerate(np.ascontiguousarray(nb["F12"], np.float32).reshape(9)):
            arr[k].F12[i] = float(v)
        arr[k].ex = float(nb["ex"]); arr[k].ey = float(nb["ey"]); arr[k].match12 = _p(m12).value if len(m12) else None
    _check(L.orbhip_search_for_triangulation_batch(device, C.byref(s1), len(neighbours), arr, int(only_stereo), int(check_ori)), "orbhip_search_for_triangulation_batch", L)
    return [(int(arr[k].nmatches), outs[k]) for k in range(len(neighbours))]


BEST_QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("radius", "<f4"), ("ur", "<f4"), ("level", "<i4")])


def search_best_in_window_batch(slots, chi2_gate, device=0, library=None):
    """orbhip_search_best_in_window_batch.  slots = [dict(kps=, desc=, u_right= (or None), bounds=(min_x, min_y, max_x, max_y), inv_level_sigma2=, queries=, qdesc=), ...]
    -> [(best_idx, best_dist), ...]"""
    L = lib(library)
    keep, outs = [], []
    arr = (BestSlot * max(len(slots), 1))()
    for k, sl in enumerate(slots):
        kps = np.ascontiguousarray(sl["kps"], KEYPOINT_DTYPE); desc = np.ascontiguousarray(sl["desc"], np.uint8)
        q = np.ascontiguousarray(sl["queries"], BEST_QUERY_DTYPE); qd = np.ascontiguousarray(sl["qdesc"], np.uint8)
        inv = np.ascontiguousarray(sl["inv_level_sigma2"], np.float32)
        ur = None if sl.get("u_right") is None else np.ascontiguousarray(sl["u_right"], np.float32)
        bi = np.full(len(q), -1, np.int32); bd = np.full(len(q), 256, np.int32)
        keep.extend([kps, desc, q, qd, inv, ur]); outs.append((bi, bd))
        a = arr[k]
        a.kps = _p(kps).value if len(kps) else None; a.desc = _p(desc).value if len(desc) else None; a.u_right = None if ur is None else _p(ur).value; a.n = len(kps)
        a.bounds = Bounds.of(sl["bounds"]); a.inv_level_sigma2 = _p(inv).value; a.nlevels = len(inv)
        a.queries = _p(q).value if len(q) else None; a.query_desc = _p(qd).value if len(qd) else None; a.nq = len(q)
        a.best_idx = _p(bi).value if len(q) else None; a.best_dist = _p(bd).value if len(q) else None
    _check(L.orbhip_search_best_in_window_batch(device, len(slots), arr, int(chi2_gate)), "orbhip_search_best_in_window_batch", L)
    return outs


def search_best_in_window(kps, desc, im_w, im_h, inv_level_sigma2, queries, qdesc, chi2_gate, u_right=None, device=0, library=None, bounds=None):
    """Candidate loop of ORBmatcher::Fuse / SearchBySim3 on flat data (include/orbhip.h) -> (best_idx[nq], best_dist[nq])"""
    L = lib(library)
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    queries = np.ascontiguousarray(queries, BEST_QUERY_DTYPE); qdesc = np.ascontiguousarray(qdesc, np.uint8)
    inv = np.ascontiguousarray(inv_level_sigma2, np.float32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    bi = np.full(len(queries), -1, np.int32); bd = np.full(len(queries), 256, np.int32)
    if bounds is None:
        _check(L.orbhip_search_best_in_window(device, _p(kps), _p(desc), None if ur is None else _p(ur), len(kps), im_w, im_h, _p(inv), len(inv),
                                              _p(queries), _p(qdesc), len(queries), int(chi2_gate), _p(bi), _p(bd)), "orbhip_search_best_in_window", L)
    else:
        b = Bounds.of(bounds)
        _check(L.orbhip_search_best_in_window_bounds(device, _p(kps), _p(desc), None if ur is None else _p(ur), len(kps), C.byref(b), _p(inv), len(inv),
                                                     _p(queries), _p(qdesc), len(queries), int(chi2_gate), _p(bi), _p(bd)), "orbhip_search_best_in_window_bounds", L)
    return bi, bd


def predict_scale_table(log_scale_factor, nlevels, level_of=None, library=None):
    """orbhip_predict_scale_table -> level_ratio[16].  level_of(ratio) = the caller's own MapPoint::PredictScale expression (the C++ drop-in passes its
    translation unit's; the tests pass the oracle's, which calls this machine's logf); default: the expression in numpy float32 - whose log is numpy's own
    and may round differently from libm's at a level boundary."""
    L = lib(library)
    lsf = np.float32(log_scale_factor)

    def default(ratio):
        with np.errstate(all="ignore"):
            v = np.ceil(np.log(np.float32(ratio)) / lsf)
        return 0 if not np.isfinite(v) else int(min(max(int(v), 0), nlevels - 1))
    fn = level_of or default
    out = np.zeros(MAX_PROJ_LEVELS, np.float32)
    _check(L.orbhip_predict_scale_table(LEVEL_OF(lambda ratio, _: int(fn(ratio))), None, nlevels, _p(out)), "orbhip_predict_scale_table", L)
    return out


def make_projection(kind, R, t, fx, fy, cx, cy, bounds, th, scale_factors, level_ratio, Ow=(0, 0, 0), bf=0.0, R2=None, t2=None, gemm_mode=0, forward=False, backward=False):
    P = Projection()
    P.kind = kind; P.gemm_mode = gemm_mode
    P.R = (C.c_float * 9)(*np.asarray(R, np.float32).reshape(9)); P.t = (C.c_float * 3)(*np.asarray(t, np.float32).reshape(3))
    if R2 is not None:
        P.R2 = (C.c_float * 9)(*np.asarray(R2, np.float32).reshape(9)); P.t2 = (C.c_float * 3)(*np.asarray(t2, np.float32).reshape(3))
    P.Ow = (C.c_float * 3)(*np.asarray(Ow, np.float32).reshape(3))
    P.fx, P.fy, P.cx, P.cy, P.bf = fx, fy, cx, cy, bf
    P.min_x, P.min_y, P.max_x, P.max_y = bounds
    P.th = th; P.forward = int(forward); P.backward = int(backward)
    sf = np.asarray(scale_factors, np.float32)
    P.nlevels = len(sf)
    P.scale_factors = (C.c_float * MAX_PROJ_LEVELS)(*(list(sf) + [0.0] * (MAX_PROJ_LEVELS - len(sf))))
    P.level_ratio = (C.c_float * MAX_PROJ_LEVELS)(*np.asarray(level_ratio, np.float32))
    return P


def project_search(kps, desc, bounds, proj, points, point_desc, nnratio=0.9, th_high=100, check_ori=True, u_right=None, blocked=None, device=0, library=None):
    """orbhip_project_search_bounds -> (nmatches, feature_query[n], queries_out[np]): the projection of the pose-guided SearchByProjection overloads on the device"""
    L = lib(library)
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    points = np.ascontiguousarray(points, MAP_POINT_DTYPE); point_desc = np.ascontiguousarray(point_desc, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    bl = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
    fq = np.full(len(kps), -1, np.int32); nm = C.c_int(0); qo = np.zeros(len(points), PROJ_QUERY_DTYPE)
    b = Bounds.of(bounds)
    _check(L.orbhip_project_search_bounds(device, _p(kps), _p(desc), None if ur is None else _p(ur), None if bl is None else _p(bl), len(kps), C.byref(b), C.byref(proj),
                                          _p(points), _p(point_desc), len(points), nnratio, th_high, int(check_ori), _p(fq), C.byref(nm), _p(qo)), "orbhip_project_search_bounds", L)
    return nm.value, fq, qo


def project_best_in_window(kps, desc, bounds, inv_level_sigma2, proj, points, point_desc, chi2_gate, u_right=None, device=0, library=None):
    """orbhip_project_best_in_window_bounds -> (best_idx[np], best_dist[np], queries_out[np]): Fuse x2 / SearchBySim3 with the projection on the device"""
    L = lib(library)
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    points = np.ascontiguousarray(points, MAP_POINT_DTYPE); point_desc = np.ascontiguousarray(point_desc, np.uint8)
    inv = np.ascontiguousarray(inv_level_sigma2, np.float32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    bi = np.full(len(points), -1, np.int32); bd = np.full(len(points), 256, np.int32); qo = np.zeros(len(points), BEST_QUERY_DTYPE)
    b = Bounds.of(bounds)
    _check(L.orbhip_project_best_in_window_bounds(device, _p(kps), _p(desc), None if ur is None else _p(ur), len(kps), C.byref(b), _p(inv), len(inv), C.byref(proj),
                                                  _p(points), _p(point_desc), len(points), int(chi2_gate), _p(bi), _p(bd), _p(qo)), "orbhip_project_best_in_window_bounds", L)
    return bi, bd, qo


class ProjectBestSlot(C.Structure):  # orbhip_project_best_slot
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("u_right", C.c_void_p), ("n", C.c_int32), ("bounds", Bounds), ("inv_level_sigma2", C.c_void_p), ("nlevels", C.c_int32),
                ("proj", C.POINTER(Projection)), ("points", C.c_void_p), ("point_desc", C.c_void_p), ("np", C.c_int32), ("best_idx", C.c_void_p), ("best_dist", C.c_void_p)]


def project_best_in_window_batch(slots, chi2_gate, device=0, library=None):
    """orbhip_project_best_in_window_batch.  slots = [dict(kps=, desc=, u_right= (or None), bounds=, inv_level_sigma2=, proj=, points=, pdesc=), ...] -> [(best_idx, best_dist), ...]"""
    L = lib(library)
    keep, outs = [], []
    arr = (ProjectBestSlot * max(len(slots), 1))()
    for k, sl in enumerate(slots):
        kps = np.ascontiguousarray(sl["kps"], KEYPOINT_DTYPE); desc = np.ascontiguousarray(sl["desc"], np.uint8)
        pts = np.ascontiguousarray(sl["points"], MAP_POINT_DTYPE); pd = np.ascontiguousarray(sl["pdesc"], np.uint8)
        inv = np.ascontiguousarray(sl["inv_level_sigma2"], np.float32)
        ur = None if sl.get("u_right") is None else np.ascontiguousarray(sl["u_right"], np.float32)
        bi = np.full(len(pts), -1, np.int32); bd = np.full(len(pts), 256, np.int32)
        keep.extend([kps, desc, pts, pd, inv, ur, sl["proj"]]); outs.append((bi, bd))
        a = arr[k]
        a.kps = _p(kps).value if len(kps) else None; a.desc = _p(desc).value if len(desc) else None; a.u_right = None if ur is None else _p(ur).value; a.n = len(kps)
        a.bounds = Bounds.of(sl["bounds"]); a.inv_level_sigma2 = _p(inv).value; a.nlevels = len(inv)
        a.proj = C.pointer(sl["proj"]); a.points = _p(pts).value if len(pts) else None; a.point_desc = _p(pd).value if len(pd) else None; a.np = len(pts)
        a.best_idx = _p(bi).value if len(pts) else None; a.best_dist = _p(bd).value if len(pts) else None
    _check(L.orbhip_project_best_in_window_batch(device, len(slots), arr, int(chi2_gate)), "orbhip_project_best_in_window_batch", L)
    return outs


def project_best_in_window_shared(slots, points, pdesc, skip, chi2_gate, device=0, library=None):
    """orbhip_project_best_in_window_shared: ONE set of points (uploaded once) offered to every slot; skip[k] bit s set = point k is not searched in slot s.
    slots = [dict(kps=, desc=, u_right= (or None), bounds=, inv_level_sigma2=, proj=), ...] -> [(best_idx, best_dist), ...]"""
    L = lib(library)
    pts = np.ascontiguousarray(points, MAP_POINT_DTYPE); pd = np.ascontiguousarray(pdesc, np.uint8)
    sk = None if skip is None else np.ascontiguousarray(skip, np.uint64)
    keep, outs = [], []
    arr = (ProjectBestSlot * max(len(slots), 1))()
    for k, sl in enumerate(slots):
        kps = np.ascontiguousarray(sl["kps"], KEYPOINT_DTYPE); desc = np.ascontiguousarray(sl["desc"], np.uint8)
        inv = np.ascontiguousarray(sl["inv_level_sigma2"], np.float32)
        ur = None if sl.get("u_right") is None else np.ascontiguousarray(sl["u_right"], np.float32)
        bi = np.full(len(pts), -1, np.int32); bd = np.full(len(pts), 256, np.int32)
        keep.extend([kps, desc, inv, ur, sl["proj"]]); outs.append((bi, bd))
        a = arr[k]
        a.kps = _p(kps).value if len(kps) else None; a.desc = _p(desc).value if len(desc) else None; a.u_right = None if ur is None else _p(ur).value; a.n = len(kps)
        a.bounds = Bounds.of(sl["bounds"]); a.inv_level_sigma2 = _p(inv).value; a.nlevels = len(inv)
        a.proj = C.pointer(sl["proj"]); a.points = _p(pts).value if len(pts) else None; a.point_desc = _p(pd).value if len(pd) else None; a.np = len(pts)
        a.best_idx = _p(bi).value if len(pts) else None; a.best_dist = _p(bd).value if len(pts) else None
    _check(L.orbhip_project_best_in_window_shared(device, len(slots), arr, None if sk is None else _p(sk), int(chi2_gate)), "orbhip_project_best_in_window_shared", L)
    return outs


def project_best_in_window_held(slot, proj, points, pdesc, chi2_gate, device=0, library=None, check=True):
    """orbhip_project_best_in_window_held: slot `slot` of this thread's last project_best_in_window_shared call searched again with other points.
    check=False: returns (status, best_idx, best_dist) instead of raising"""
    L = lib(library)
    pts = np.ascontiguousarray(points, MAP_POINT_DTYPE); pd = np.ascontiguousarray(pdesc, np.uint8)
    bi = np.full(len(pts), -1, np.int32); bd = np.full(len(pts), 256, np.int32)
    st = L.orbhip_project_best_in_window_held(device, int(slot), C.byref(proj), _p(pts), _p(pd), len(pts), int(chi2_gate), _p(bi), _p(bd))
    if not check:
        return st, bi, bd
    _check(st, "orbhip_project_best_in_window_held", L)
    return bi, bd


def undistort_points(camera, xy, device=0, library=None):
    """cv::undistortPoints(xy, xy, K, D, Mat(), K) for [n,2] float points on the device (Frame.cc:421)"""
    L = lib(library)
    cam = Camera.of(camera)
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    _check(L.orbhip_undistort_points(device, C.byref(cam), _p(xy), len(xy), _p(out)), "orbhip_undistort_points", L)
    return out


def image_bounds(camera, im_w, im_h, device=0, library=None):
    """Frame::ComputeImageBounds (Frame.cc:436-464) -> (mnMinX, mnMinY, mnMaxX, mnMaxY)"""
    L = lib(library)
    cam = Camera.of(camera)
    b = Bounds()
    _check(L.orbhip_image_bounds(device, C.byref(cam), im_w, im_h, C.byref(b)), "orbhip_image_bounds", L)
    return b.array()
