"""orb_slam2_amd — MI355X-native ORB front-end (extract + match) behind the ORB_SLAM2 class signatures.

Only what the hot path needs: csrc/ (hand-written HIP kernels for gfx950 + the C ABI of include/orbhip.h),
orbhip.py (ctypes mirror of ORB_SLAM2::ORBextractor / ORBmatcher), sharding.py (frames / DB shards across the GPUs
of a node, no collective) and synth.py (seeded synthetic frames).  No CPU fallback lives in this package.
"""
from .orbhip import (DeviceBuffer, device_synchronize, mapped_hip_runtimes, runtime_info, search_by_projection_batch, device_count, MultiGpuExtractor, pinned_array, reloc_candidates, BEST_QUERY_DTYPE, KEYPOINT_DTYPE, PROJ_QUERY_DTYPE, ORBextractor, ORBmatcher, ORBVocabulary, OrbHipError, hamming_nn, hamming_nn_device, hamming_nn_device_expanded, nn_expand_device, nn_expanded_size, lib,
                     library_path, search_by_bow, search_by_bow_batch, search_for_triangulation_batch, search_best_in_window_batch, search_best_in_window, search_by_projection, search_for_triangulation, undistort_points, image_bounds)

__all__ = ["DeviceBuffer", "device_synchronize", "mapped_hip_runtimes", "runtime_info", "search_by_projection_batch", "device_count", "MultiGpuExtractor", "pinned_array", "reloc_candidates", "KEYPOINT_DTYPE", "PROJ_QUERY_DTYPE", "search_by_projection", "search_by_bow", "search_by_bow_batch", "search_for_triangulation_batch", "search_best_in_window_batch", "search_for_triangulation", "search_best_in_window", "BEST_QUERY_DTYPE", "ORBextractor", "ORBmatcher", "ORBVocabulary", "OrbHipError", "hamming_nn", "hamming_nn_device", "hamming_nn_device_expanded", "nn_expand_device", "nn_expanded_size", "lib",
           "library_path", "undistort_points", "image_bounds"]
