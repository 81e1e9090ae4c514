/* orbhip.h — C ABI of the MI355X-native ORB front-end (liborbhip.so, gfx950 only).
 *
 * This is the drop-in boundary for the ONE hot path of raulmur/ORB_SLAM2 that this repository replaces:
 * src/ORBextractor.cc + the Hamming / Frame-to-Frame part of src/ORBmatcher.cc.  The reference has no FFI
 * layer (it is a single C++ library); the C++ classes ORB_SLAM2::ORBextractor / ORB_SLAM2::ORBmatcher in
 * include/ORBextractor.h / include/ORBmatcher.h keep the reference's signatures and forward to the entry
 * points below (see INTEGRATION.md for the binding a maintainer adds).  Plain C types only: pointers and
 * sizes, int status returns, caller-allocated outputs, one opaque context per extractor instance.
 *
 * Threading: one context is used by one thread at a time (like one ORBextractor instance,
 * Frame.cc:78-81 runs the left and right instances on two threads -> use two contexts); the stateless
 * entry points (orbhip_descriptor_distance, orbhip_hamming_nn, orbhip_search_for_initialization) are
 * re-entrant (Tracking / LocalMapping / LoopClosing threads all call the matcher).
 *
 * Error behaviour: the reference's operator() returns void and silently returns on an empty image
 * (ORBextractor.cc:1046-1047).  Here every call returns an orbhip_status; an empty image yields ORBHIP_OK
 * with *n_out = 0.  There is NO CPU fallback: without a usable HIP device orbhip_create fails with
 * ORBHIP_ERR_HIP and orbhip_last_error() says why.
 */
#ifndef ORBHIP_H
#define ORBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbhip_ctx orbhip_ctx;

typedef enum {
    ORBHIP_OK = 0,
    ORBHIP_ERR_INVALID = 1,     /* bad argument */
    ORBHIP_ERR_HIP = 2,         /* HIP runtime error (no device, launch failure, ...) */
    ORBHIP_ERR_CAPACITY = 3,    /* caller buffer too small; *n_out still reports the needed size */
    ORBHIP_ERR_UNSUPPORTED = 4  /* configuration outside the supported envelope */
} orbhip_status;

/* == cv::KeyPoint memory layout (pt.x, pt.y, size, angle, response, octave, class_id), 28 bytes */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbhip_keypoint;

typedef struct {
    /* ORBextractor constructor arguments, ORBextractor.h:52-53 / ORBextractor.cc:410-413 */
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;      /* any integer: used clamped to [0, 255], as cv::FAST does with the threshold the reference passes it (ORBextractor.cc:809-816) */
    int32_t min_th_fast;
    /* geometry of the frames this context processes (the reference re-allocates per call; here the pyramid,
       candidate and keypoint buffers are laid out once in HBM for a fixed size) */
    int32_t width, height;
    int32_t max_batch;        /* frames (independent cameras) processed per batched call, >= 1 */
    int32_t device;           /* HIP device ordinal */
    void* stream;             /* hipStream_t to run on; NULL = the context creates its own stream */
    int32_t blur_round_mode;  /* 0 = OpenCV generic C++ rounding (default), 1 = x86 SSE2 build rounding; DESIGN.md */
    int32_t num_streams;      /* >1: a batched call is split into this many groups of camera slots, each enqueued on its own HIP
                                 stream so latency-bound kernels of one group overlap throughput kernels of another; 0/1 = one stream */
} orbhip_config;

const char* orbhip_version(void);
/* The stateless matcher entry points keep a per-thread device scratch and pinned mirror (grow-only, no hipMalloc per call).  A worker thread
   that exits returns them automatically; orbhip_thread_release() does it on demand for the calling thread. */
void orbhip_thread_release(void);
/* Wall time, in milliseconds, the CALLING thread has spent inside this library's extraction and matcher entry points since the last reset (reset != 0
   clears it): lets a measurement of a caller - an ORBmatcher member around one library call - say how much of its time is the library's. */
double orbhip_thread_api_ms(int reset);
/* HIP devices this process can use (0 = none: nothing in this library can run, there is no CPU fallback) */
int orbhip_device_count(void);
/* last error message of the calling thread ("" if none) */
const char* orbhip_last_error(void);
/* Which HIP runtime this library is running on, as one line of text: runtime / driver version, the file libamdhip64 was mapped from
   (dladdr), device 0's name, ISA and compute-unit count.  Truncated to cap bytes (NUL-terminated).  Touches the device: on a box whose
   GPU cannot be used it fails with ORBHIP_ERR_HIP — before any kernel of this library has run. */
orbhip_status orbhip_runtime_info(char* buf, int cap);
/* Device memory for callers that keep frames resident in HBM (orbhip_extract_device*) but do not link the HIP runtime themselves:
   hipMalloc / hipFree / blocking hipMemcpy on `device`, and hipDeviceSynchronize.  (A caller that already owns device memory — from its
   own hipMalloc or another framework on the SAME HIP runtime — passes those pointers instead.) */
orbhip_status orbhip_device_alloc(int device, size_t bytes, void** out);
orbhip_status orbhip_device_free(int device, void* p);
orbhip_status orbhip_device_upload(int device, void* dst, const void* src_host, size_t bytes);
orbhip_status orbhip_device_download(int device, void* dst_host, const void* src, size_t bytes);
orbhip_status orbhip_device_synchronize(int device);

/* -------- ORBextractor (ORBextractor.h:45-111) ------------------------------------------------------- */
orbhip_status orbhip_create(orbhip_ctx** out, const orbhip_config* cfg);            /* ORBextractor::ORBextractor */
void orbhip_destroy(orbhip_ctx* ctx);                                               /* ~ORBextractor */
int orbhip_keypoint_capacity(const orbhip_ctx* ctx);  /* upper bound of keypoints per frame: sum_l max(N_l+3, 4*nIni_l) */
/* GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
   (ORBextractor.h:63-84) and mnFeaturesPerLevel; each output may be NULL, each holds nlevels entries */
orbhip_status orbhip_get_scale_tables(const orbhip_ctx* ctx, float* scale_factors, float* inv_scale_factors,
                                      float* level_sigma2, float* inv_level_sigma2, int32_t* features_per_level);
orbhip_status orbhip_level_size(const orbhip_ctx* ctx, int level, int* w, int* h);
/* blur_round_mode of the configuration, changeable between calls (ORBextractor::SetBlurRounding) */
orbhip_status orbhip_set_blur_rounding(orbhip_ctx* ctx, int mode);
/* The pattern rotation of computeOrbDescriptor (ORBextractor.cc:118-120): 0 (default) = two roundings per expression, the form of a build
   with -ffp-contract=off; 1 = the fused forms gcc emits with the reference's own flags (CMakeLists.txt:11-14, -O3 -march=native on an
   FMA-capable host): fma(x, b, y*a) and fma(x, a, -(y*b)).  The two differ in about one descriptor bit per few hundred frames (H3). */
orbhip_status orbhip_set_fp_contract(orbhip_ctx* ctx, int mode);

/* ORBextractor::operator() (ORBextractor.h:59-61, ORBextractor.cc:1043-1105): host image in, host keypoints +
   descriptors out, synchronous.  Keypoints are level-major, quadtree-list order within a level; descriptor
   row i belongs to keypoint i.  img == NULL or an empty image -> *n_out = 0 (silent return of the reference). */
orbhip_status orbhip_extract(orbhip_ctx* ctx, const uint8_t* img, int stride_bytes,
                             orbhip_keypoint* kps, uint8_t* desc /* cap x 32 */, int cap, int* n_out);
/* the same for nimg <= max_batch frames in one pass; outputs are [nimg][cap] */
orbhip_status orbhip_extract_batch(orbhip_ctx* ctx, int nimg, const uint8_t* const* imgs, int stride_bytes,
                                   orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out /* nimg */);
/* Colour frames: Tracking::GrabImageStereo/RGBD/Monocular (Tracking.cc:172-198, 217-229, 248-260) run
   cv::cvtColor(RGB2GRAY | BGR2GRAY | RGBA2GRAY | BGRA2GRAY) on the CPU before the extractor is called; these entry
   points take the interleaved 8-bit colour frame instead and convert on the device (OpenCV 3.2 fixed point:
   gray = (4899 R + 9617 G + 1868 B + 8192) >> 14, alpha ignored) into the context's level-0 plane.
   channels = 3 or 4; rgb_order != 0 means R is the first channel (mbRGB, Tracking.cc:82), 0 means B first. */
orbhip_status orbhip_extract_batch_color(orbhip_ctx* ctx, int nimg, const uint8_t* const* imgs, int stride_bytes,
                                         int channels, int rgb_order,
                                         orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out /* nimg */);
/* mvImagePyramid[level] of frame `frame` of the last call (ORBextractor.h:85; read by Frame.cc:473,563-580) */
orbhip_status orbhip_pyramid_level(orbhip_ctx* ctx, int frame, int level, uint8_t* dst, int dst_stride);
/* every level of that frame in one go (nlevels async copies, ONE synchronisation): dst[l] / dst_stride[l] per level; what the
   drop-in class calls the first time a caller touches mvImagePyramid after an extraction */
orbhip_status orbhip_pyramid_fetch_all(orbhip_ctx* ctx, int frame, uint8_t* const* dst, const int* dst_stride);

/* -------- pipelined host-buffer path (Frame::ExtractORB callers, Frame.cc:247-253) ---------------------- */
/* Every real caller hands over host images and wants host results.  orbhip_extract_batch splits a batch into chunks and
   overlaps, on three HIP streams, the upload of chunk k+1 with the kernels of chunk k and the download of chunk k-1
   (pinned staging mirrors; pageable -> pinned gathers are spread over a few helper threads).  orbhip_submit /
   orbhip_collect expose the same machinery across calls: up to orbhip_ring_depth() batches in flight per context, so the
   upload of batch t+1 also overlaps the tail of batch t.  Tickets are collected in submission order; a submit on a full
   ring fails with ORBHIP_ERR_INVALID.  Caller buffers that are pinned (orbhip_host_alloc, hipHostMalloc, hipHostRegister)
   are read / written by DMA directly, without the staging copy.  Results are bit-identical to orbhip_extract_batch. */
orbhip_status orbhip_submit(orbhip_ctx* ctx, int nimg, const uint8_t* const* imgs, int stride_bytes, int* ticket);
orbhip_status orbhip_collect(orbhip_ctx* ctx, int ticket, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out /* nimg */);
/* orbhip_submit with the result buffers named up front ([nimg][cap] key points, [nimg][cap][32] descriptors): when they are pinned the
   downloads go straight into them by DMA, and orbhip_collect of the ticket - called with the SAME buffers - only waits and fills n_out
   (no host copy: at 100 k frames/s the copy out of the staging mirrors is 12 GB/s of memcpy).  Pageable buffers behave like orbhip_submit. */
orbhip_status orbhip_submit_to(orbhip_ctx* ctx, int nimg, const uint8_t* const* imgs, int stride_bytes, orbhip_keypoint* kps, uint8_t* desc, int cap, int* ticket);
int orbhip_ring_depth(void);
/* pinned host memory for callers that do not link the HIP runtime themselves (hipHostMalloc / hipHostFree) */
void* orbhip_host_alloc(size_t bytes);
void orbhip_host_free(void* p);

/* -------- one node, G GPUs: cameras and descriptor-DB shards (BASELINE.json configs 4 and 5) --------
   The reference's only concurrency on this path is the two extractor threads of the stereo Frame constructor
   (Frame.cc:78-81).  A pool generalises that: one context + one host thread + pinned staging ring per device, camera c is
   served by devices[c mod G], no device ever talks to another (no collective).  The descriptor DB of the relocalisation
   query is split by contiguous row range, shard r resident on devices[r]; a query is broadcast by G independent copies,
   every device scans its shard and the G partial answers per query are merged on the host with the matcher's rule
   (strict '<': the lowest global index wins ties; second = second smallest of the union). */
typedef struct orbhip_pool orbhip_pool;
/* cfg->device and cfg->max_batch are ignored (per-device batch = ceil(ncameras / ndevices)); devices may repeat an ordinal
   (several contexts on one GPU) — that is how the CPU test-suite and 1-GPU boxes exercise the N > 1 path */
orbhip_status orbhip_pool_create(orbhip_pool** out, const int* devices, int ndevices, const orbhip_config* cfg, int ncameras);
void orbhip_pool_destroy(orbhip_pool* pool);
int orbhip_pool_num_devices(const orbhip_pool* pool);
int orbhip_pool_device_of(const orbhip_pool* pool, int camera);          /* devices[camera mod G], -1 if out of range */
int orbhip_pool_keypoint_capacity(const orbhip_pool* pool);
/* NUMA placement of worker r: the node its device hangs off (/sys/bus/pci/devices/<bus id>/numa_node; -1 = unknown) and, in *bound (may be
   NULL), whether the worker thread - and with it the pinned staging ring it allocates - was bound to that node's CPUs.  ORBHIP_POOL_NUMA=0
   disables the binding. */
int orbhip_pool_numa_node(const orbhip_pool* pool, int r, int* bound);
/* one frame per camera (imgs[c] == NULL: that camera has no frame this round, n_out[c] = 0); outputs are [ncameras][cap].
   orbhip_pool_extract = orbhip_pool_submit + orbhip_pool_collect; with submit / collect a caller keeps up to
   orbhip_ring_depth() rounds in flight (tickets collected in order).  A round that only SOME devices could take (a HIP failure on one
   device) still gets its ticket — the parts the other devices took must be collected in order like any other — and
   orbhip_pool_collect of that ticket delivers the healthy cameras and returns the refusing device's status; a round no device took
   fails in orbhip_pool_submit itself, without a ticket.  Either way the pool stays usable. */
orbhip_status orbhip_pool_extract(orbhip_pool* pool, const uint8_t* const* imgs, int stride_bytes,
                                  orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out /* ncameras */);
orbhip_status orbhip_pool_submit(orbhip_pool* pool, const uint8_t* const* imgs, int stride_bytes, int* ticket);
orbhip_status orbhip_pool_collect(orbhip_pool* pool, int ticket, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* descriptor DB: rows [lo_r, hi_r) of db go to devices[r] and stay there; ndb rows of 32 bytes.  A shard of 32 K rows or more is also kept in the expanded
   form the FP4 scan multiplies (orbhip_nn_expand_device: four times the bytes) when the device has the memory; ORBHIP_POOL_DB_EXPAND=0: never.  Same answers. */
orbhip_status orbhip_pool_db_load(orbhip_pool* pool, const uint8_t* db, int64_t ndb);
void orbhip_pool_db_shard(const orbhip_pool* pool, int r, int64_t* lo, int64_t* hi);
orbhip_status orbhip_pool_db_query(orbhip_pool* pool, const uint8_t* q, int nq, int64_t* best_idx, int32_t* best_dist, int32_t* second_dist);
/* relocalisation candidates from the brute-force DB (stands where Tracking::Relocalization calls
   KeyFrameDatabase::DetectRelocalizationCandidates, Tracking.cc:1344-1348, KeyFrameDatabase.cc:199-309): every query descriptor
   votes for the key frame that owns its nearest DB row if best <= th_dist and best < ratio * second; row_keyframe[i] = key frame
   id of DB row i (ascending rows of one key frame need not be contiguous), nkf = number of key frames.  Returns the top_k key
   frames by votes (ties: lower key frame id first; key frames without votes are not reported): kf_out / votes_out hold top_k
   entries, *nout the number filled.  Host arithmetic over the answer of orbhip_hamming_nn / orbhip_pool_db_query. */
orbhip_status orbhip_reloc_candidates(const int64_t* best_idx, const int32_t* best_dist, const int32_t* second_dist, int nq,
                                      const int32_t* row_keyframe, int64_t ndb, int nkf, int th_dist, float ratio,
                                      int top_k, int32_t* kf_out, int32_t* votes_out, int* nout);

/* -------- device-resident pipeline (inputs already in HBM, results stay in HBM) ----------------------- */
/* d_imgs: device pointer to nimg frames, frame f at d_imgs + f*frame_stride, rows row_stride bytes apart.
   Asynchronous on the context's stream.  If match_prev != 0 each frame (camera slot) f is additionally
   matched against the frame the same slot processed in the previous call with
   ORBmatcher(nnratio, check_ori).SearchForInitialization(F_prev, F_cur, prev = F_prev keypoints, window)
   (ORBmatcher.cc:405-520; the unit of work of BASELINE.json's metric). */
orbhip_status orbhip_extract_device(orbhip_ctx* ctx, int nimg, const uint8_t* d_imgs, size_t frame_stride,
                                    int row_stride, int match_prev, int window, float nnratio, int check_ori);
/* the same for interleaved colour frames resident in HBM (see orbhip_extract_batch_color) */
orbhip_status orbhip_extract_device_color(orbhip_ctx* ctx, int nimg, const uint8_t* d_imgs, size_t frame_stride,
                                          int row_stride, int channels, int rgb_order,
                                          int match_prev, int window, float nnratio, int check_ori);
orbhip_status orbhip_sync(orbhip_ctx* ctx);
/* copy results of the last orbhip_extract_device call to the host (synchronises) */
orbhip_status orbhip_fetch(orbhip_ctx* ctx, int nimg, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* matches of the last call with match_prev != 0: matches12 is [nimg][cap1] (index into the current frame's
   keypoints per previous-frame keypoint, -1 = none), nmatches[nimg]; n1_out[nimg] = previous-frame keypoint count */
orbhip_status orbhip_fetch_matches(orbhip_ctx* ctx, int nimg, int32_t* matches12, int cap1, int32_t* n1_out,
                                   int32_t* nmatches);

/* -------- ORBmatcher (ORBmatcher.h:37-102) ------------------------------------------------------------ */
/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:1647-1663): 256-bit Hamming distance of two 32-byte descriptors.
   Scalar host helper (one pair is not GPU work); the batched forms below are the HIP path. */
int orbhip_descriptor_distance(const uint8_t* a, const uint8_t* b);

/* Brute-force nearest neighbour (BASELINE.json config 5; matcher idiom ORBmatcher.cc:102-114,447-456):
   for every query descriptor the best (strict '<', lowest index wins ties) and second-best distance over db.
   Host buffers, synchronous.  db_index_base is added to reported indices (shards of one DB across GPUs). */
orbhip_status orbhip_hamming_nn(int device, const uint8_t* q, int nq, const uint8_t* db, int64_t ndb,
                                int64_t db_index_base, int64_t* best_idx, int32_t* best_dist, int32_t* second_dist);
/* same with device-resident buffers on a given stream (NULL = default stream); asynchronous */
orbhip_status orbhip_hamming_nn_device(void* stream, const uint8_t* d_q, int nq, const uint8_t* d_db, int64_t ndb,
                                       int64_t db_index_base, int64_t* d_best_idx, int32_t* d_best_dist,
                                       int32_t* d_second_dist);
/* A database that is queried many times (a map's key frame descriptors: BASELINE.json configs[5]) can be EXPANDED once: orbhip_nn_expanded_size(ndb) bytes
   (128 per row, rounded up to tiles of 32 rows; 16-byte aligned) filled by orbhip_nn_expand_device hold every descriptor bit as the +-1 the matrix cores
   multiply, in the scan's own tile layout - a query then stages tiles by LDS-DMA instead of expanding each row again for every 512 queries.  Same answers as
   orbhip_hamming_nn_device bit for bit (d_db is still read for databases too small for the matrix-core scan).  Asynchronous on `stream`. */
size_t orbhip_nn_expanded_size(int64_t ndb);
orbhip_status orbhip_nn_expand_device(void* stream, const uint8_t* d_db, int64_t ndb, uint8_t* d_expanded);
orbhip_status orbhip_hamming_nn_device_expanded(void* stream, const uint8_t* d_q, int nq, const uint8_t* d_db, const uint8_t* d_expanded, int64_t ndb,
                                                int64_t db_index_base, int64_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist);

/* ORBmatcher(nnratio, check_ori).SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, window)
   (ORBmatcher.h:69, ORBmatcher.cc:405-520) on host buffers.  The Frame members it reads are passed flat:
   mvKeysUn / mDescriptors of both frames and the image bounds (mnMinX = mnMinY = 0, mnMaxX = im_w,
   mnMaxY = im_h: undistorted input, Frame.cc:455-463).  prev_matched (n1 x 2 floats, x then y) is updated in
   place exactly like vbPrevMatched; matches12 has n1 entries.  Returns the match count in *nmatches. */
orbhip_status orbhip_search_for_initialization(int device,
                                               const orbhip_keypoint* kps1, const uint8_t* desc1, int n1,
                                               const orbhip_keypoint* kps2, const uint8_t* desc2, int n2,
                                               int im_w, int im_h, float* prev_matched, int32_t* matches12,
                                               int window, float nnratio, int check_ori, int* nmatches);

/* -------- projection-guided matchers --------------------------------------------------- */
/* One query per map point that passed the caller's own filters (frustum / isBad / projection inside the image):
   the caller keeps Map, MapPoint and pose types and flattens what the search loop reads. */
typedef struct {
    float x, y;          /* projected position: pMP->mTrackProjX/Y (ORBmatcher.cc:69) or u, v (ORBmatcher.cc:1369-1370) */
    float radius;        /* window radius already multiplied by the scale factor: r*F.mvScaleFactors[level] / th*mvScaleFactors[octave] */
    float ur;            /* projected right coordinate: mTrackProjXR / u - mbf*invzc; gated against mvuRight of stereo features */
    int32_t min_level;   /* level arguments of Frame::GetFeaturesInArea (Frame.cc:327) */
    int32_t max_level;
    int32_t blocks;      /* pMP->Observations() > 0: once assigned to a feature, later queries skip that feature */
    float angle;         /* mode 1 only: LastFrame.mvKeysUn[i].angle */
} orbhip_proj_query;

/* Search loop of
     mode 0: ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*>&, th)            ORBmatcher.h:50, ORBmatcher.cc:45-129
     mode 1: ORBmatcher::SearchByProjection(Frame &Current, const Frame &Last, th, bMono)      ORBmatcher.h:54, ORBmatcher.cc:1328-1470
   on one frame given flat: kps / desc = F.mvKeysUn / F.mDescriptors (n), u_right = F.mvuRight or NULL, blocked[i] =
   (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) or NULL, image bounds as for SearchForInitialization.
   feature_query[i] receives the index of the query whose map point the reference would leave in F.mvpMapPoints[i];
   -1 = untouched, -2 = claimed during the call and then removed by the rotation-consistency check (the reference leaves NULL there,
   ORBmatcher.cc:1452-1466, whatever the feature held before); *nmatches is the function's return value.  th_high = TH_HIGH (100).  Host buffers, synchronous. */
orbhip_status orbhip_search_by_projection(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right,
                                          const uint8_t* blocked, int n, int im_w, int im_h,
                                          const orbhip_proj_query* queries, const uint8_t* query_desc, int nq,
                                          int mode, float nnratio, int th_high, int check_ori,
                                          int32_t* feature_query, int* nmatches);

/* -------- Frame::ComputeStereoMatches (Frame.h:94-95, Frame.cc:466-640) ------------------------------- */
/* For nimg stereo pairs: slot f of `left` against slot f of `right`, using the keypoints, descriptors and image
   pyramids both contexts still hold in HBM from their LAST extract call (the reference reads mvKeys/mvKeysRight,
   mDescriptors/mDescriptorsRight and both extractors' mvImagePyramid).  mbf / mb as in Frame.h:104-107.
   u_right / depth are [nimg][cap] (mvuRight / mvDepth, -1 = no match).  Synchronous.  The contexts must share device,
   image size, level count and scale factor. */
orbhip_status orbhip_compute_stereo_matches(orbhip_ctx* left, orbhip_ctx* right, int nimg, float mbf, float mb,
                                            float* u_right, float* depth, int cap);

/* The stereo pair as ONE call: what the stereo Frame constructor does with its two extractor threads and ComputeStereoMatches (Frame.cc:78-90) on a
   single context created with max_batch >= 2 (2 is best: the whole result block then travels in one copy).  Both images are uploaded together and
   run through one launch chain, the stereo matcher (left = slot 0, right = slot 1) is queued behind it; kps / desc are [2][cap] (left, right),
   n_out[2]; u_right / depth hold cap entries for the LEFT key points (-1 = no match).  mb is passed explicitly.  Afterwards the
   context's frame 0 is the left image: orbhip_search_by_projection_frame / orbhip_search_best_in_window_frame(ctx, 0, n_out[0], use_u_right = 1, ...),
   orbhip_compute_bow and orbhip_fetch_undistorted work on it as after a single-image call.  Results are identical to two orbhip_extract calls +
   orbhip_compute_stereo_matches. */
orbhip_status orbhip_extract_stereo(orbhip_ctx* ctx, const uint8_t* img_left, const uint8_t* img_right, int stride_bytes,
                                    orbhip_keypoint* kps, uint8_t* desc /* 2 x cap x 32 */, int cap, int* n_out /* 2 */,
                                    float mbf, float mb, float* u_right, float* depth);

/* -------- Frame::ComputeStereoFromRGBD (Frame.h:97-98, Frame.cc:643-665) ------------------------------ */
/* RGB-D sensors: for the first nimg frames of the context's LAST extract call, d = imDepth(v, u) at every key point (mvKeys, the
   distorted position, coordinates truncated like cv::Mat::at<float>(float, float)); d > 0 -> mvDepth = d, mvuRight = mvKeysUn.x - mbf / d,
   else both -1.  mvKeysUn comes from the attached camera (orbhip_set_camera; none = mvKeys).  The depth maps are width x height like
   the images: depth_type 0 = CV_32F, 1 = CV_16U (the TUM png files).  The imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) of
   Tracking::GrabImageRGBD (Tracking.cc:226-227) is folded in under the reference's own condition — applied iff
   |depth_factor - 1| > 1e-5 or the map is not CV_32F: d = (float)raw * depth_factor.  Host maps (one pointer per frame), synchronous;
   u_right / depth are [nimg][cap]. */
orbhip_status orbhip_compute_stereo_from_rgbd(orbhip_ctx* ctx, int nimg, const void* const* depth_maps, int stride_bytes,
                                              int depth_type, float depth_factor, float mbf, float* u_right, float* depth, int cap);

/* One frame at a time, depth map in host memory (Tracking::GrabImageRGBD's case): the reference's own N-sample loop (Frame.cc:648-664) is cheaper than moving
   the 1.2 MB map, and the frame that is still on the device only needs its result - mvuRight, N floats - for the right-coordinate test of the resident
   searches (use_u_right of orbhip_search_by_projection_frame / _best_in_window_frame; ORBmatcher.cc:1418-1424, 96-101).  Copies u_right[0..n) before it returns,
   uploads on the context's stream (nothing waits); replaces the columns a stereo / RGB-D step of the same extraction left. */
orbhip_status orbhip_set_stereo_columns(orbhip_ctx* ctx, int frame, const float* u_right, int n);

/* The candidate loop of ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (ORBmatcher.h:75, ORBmatcher.cc:825-972), of its
   Sim3 overload (ORBmatcher.h:78, :974-1100: the same loop without the chi-square gate) and of the two passes of
   ORBmatcher::SearchBySim3 (ORBmatcher.h:71-72, ORBmatcher.cc:1102-1326) on flat data.  The caller projects its
   map points (pose / Sim3 algebra, frustum and distance tests, PredictScale) and hands over one query per surviving point:
   (x, y) = projection, radius = th * mvScaleFactors[level], ur = u - bf/z (only read by the stereo chi-square test), level = the
   predicted octave.  Per query the key points of KeyFrame::GetFeaturesInArea(x, y, radius) with octave level-1 .. level are
   compared; with chi2_gate != 0 Fuse's reprojection test (7.8 stereo / 5.99 mono, ORBmatcher.cc:901-926; the Sim3 overload of
   Fuse and SearchBySim3 use chi2_gate = 0) is applied first.  best_idx[q] = the first key point with the smallest descriptor
   distance (-1: none; a candidate at distance 256 is never reported — it cannot pass either threshold), best_dist[q] its distance;
   the caller applies `<= TH_LOW` (Fuse) / `<= TH_HIGH` (SearchBySim3) and does
   the map surgery / the mutual-consistency check.  Queries do not interact.
   The other pose-guided matchers are parameterisations of orbhip_search_by_projection mode 1: SearchByProjection(KeyFrame*, Scw,
   vpPoints, vpMatched, th) (ORBmatcher.cc:290-403) = levels [L-1, L], th_high = TH_LOW, no orientation check, blocked = vpMatched
   set; SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (:1472-1599) = levels [L-1, L+1], th_high = ORBdist,
   blocked = any map point (tests/test_reference_matchers.py runs both against the reference's own code). */
typedef struct { float x, y, radius, ur; int32_t level; } orbhip_best_query;
orbhip_status orbhip_search_best_in_window(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right /* may be NULL */,
                                           int n, int im_w, int im_h, const float* inv_level_sigma2, int nlevels,
                                           const orbhip_best_query* queries, const uint8_t* query_desc, int nq, int chi2_gate,
                                           int32_t* best_idx, int32_t* best_dist);

/* -------- DBoW2 vocabulary ----------------------------------------------------------
   ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (include/ORBVocabulary.h:31-32).  BowVector and
   FeatureVector (std::map in the reference) are returned flattened in map (ascending key) order:
     bow_id[i], bow_val[i]                      i < nbow           BowVector: word id -> weight (double)
     fv_node[j], fv_feat[fv_off[j] .. fv_off[j+1])  j < nfv        FeatureVector: node id -> feature indices, ascending
   Caller buffers hold n entries (fv_off: n + 1).  At most 7168 features per frame. */
typedef struct orbhip_voc orbhip_voc;
/* TemplatedVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1425; System.cc:68).  Blank lines
   are ignored (the reference reads uninitialised variables on the empty last line of a newline-terminated file). */
orbhip_status orbhip_voc_load_text(orbhip_voc** out, const char* path, int device);
void orbhip_voc_destroy(orbhip_voc* voc);
/* m_k, m_L, m_scoring, m_weighting, m_nodes.size(), size() */
orbhip_status orbhip_voc_info(const orbhip_voc* voc, int* k, int* L, int* scoring, int* weighting, int* nnodes, int* nwords);
/* transform(feature, word_id, weight, nid, levelsup) for n features (TemplatedVocabulary.h:1218-1262); outputs may be NULL */
orbhip_status orbhip_voc_transform_features(orbhip_voc* voc, const uint8_t* desc /* n x 32 */, int n, int levelsup,
                                            uint32_t* word, double* weight, uint32_t* node);
/* transform(features, BowVector&, FeatureVector&, levelsup) (TemplatedVocabulary.h:1127-1194), host descriptors in, synchronous */
orbhip_status orbhip_voc_transform(orbhip_voc* voc, const uint8_t* desc /* n x 32 */, int n, int levelsup,
                                   uint32_t* bow_id, double* bow_val, int* nbow,
                                   uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int* nfv);
/* Frame::ComputeBoW (Frame.cc:395-402) for the first nimg frames of the extractor's last call: reads the descriptors where
   the extraction left them in HBM, asynchronous on the extractor's stream; orbhip_fetch_bow copies one frame's result out. */
orbhip_status orbhip_compute_bow(orbhip_ctx* ctx, orbhip_voc* voc, int nimg, int levelsup);
orbhip_status orbhip_fetch_bow(orbhip_ctx* ctx, orbhip_voc* voc, int frame, uint32_t* bow_id, double* bow_val, int* nbow,
                               uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int* nfv);
/* ORBmatcher::SearchByBoW on flat data.  mode 0 = SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (ORBmatcher.h:60,
   ORBmatcher.cc:159-288), mode 1 = SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORBmatcher.h:61, ORBmatcher.cc:522-655).
   Side 1 is the key frame whose map points are handed over: valid1[i] = "vpMapPoints1[i] != NULL && !isBad()", angle1 =
   mvKeysUn[i].angle; side 2 the frame (mode 0: mvKeys[i].angle, valid2 ignored) or the second key frame (mode 1: valid2
   likewise).  fvX_* = the FeatureVector flattened as returned by orbhip_voc_transform.  match12[i1] = index on side 2 or -1;
   the caller writes vpMapPointMatches[match12[i1]] = vpMapPoints1[i1] (mode 0) / vpMatches12[i1] = vpMapPoints2[match12[i1]]
   (mode 1).  *nmatches = the reference's return value.  Synchronous, host pointers. */
orbhip_status orbhip_search_by_bow(int device, int mode,
                                   const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1,
                                   const uint32_t* fv1_node, const int32_t* fv1_off, const uint32_t* fv1_feat, int nfv1,
                                   const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2,
                                   const uint32_t* fv2_node, const int32_t* fv2_off, const uint32_t* fv2_feat, int nfv2,
                                   float nnratio, int check_ori, int32_t* match12, int* nmatches);
/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (ORBmatcher.h:65-66, ORBmatcher.cc:657-823,
   CheckDistEpipolarLine :140-157) on flat data.  Per key frame: descriptors, kp = 4 floats per feature (mvKeysUn x, y, angle,
   octave), has_mp[i] = "GetMapPoint(i) != NULL" (such features are skipped), stereo[i] = "mvuRight[i] >= 0", the FeatureVector.
   F12 = the 3x3 fundamental matrix, row-major; (ex, ey) = the epipole the caller computes from the poses (:663-669);
   scale_factors2 / level_sigma2_2 = pKF2->mvScaleFactors / mvLevelSigma2 (nlevels2 entries).  match12[i1] = index in key frame 2
   or -1 (vMatchedPairs = the pairs (i1, match12[i1]) in ascending i1); *nmatches = the return value.  Synchronous, host pointers. */
orbhip_status orbhip_search_for_triangulation(int device,
    const uint8_t* desc1, const float* kp1, const uint8_t* has_mp1, const uint8_t* stereo1, int n1,
    const uint32_t* fv1_node, const int32_t* fv1_off, const uint32_t* fv1_feat, int nfv1,
    const uint8_t* desc2, const float* kp2, const uint8_t* has_mp2, const uint8_t* stereo2, int n2,
    const uint32_t* fv2_node, const int32_t* fv2_off, const uint32_t* fv2_feat, int nfv2,
    const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2, int nlevels2,
    int only_stereo, int check_ori, int32_t* match12, int* nmatches);
/* -------- batched forms of the back end's matcher loops --------------------------------------------------------
   The reference calls these members in loops: one SearchByBoW per relocalisation candidate (Tracking.cc:1357-1380) / per loop candidate
   (LoopClosing.cc:239-375), one SearchForTriangulation per neighbour key frame (LocalMapping.cc:237-268), one Fuse per target key frame
   (LocalMapping.cc:483-514, LoopClosing.cc:589-599).  At 0.07-0.13 ms a call is launch latency, not work.  The batch entries run a whole loop
   as ONE upload (every distinct key frame / frame travels once), ONE launch set and ONE download; per pair / slot the answers are identical to
   the per-call entries above.  Host pointers, synchronous, on the calling thread's own stream. */
typedef struct {
    const uint8_t* desc; const float* angle; const uint8_t* valid; int32_t n;                   /* as desc1 / angle1 / valid1 / n1 of orbhip_search_by_bow (valid may be NULL on side 2 of mode 0) */
    const uint32_t* fv_node; const int32_t* fv_off; const uint32_t* fv_feat; int32_t nfv;      /* the FeatureVector, flattened */
} orbhip_bow_side;
typedef struct { const orbhip_bow_side* side1; const orbhip_bow_side* side2; int32_t* match12 /* side1->n entries */; int32_t nmatches; } orbhip_bow_pair;
/* npairs independent SearchByBoW calls (mode as in orbhip_search_by_bow).  Sides are recognised by POINTER: a side that several pairs name
   (the current frame of Relocalization, the current key frame of LoopClosing::ComputeSim3) is uploaded once. */
orbhip_status orbhip_search_by_bow_batch(int device, int mode, int npairs, orbhip_bow_pair* pairs, float nnratio, int check_ori);

typedef struct {
    const uint8_t* desc; const float* kp; const uint8_t* has_mp; const uint8_t* stereo; int32_t n;   /* as desc1 / kp1 / has_mp1 / stereo1 / n1 of orbhip_search_for_triangulation */
    const uint32_t* fv_node; const int32_t* fv_off; const uint32_t* fv_feat; int32_t nfv;
    const float* scale_factors; const float* level_sigma2; int32_t nlevels;                          /* mvScaleFactors / mvLevelSigma2: read of a pair's SECOND key frame only */
} orbhip_tri_side;
typedef struct { const orbhip_tri_side* kf2; float F12[9]; float ex, ey; int32_t* match12 /* kf1->n entries */; int32_t nmatches; } orbhip_tri_pair;
/* SearchForTriangulation of ONE key frame against npairs neighbours (LocalMapping::CreateNewMapPoints).  Every pair is searched with kf1->has_mp as
   handed in.  The reference's loop gives key frame 1 new map points between neighbours (LocalMapping.cc:437 AddMapPoint); its SearchForTriangulation
   never marks a feature of key frame 2 as taken (ORBmatcher.cc:677, 725: vbMatched2 is read, never written), so the features of key frame 1 do not
   interact and - WITHOUT the orientation check, which is how LocalMapping constructs its matcher (LocalMapping.cc:215: ORBmatcher(0.6, false)) - the
   reference's answer for neighbour i is this batch's answer minus the features that received a map point from neighbours < i: the caller drops
   those pairs (TriangulationPairs of orb_slam2_amd/cpp/ORBmatcher.cc does).  With check_ori != 0 the rotation histogram couples the features: use the
   batch only when key frame 1's map points do not change inside the loop. */
orbhip_status orbhip_search_for_triangulation_batch(int device, const orbhip_tri_side* kf1, int npairs, orbhip_tri_pair* pairs, int only_stereo, int check_ori);

/* TemplatedVocabulary::score(v1, v2) with the scoring object named by the file header (ScoringObject.cpp:24-313;
   KeyFrameDatabase.cc:133,249, LoopClosing.cc:134); host arithmetic */
double orbhip_voc_score(const orbhip_voc* voc, const uint32_t* id1, const double* val1, int n1,
                        const uint32_t* id2, const double* val2, int n2);

/* -------- distorted cameras and rectification -------------------------------------------
   Monocular / RGB-D cameras with lens distortion (TUM1-3.yaml): Frame's constructors call UndistortKeyPoints (Frame.cc:404-434)
   and, once, ComputeImageBounds (Frame.cc:436-464); the 64x48 grid and every windowed search then work on mvKeysUn inside the
   undistorted bounds.  Both run cv::undistortPoints(pts, pts, mK, mDistCoef, cv::Mat(), mK): double arithmetic, five fixed-point
   iterations (OpenCV 3.2 imgproc/undistort.cpp). */
typedef struct { float fx, fy, cx, cy; float k1, k2, p1, p2, k3; } orbhip_camera;   /* mK (Tracking.cc:60-68), mDistCoef (:70-82; k3 = 0 if absent) */
typedef struct { float min_x, min_y, max_x, max_y; } orbhip_bounds;                  /* Frame::mnMinX, mnMinY, mnMaxX, mnMaxY (Frame.h:187-190) */
/* cv::undistortPoints(xy, xy_out, K, D, Mat(), K) for n points (x, y interleaved), host buffers, synchronous */
orbhip_status orbhip_undistort_points(int device, const orbhip_camera* cam, const float* xy, int n, float* xy_out);
/* Frame::ComputeImageBounds for an im_w x im_h image (k1 == 0: 0, 0, im_w, im_h without touching the device) */
orbhip_status orbhip_image_bounds(int device, const orbhip_camera* cam, int im_w, int im_h, orbhip_bounds* out);
/* Attach the camera to an extractor context (NULL or k1 == 0: undistorted, the default).  With a distorted camera every
   extraction also leaves mvKeysUn in HBM (Frame::UndistortKeyPoints fused behind the descriptor kernel) and the match_prev
   matcher of the device-resident pipeline works on mvKeysUn over the undistorted bounds, like SearchForInitialization on
   distorted TUM frames.  orbhip_fetch_undistorted copies mvKeysUn of the last call ([nimg][cap], same counts as orbhip_fetch;
   == the key points themselves for an undistorted camera, Frame.cc:406-410). */
orbhip_status orbhip_set_camera(orbhip_ctx* ctx, const orbhip_camera* cam);
orbhip_status orbhip_get_bounds(const orbhip_ctx* ctx, orbhip_bounds* out);
orbhip_status orbhip_fetch_undistorted(orbhip_ctx* ctx, int nimg, orbhip_keypoint* kps_un, int cap);
/* The windowed searches above with explicit image bounds (their im_w / im_h forms are these with bounds = 0, 0, im_w, im_h) */
orbhip_status orbhip_search_for_initialization_bounds(int device,
                                               const orbhip_keypoint* kps1, const uint8_t* desc1, int n1,
                                               const orbhip_keypoint* kps2, const uint8_t* desc2, int n2,
                                               const orbhip_bounds* bounds, float* prev_matched, int32_t* matches12,
                                               int window, float nnratio, int check_ori, int* nmatches);
orbhip_status orbhip_search_by_projection_bounds(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right,
                                          const uint8_t* blocked, int n, const orbhip_bounds* bounds,
                                          const orbhip_proj_query* queries, const uint8_t* query_desc, int nq,
                                          int mode, float nnratio, int th_high, int check_ori,
                                          int32_t* feature_query, int* nmatches);
orbhip_status orbhip_search_best_in_window_bounds(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right /* may be NULL */,
                                           int n, const orbhip_bounds* bounds, const float* inv_level_sigma2, int nlevels,
                                           const orbhip_best_query* queries, const uint8_t* query_desc, int nq, int chi2_gate,
                                           int32_t* best_idx, int32_t* best_dist);

/* The same search for several frames (camera slots) in ONE pass (the multi-camera form of M2 / M3): slot s searches its
   own frame with its own queries, the slots share mode, thresholds and image bounds (one camera geometry, e.g. the cameras of a rig or a
   queue of frames of one camera).  One upload, one launch sequence with one workgroup per slot in the order-dependent kernel, one download
   — the per-slot results are identical to nslots calls of orbhip_search_by_projection_bounds.  u_right / blocked may be NULL per slot. */
typedef struct {
    const orbhip_keypoint* kps; const uint8_t* desc; const float* u_right; const uint8_t* blocked; int32_t n;        /* the frame searched */
    const orbhip_proj_query* queries; const uint8_t* query_desc; int32_t nq;                                         /* its map points */
    int32_t* feature_query; int32_t nmatches;                                                                        /* out: n entries; the return value */
} orbhip_proj_slot;
orbhip_status orbhip_search_by_projection_batch(int device, int nslots, orbhip_proj_slot* slots, const orbhip_bounds* bounds,
                                                int mode, float nnratio, int th_high, int check_ori);

/* orbhip_search_best_in_window_bounds for several key frames in ONE pass (Fuse over all target key frames, LocalMapping.cc:483-514: the same
   map points projected into every target; LoopClosing::SearchAndFuse, LoopClosing.cc:589-599): slot s searches its own key frame with its own
   queries.  Per slot the answers equal a call of orbhip_search_best_in_window_bounds.  (Fuse's map surgery can change a later target's inputs -
   Replace() makes a point bad and recomputes the survivor's descriptor, MapPoint.cc:205-235: the caller re-runs the queries whose point changed.) */
typedef struct {
    const orbhip_keypoint* kps; const uint8_t* desc; const float* u_right; int32_t n;                    /* the key frame searched: mvKeysUn, mDescriptors, mvuRight (may be NULL) */
    orbhip_bounds bounds; const float* inv_level_sigma2; int32_t nlevels;
    const orbhip_best_query* queries; const uint8_t* query_desc; int32_t nq;
    int32_t* best_idx; int32_t* best_dist;                                                              /* out: nq entries each */
} orbhip_best_slot;
orbhip_status orbhip_search_best_in_window_batch(int device, int nslots, orbhip_best_slot* slots, int chi2_gate);

/* The same two searches on a frame that is still on the device (frame `frame` of the context's last extract call): key points
   (mvKeysUn when a distorted camera is attached), descriptors, the image bounds and — with use_u_right != 0 — mvuRight of the last
   orbhip_compute_stereo_matches (this context = left) / orbhip_compute_stereo_from_rgbd are read in HBM; only the queries, the
   `blocked` flags and the results cross PCIe.  n = the frame's key point count as reported by orbhip_fetch (F.N); blocked / feature_query
   hold n entries; inv_level_sigma2 of the best-in-window form = this extractor's own table. */
orbhip_status orbhip_search_by_projection_frame(orbhip_ctx* ctx, int frame, int n, int use_u_right, const uint8_t* blocked,
                                                const orbhip_proj_query* queries, const uint8_t* query_desc, int nq,
                                                int mode, float nnratio, int th_high, int check_ori, int32_t* feature_query, int* nmatches);
orbhip_status orbhip_search_best_in_window_frame(orbhip_ctx* ctx, int frame, int n, int use_u_right,
                                                 const orbhip_best_query* queries, const uint8_t* query_desc, int nq, int chi2_gate,
                                                 int32_t* best_idx, int32_t* best_dist);

/* -------- the pose-guided matchers with the PROJECTION on the device ----------------------------------------
   The five projection-guided members of ORBmatcher evaluate, per map point, a few lines of cv::Mat algebra before their window search:
       SearchByProjection(Frame&, const Frame&, th, bMono)                 ORBmatcher.cc:1353-1395
       SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)   ORBmatcher.cc:1490-1528
       SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)         ORBmatcher.cc:316-362
       Fuse(KeyFrame*, vpMapPoints, th) / Fuse(KeyFrame*, Scw, ...)        ORBmatcher.cc:850-892 / :1004-1051
       SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)            ORBmatcher.cc:1154-1191 / :1234-1271
   The entries below take what the member READS - one orbhip_map_point per map point that passed the member's own pointer / isBad / already-found
   filters, the pose(s) and intrinsics of the call - and run transform -> depth / image-bounds / distance / viewing-angle gates -> PredictScale ->
   radius -> window search in ONE launch chain; the host keeps the MapPoint gathers and the map surgery.  A point that fails a gate is a query
   without candidates (feature_query never names it, best_idx = -1).  Every float operation is the reference's, in its order, each rounded once
   (no contraction); `kind` selects the member's statement sequence (they differ: 1/z vs 1.0/z, fx*x+cx vs fx*xc*invzc+cx, `<` vs `<=` bounds,
   which gates exist).  orb_slam2_amd/cpp/ORBmatcher.cc is the caller. */
#define ORBHIP_MAX_PROJ_LEVELS 16
typedef struct {
    float x, y, z;              /* MapPoint::GetWorldPos() */
    float cam_x, cam_y, cam_z;  /* read only with gemm_mode 2: the point in the frame the member projects from, from the caller's own cv::Mat expression */
    float nx, ny, nz;           /* MapPoint::GetNormal(): kinds with the viewing-angle gate (PO.dot(Pn) < 0.5*dist), else unused */
    float min_dist, max_dist;   /* MapPoint::GetMinDistanceInvariance() / GetMaxDistanceInvariance(): kinds with the distance gate */
    float scale_dist;           /* mfMaxDistance, the numerator of MapPoint::PredictScale (MapPoint.cc:390, 407); read when level < 0 */
    int32_t level;              /* >= 0: the level is given (ORBmatcher.cc:1376: LastFrame.mvKeys[i].octave); -1: PredictScale on the device */
    int32_t blocks;             /* orbhip_proj_query.blocks */
    float angle;                /* orbhip_proj_query.angle */
} orbhip_map_point;             /* 60 bytes */
typedef enum {
    ORBHIP_PROJ_LAST_FRAME = 0, /* :1353-1395  x3Dc = Rcw*x3Dw+tcw; invzc = 1.0/z; invzc<0 -> skip; u = fx*xc*invzc+cx; u<minX || u>maxX -> skip;
                                               radius = th*sf[octave]; levels by forward / backward; ur = u - mbf*invzc */
    ORBHIP_PROJ_FRAME_KF = 1,   /* :1490-1528  the same projection WITHOUT the depth test; PO = x3Dw-Ow, dist3D = norm(PO) in [min, max];
                                               PredictScale; radius = th*sf[level]; levels level-1 .. level+1 */
    ORBHIP_PROJ_KF_SIM3 = 2,    /* :316-362    p3Dc = Rcw*p3Dw+tcw; z<0 -> skip; invz = 1/z; x = X*invz; u = fx*x+cx; IsInImage; distance gate on
                                               norm(p3Dw-Ow); PO.dot(Pn) < 0.5*dist -> skip; PredictScale; radius = th*sf[level]; levels level-1 .. level */
    ORBHIP_PROJ_FUSE = 3,       /* :850-892    as KF_SIM3, + ur = u - bf*invz */
    ORBHIP_PROJ_FUSE_SIM3 = 4,  /* :1004-1051  as KF_SIM3 with invz = 1.0/z */
    ORBHIP_PROJ_SIM3 = 5        /* :1154-1191  p3Dc1 = R*p3Dw+t; p3Dc2 = R2*p3Dc1+t2; z<0 -> skip; invz = 1.0/z; IsInImage; distance gate on norm(p3Dc2);
                                               PredictScale; radius = th*sf[level]; no viewing-angle gate */
} orbhip_projection_kind;
typedef struct {
    int32_t kind;               /* orbhip_projection_kind */
    int32_t gemm_mode;          /* how `R*x+t` rounds: 0 = products accumulated in double, rounded to float, then + t in float
                                   (cv::gemm's generic kernel; include/cvlite); 1 = OpenCV's small-matrix path: a0*b0 + a1*b1 + a2*b2 in float,
                                   then (float)((double)t0 + (double)c); 2 = not on the device: cam_x/y/z of every point hold the result of the
                                   caller's own cv::Mat expression (after BOTH transforms for SIM3) */
    float R[9], t[3];           /* Rcw | tcw, row-major (R1w | t1w for SIM3) */
    float R2[9], t2[3];         /* SIM3: sR21 | t21 (searching key frame 2) or sR12 | t12 (searching key frame 1) */
    float Ow[3];                /* camera centre the member computes (-Rcw.t()*tcw / KeyFrame::GetCameraCenter()) */
    float fx, fy, cx, cy, bf;
    float min_x, min_y, max_x, max_y;   /* Frame::mnMinX .. (float) / KeyFrame::mnMinX .. (int, converted): what the member's bounds test reads */
    float th;
    int32_t forward, backward;  /* LAST_FRAME: bForward / bBackward (:1348-1349) */
    int32_t nlevels;            /* mnScaleLevels */
    float scale_factors[ORBHIP_MAX_PROJ_LEVELS];    /* mvScaleFactors */
    /* MapPoint::PredictScale (MapPoint.cc:385-421) without a device logarithm: nScale = ceil(log(ratio)/mfLogScaleFactor), clamped to [0, nlevels-1],
       is a non-decreasing step function of ratio = mfMaxDistance/dist, so it equals the number of i < nlevels-1 with ratio >= level_ratio[i], where
       level_ratio[i] = the smallest float ratio the HOST's own expression maps to a level > i (orbhip_predict_scale_table). */
    float level_ratio[ORBHIP_MAX_PROJ_LEVELS];
} orbhip_projection;
/* level_ratio[0 .. nlevels-2] for a given mfLogScaleFactor by bisection over the floats with `level_of(ratio, user)` = the caller's own
   PredictScale expression (so that whatever log / ceil overloads the caller's translation unit resolves to are the ones reproduced);
   entries from nlevels-1 on are +inf.  Fails if level_of is not monotone around a threshold (checked +-64 floats). */
orbhip_status orbhip_predict_scale_table(int (*level_of)(float ratio, void* user), void* user, int nlevels, float* level_ratio /* ORBHIP_MAX_PROJ_LEVELS */);

/* orbhip_search_by_projection_bounds / _frame with one orbhip_map_point per query instead of one orbhip_proj_query; mode 1 semantics (best only,
   rotation histogram when check_ori).  queries_out (may be NULL): np orbhip_proj_query records as the device derived them - radius < 0 marks a
   point that failed a gate - for inspection and the parity tests. */
orbhip_status orbhip_project_search_bounds(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right,
                                           const uint8_t* blocked, int n, const orbhip_bounds* bounds,
                                           const orbhip_projection* proj, const orbhip_map_point* points, const uint8_t* point_desc, int np,
                                           float nnratio, int th_high, int check_ori, int32_t* feature_query, int* nmatches,
                                           orbhip_proj_query* queries_out);
orbhip_status orbhip_project_search_frame(orbhip_ctx* ctx, int frame, int n, int use_u_right, const uint8_t* blocked,
                                          const orbhip_projection* proj, const orbhip_map_point* points, const uint8_t* point_desc, int np,
                                          float nnratio, int th_high, int check_ori, int32_t* feature_query, int* nmatches,
                                          orbhip_proj_query* queries_out);
/* orbhip_search_best_in_window_bounds / _batch likewise (Fuse x2, SearchBySim3's two passes as two slots of one call).  queries_out as above
   (orbhip_best_query records, radius < 0 = gated out). */
orbhip_status orbhip_project_best_in_window_bounds(int device, const orbhip_keypoint* kps, const uint8_t* desc, const float* u_right /* may be NULL */,
                                                   int n, const orbhip_bounds* bounds, const float* inv_level_sigma2, int nlevels,
                                                   const orbhip_projection* proj, const orbhip_map_point* points, const uint8_t* point_desc, int np,
                                                   int chi2_gate, int32_t* best_idx, int32_t* best_dist, orbhip_best_query* queries_out);
typedef struct {
    const orbhip_keypoint* kps; const uint8_t* desc; const float* u_right; int32_t n;                    /* the key frame searched */
    orbhip_bounds bounds; const float* inv_level_sigma2; int32_t nlevels;
    const orbhip_projection* proj; const orbhip_map_point* points; const uint8_t* point_desc; int32_t np;
    int32_t* best_idx; int32_t* best_dist;                                                              /* out: np entries each */
} orbhip_project_best_slot;
orbhip_status orbhip_project_best_in_window_batch(int device, int nslots, orbhip_project_best_slot* slots, int chi2_gate);
/* The same with ONE set of points offered to every slot - LocalMapping::SearchInNeighbors (LocalMapping.cc:483-514) fuses the same map points into every
   neighbour.  points / point_desc / np are taken from slots[0] and travel once (every slot must name the same arrays); proj, the key frame and the answers
   are per slot (best_idx / best_dist: np entries each).  skip (may be NULL): np 64-bit masks, bit s set = point k is not searched in slot s (it is in that key
   frame already, ORBmatcher.cc:848-849): best_idx -1, best_dist 256.  At most 64 slots. */
orbhip_status orbhip_project_best_in_window_shared(int device, int nslots, orbhip_project_best_slot* slots, const uint64_t* skip, int chi2_gate);
/* Slot `slot` of the calling thread's LAST orbhip_project_best_in_window_shared call searched again with other points (np entries in, np answers out): the
   slot's key frame, descriptors and grid table are still in the thread's device scratch, only the points travel.  ORBHIP_ERR_INVALID when the thread has made
   another scratch-using call since (any stateless matcher entry) or the held scratch has no room for np points: run the full entry then. */
orbhip_status orbhip_project_best_in_window_held(int device, int slot, const orbhip_projection* proj, const orbhip_map_point* points, const uint8_t* point_desc, int np,
                                                 int chi2_gate, int32_t* best_idx, int32_t* best_dist);

/* Stereo rectification on the input side: the EuRoC example runs cv::remap(raw, rect, M1, M2, cv::INTER_LINEAR) on the CPU for both
   images of every pair before TrackStereo (Examples/Stereo/stereo_euroc.cc:136-137; maps from cv::initUndistortRectifyMap(..,
   CV_32F, ..) :97-98).  orbhip_set_rectification uploads the two CV_32FC1 maps ([height][width] of the context, contiguous; they
   address a src_w x src_h raw image) once; the *_rectify entry points then take RAW frames and remap on the device (OpenCV's
   fixed-point bilinear: 5 fractional bits, constant 0 border) straight into the context's level-0 plane — the rectified image never
   exists on the host.  map_x == NULL removes the maps. */
orbhip_status orbhip_set_rectification(orbhip_ctx* ctx, const float* map_x, const float* map_y, int src_w, int src_h);
orbhip_status orbhip_extract_batch_rectify(orbhip_ctx* ctx, int nimg, const uint8_t* const* raw_imgs, int stride_bytes,
                                           orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out /* nimg */);
orbhip_status orbhip_extract_device_rectify(orbhip_ctx* ctx, int nimg, const uint8_t* d_raw, size_t frame_stride, int row_stride,
                                            int match_prev, int window, float nnratio, int check_ori);

/* -------- measurement + stage dumps (parity tests) ----------------------------------------------------- */
/* per-kernel HIP-event timing on the context's stream: enable, run, then read accumulated stats */
orbhip_status orbhip_profile_enable(orbhip_ctx* ctx, int on);
int orbhip_profile_num_kernels(const orbhip_ctx* ctx);
orbhip_status orbhip_profile_get(orbhip_ctx* ctx, int k, const char** name, double* total_ms, int64_t* launches);
orbhip_status orbhip_profile_reset(orbhip_ctx* ctx);
/* Tiles of the last pyramid level that k_pyramid_cascade's workgroups own (calls with up to eight frames compute every level in one launch,
 * DESIGN.md section 4), or 0 where the context's shape takes the level-by-level kernels instead (scale factors from ~1.6, very large levels). */
int orbhip_pyramid_cascade_tiles(const orbhip_ctx* ctx);
/* algorithmic bytes one frame moves (BASELINE.md §3 formula B(W,H,N)) */
int64_t orbhip_algorithmic_bytes_per_frame(const orbhip_ctx* ctx);
/* the terms of that formula moved by profiled kernel k (0 for kernels outside the formula) */
int64_t orbhip_algorithmic_bytes_per_frame_kernel(const orbhip_ctx* ctx, int k);

orbhip_status orbhip_debug_blurred_level(orbhip_ctx* ctx, int frame, int level, uint8_t* dst, int dst_stride);
/* (x, y, score) int32 triples in vToDistributeKeys order (cell-space coordinates, ORBextractor.cc:820-825) */
orbhip_status orbhip_debug_candidates(orbhip_ctx* ctx, int frame, int level, int32_t* xys, int cap, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* ORBHIP_H */
