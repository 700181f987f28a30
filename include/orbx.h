/* orbx.h — C ABI of the MI355X-native ORB front-end (liborbx_hip.so).
 *
 * Drop-in boundary for the two reference classes that own the hot path:
 *   ORB_SLAM3::ORBextractor   /root/reference/include/ORBextractor.h:43-109
 *   ORB_SLAM3::ORBmatcher     /root/reference/include/ORBmatcher.h:36-103  (+ the stereo association that
 *                             Frame runs on the extractor's pyramids, src/Frame.cc:1102-1358, :1530-1587)
 * The reference has no FFI layer of its own: callers use those classes directly.  The C++ facade in
 * include/orb_slam3_amd/ keeps the reference's class names, signatures and return conventions and forwards to
 * the functions below; INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions: plain pointers and sizes only.  Every function returns an int status (0 = ORBX_OK, < 0 =
 * error) unless stated otherwise.  A handle owns one GPU stream pair and all device memory it needs; calls
 * on ONE handle must not overlap in time (the reference never calls one ORBextractor instance
 * concurrently, src/Frame.cc:136-141), calls on DIFFERENT handles may run from different threads.
 * Nothing here ever computes on the CPU: if no GPU / HIP runtime is usable, orbx_create fails.
 */
#ifndef ORBX_H
#define ORBX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_OK 0
#define ORBX_E_EMPTY (-1)      /* empty image: ORBextractor::operator() returns -1, src/ORBextractor.cc:1561-1562 */
#define ORBX_E_ARG (-2)        /* bad argument / image too small for the 35-px FAST cell grid */
#define ORBX_E_DEVICE (-3)     /* HIP error (no device, allocation or launch failure) */
#define ORBX_E_CAPACITY (-4)   /* caller buffer too small (n_out still reports the required size) */
#define ORBX_E_INTERNAL (-5)   /* device-side capacity check tripped */

/* cv::KeyPoint as the reference fills it (src/ORBextractor.cc:1184-1198, :587): 28 bytes */
typedef struct OrbxKeyPoint {
    float x, y;        /* level-0 pixel coordinates (level coords * mvScaleFactor[octave]) */
    float size;        /* int(31 * scale) */
    float angle;       /* degrees [0,360], cv::fastAtan2 of the intensity-centroid moments */
    float response;    /* FAST corner score (OpenCV cornerScore), NOT a Harris score */
    int32_t octave;
    int32_t class_id;  /* always -1 */
} OrbxKeyPoint;

typedef struct orbx_extractor orbx_extractor;

/* number of usable GPUs (0 if none) */
int orbx_device_count(void);
/* How the host threads of this process wait for GPU `device_id` inside the synchronous calls (orbx_extract, orbx_fetch, orbx_sync, the orbm_* searches):
 * 0 = the HIP runtime's default (the waiting thread spins: lowest latency - one pair per call is 0.127 ms - and one host core per waiting thread),
 * 1 = blocking (the thread sleeps on the completion interrupt: a wake-up of ~10 - 20 us per wait and no core: what a host that runs one process per GPU on
 * fewer cores than 2 x GPUs wants - bench.py --gpus N uses it for N > 1), 2 = spin, 3 = yield.  Process-wide per device (hipSetDeviceFlags). */
int orbx_set_host_wait(int device_id, int mode);

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)  include/ORBextractor.h:49-50,
 * src/ORBextractor.cc:468-571.  device_id selects the GPU (one process per GPU in multi-GPU runs).
 * Limits, all reported loudly: images of at most 4127 x 4127 pixels and at least the 35-px FAST cell grid on the smallest level (ORBX_E_ARG from the
 * extraction); fewer than 65 535 keypoints per image over all levels (16-bit node indices; the reference's settings files ask for 500 - 2 000 and
 * Tracking builds its monocular initialisation extractor with five times that, src/Tracking.cc:634-635, :1331-1332: up to 10 000).
 * The quadtree of a level (src/ORBextractor.cc:711-1057) works in LDS when its node lists fit - 81 bytes per node beside 16 - 55 KB of bucket
 * counters and coordinate tables: with the 160 KB of gfx950 that is every level of up to ~1 400 - 1 800 keypoints, i.e. every level of every stereo,
 * RGB-D and post-initialisation monocular setting - and in a global node pool otherwise (level 0 of 10 000 features at 1241 x 376 holds 2 172):
 * the same algorithm and the same result, without a bound of its own. */
int orbx_create(orbx_extractor** out, int nfeatures, float scale_factor, int nlevels,
                int ini_th_fast, int min_th_fast, int device_id);
void orbx_destroy(orbx_extractor* h);

/* Which 8-bit 7x7 sigma=2 Gaussian the linked OpenCV would apply (src/ORBextractor.cc:1632):
 * 0 = OpenCV >= 3.4/4.x fixed-point path, taps [18,34,48,56,48,34,18]/256 (default)
 * 1 = OpenCV 3.2 integer separable filter, taps [18,34,49,55,49,34,18]/256 */
int orbx_set_gaussian_taps(orbx_extractor* h, int variant);

/* Pre-allocate for images of width x height and batches up to max_batch (otherwise done lazily). */
int orbx_reserve(orbx_extractor* h, int width, int height, int max_batch);

/* Getters of include/ORBextractor.h:61-81 (by value in the reference). */
int orbx_get_levels(const orbx_extractor* h);
float orbx_get_scale_factor(const orbx_extractor* h);
int orbx_get_level_tables(const orbx_extractor* h, float* scale, float* inv_scale, float* sigma2,
                          float* inv_sigma2, int* features_per_level, int* umax16);
/* Upper bound of keypoints per image (nfeatures + 3 per level, src/ORBextractor.cc:912,1006): size output buffers with it. */
int orbx_max_keypoints(const orbx_extractor* h);

/* ORBextractor::operator()(image, mask, keypoints, descriptors, vLappingArea)  include/ORBextractor.h:57-59,
 * src/ORBextractor.cc:1557-1682.  image: 8-bit grey, host memory.  kps: cap records, desc: cap x 32 bytes (row i <->
 * keypoint i).  *n_out = number of keypoints, *mono_index_out = the reference's return value (monoIndex).
 * Blocking (the reference call is synchronous). */
int orbx_extract(orbx_extractor* h, const uint8_t* image, int width, int height, int stride,
                 int lap0, int lap1, OrbxKeyPoint* kps, uint8_t* desc, int cap,
                 int* n_out, int* mono_index_out);

/* Batched form: B independent images of identical size, image b at images + b*image_stride.
 * images_on_device != 0: `images` is device memory of this handle's GPU (see orbx_device_alloc).
 * Asynchronous: enqueues on the handle's stream; results stay resident on the device for orbx_fetch and
 * for the orbm_* matchers.  */
int orbx_extract_batch(orbx_extractor* h, int B, const uint8_t* images, int width, int height, int stride,
                       size_t image_stride, int images_on_device, int lap0, int lap1);
/* Copy the results of the last batch to the host (blocking).  kps: [B][cap], desc: [B][cap][32], n_out/mono_out: [B]. */
int orbx_fetch(orbx_extractor* h, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out, int* mono_out);
int orbx_sync(orbx_extractor* h);

/* mvImagePyramid[level] of image `image_index` of the last batch (public member include/ORBextractor.h:83; the
 * un-bordered level — the 19-px reflect border is never read by the hot path).  blurred != 0 returns the
 * GaussianBlur'ed level used for the descriptors.  dst may be NULL to query the size. */
int orbx_pyramid_level(orbx_extractor* h, int image_index, int level, int blurred, uint8_t* dst,
                       int dst_stride, int* width, int* height);
/* all levels of one image with a single device-to-host copy: dst[l] receives level l (width x height of orbx_pyramid_level) with row pitch dst_stride[l] */
int orbx_pyramid_fetch(orbx_extractor* h, int image_index, int blurred, uint8_t* const* dst, const int* dst_stride);

/* device memory helpers so a caller can keep inputs resident (bench, multi-camera rigs) */
int orbx_device_alloc(orbx_extractor* h, size_t bytes, void** dptr);
int orbx_device_free(orbx_extractor* h, void* dptr);
int orbx_device_upload(orbx_extractor* h, void* dptr, const void* host, size_t bytes);
/* the same on the handle's copy stream, without waiting: the next orbx_extract_batch(..., on_device = 1) of this handle waits for it on the
 * device, and the copy itself waits until the previous extraction has read its input (upload batch i + 1 while batch i is processed;
 * `host` should be page-locked, orbx_host_alloc) */
int orbx_device_upload_async(orbx_extractor* h, void* dptr, const void* host, size_t bytes);

/* Frame::UndistortKeyPoints (src/Frame.cc:1003-1034) on the device: with K = (fx, fy, cx, cy) and dist = mDistCoef (k1, k2, p1, p2[, k3]; ndist 4 or 5)
 * every extraction also produces mvKeysUn - cv::undistortPoints(keys, K, dist, R = empty, P = K) in OpenCV's double arithmetic (restated;
 * opencv_variant 0 = OpenCV >= 3.4.2, 1 = 3.2) - for orbx_fetch_undistorted and for the device-resident consumers (orbm_stereo_from_depth,
 * orbm_search_local_points_batch).  dist == NULL or dist[0] == 0: mvKeysUn = mvKeys, as the reference decides (:1005-1009).
 * orbx_undistorted_bounds = Frame::ComputeImageBounds (:1043-1075): out = mnMinX, mnMaxX, mnMinY, mnMaxY. */
int orbx_set_undistort(orbx_extractor* h, const float K[4], const float* dist, int ndist, int opencv_variant);
int orbx_fetch_undistorted(orbx_extractor* h, OrbxKeyPoint* kps_un, int cap);       /* [B][cap], blocking */
int orbx_undistorted_bounds(const orbx_extractor* h, int width, int height, float out[4]);

/* Zero-copy input: the address and layout of pyramid level 0 inside the handle, for B images of width x height (reserves like orbx_reserve).
 * A producer that can write there - a camera DMA, a decoder, orbx_input_upload below - hands its frames over without the import pass
 * (one read and one write of every pixel): call orbx_extract_batch(h, B, *dptr, width, height, *stride, *image_stride, 1, ...) with exactly
 * these values and the extraction reads level 0 in place.  Image b starts at dptr + b * image_stride, rows are `stride` bytes apart; bytes
 * beyond `width` in a row are padding: write ONLY bytes [0, width) of a row.  The padding was zeroed when the buffer was laid out and the
 * kernels load it as part of whole dwords; no output depends on its value (tests/test_emu_parity.py: test_zero_copy_input_garbage_padding_* fill it with garbage), but a
 * producer that copies full-pitch rows makes every run read bytes it does not control.  The buffer stays valid until the handle is reconfigured for another size or a larger batch; level 0
 * is never written by the extraction, so a resident batch can be extracted repeatedly. */
int orbx_input_buffer(orbx_extractor* h, int width, int height, int B, void** dptr, int* stride, size_t* image_stride);
/* B host images (stride / image_stride as in orbx_extract_batch) into that buffer; blocking */
int orbx_input_upload(orbx_extractor* h, int B, const uint8_t* images, int width, int height, int stride, size_t image_stride);

/* device addresses of the results of the last batch: keypoints [B][cap] (28-byte records), descriptors [B][cap][32] (rows beyond n[b] are
 * zero), n [B], monoIndex [B]; valid until the next extraction / reconfiguration of the handle (synchronise with orbx_sync first).  For
 * consumers that stay on the device, e.g. an RCCL all-gather of the descriptor blocks (orb_slam3_detailed_comments_amd/multi.py). */
int orbx_device_outputs(orbx_extractor* h, void** kps, void** desc, void** n, void** mono, int* cap, int* B);
/* A SNAPSHOT of the descriptor blocks and counts of the last batch in device memory the CALLER owns (desc_dst: B * cap * 32 bytes, n_dst: B ints,
 * either may be NULL): device-to-device copies on the handle's stream behind the extraction, complete when the call returns.  The handle's own
 * buffers are rewritten by its next extraction, so a consumer that overlaps that extraction (an asynchronous collective) works on a snapshot. */
int orbx_device_snapshot(orbx_extractor* h, void* desc_dst, void* n_dst);
/* ---- the exchange step of the multi-GPU mode (BASELINE.json configs[4]) ----
 * Streams are independent: one process (or thread) per GPU, each with its own extractor / matcher handles created with that GPU's device_id,
 * and no collective on the hot path.  The ONE exchange a host may want - every rank's descriptor blocks on every rank, in front of a global
 * matcher - is an all-gather over RCCL / xGMI, offered here so that a C++ host needs neither Python nor torch.distributed:
 *   rank 0:      orbx_comm_unique_id(id)  ->  id reaches the other ranks by whatever the host has (MPI, a socket, a file, torch.distributed)
 *   every rank:  orbx_comm_create(&c, world, rank, id, device_id)         (collective: ncclCommInitRank; or orbx_comm_adopt for an ncclComm_t it owns)
 *   per batch:   orbx_extract_batch(h, ...); orbx_allgather_descriptors(h, c, &desc_all, &n_all, &B, &cap);  ...next extraction...;  orbx_comm_wait(c)
 * orbx_allgather_descriptors snapshots the descriptor block [B][cap][32] (rows beyond n[b] zero) and the counts [B] of h's last batch behind the
 * extraction, then gathers both over the communicator's own stream: ASYNCHRONOUS - the handle may extract its next batch at once.  desc_all
 * [world][B][cap][32] and n_all [world][B] are device memory of the communicator, valid after orbx_comm_wait and until the next call; every
 * rank must gather blocks of the same shape (same B, same extractor geometry).  orbx_comm_fetch waits and copies them to the host.
 * A call that finds the communicator's previous exchange still in flight does not wait for it on the host: its snapshot is ordered behind that exchange
 * on the device (several handles may share one communicator).  The gathered blocks are double-buffered: what a call returned is written again by the
 * SECOND call after it, not by the next one, so a consumer of exchange k (on a stream of its own, after orbx_comm_wait) may run beside exchange k + 1;
 * n_all sits behind desc_all at world * B * cap * 32 bytes - take both pointers from the call, they move when B changes.  After a failed call
 * orbx_comm_fetch refuses (nothing was gathered).
 * librccl is loaded (dlopen) by the first orbx_comm_* call; hosts that never call them never load it. */
typedef struct orbx_comm orbx_comm;
#define ORBX_COMM_ID_BYTES 128            /* sizeof(ncclUniqueId) */
int orbx_comm_unique_id(uint8_t id[ORBX_COMM_ID_BYTES]);
int orbx_comm_create(orbx_comm** c, int world, int rank, const uint8_t id[ORBX_COMM_ID_BYTES], int device_id);
int orbx_comm_adopt(orbx_comm** c, void* nccl_comm /* ncclComm_t of the caller, not destroyed by orbx_comm_destroy */, int world, int rank, int device_id);
void orbx_comm_destroy(orbx_comm* c);
int orbx_comm_world(const orbx_comm* c);
int orbx_comm_rank(const orbx_comm* c);
int orbx_allgather_descriptors(orbx_extractor* h, orbx_comm* c, void** desc_all, void** n_all, int* B, int* cap);   /* out pointers may be NULL */
int orbx_comm_wait(orbx_comm* c);
int orbx_comm_fetch(orbx_comm* c, uint8_t* desc_all_host, int* n_all_host);                                           /* either may be NULL */

/* GPU index the handle was created on; ORBX_DEVICE_HOST when the library's "device" memory is plain host memory (only the CPU emulator build of
 * the kernel sources that the tests use - the product library never returns it) */
#define ORBX_DEVICE_HOST (-1)
int orbx_device_id(const orbx_extractor* h);
/* page-locked host memory for output buffers: with cap == orbx_max_keypoints() orbx_fetch / orbm_stereo_fetch copy straight
 * into the caller's arrays (no staging, no repacking) and pinned memory makes that copy run at PCIe speed */
int orbx_host_alloc(orbx_extractor* h, size_t bytes, void** hptr);
int orbx_host_free(orbx_extractor* h, void* hptr);

/* How ComputePyramid (src/ORBextractor.cc:1687-1738) is launched: 0 (default) = by batch size - all levels in one launch for small
 * batches, where a chain of one launch per level is pure launch latency, one streaming launch per level for large ones; 1 / 2 force
 * either form (tests compare them).  The levels are bit-identical in every mode. */
int orbx_set_pyramid_mode(orbx_extractor* h, int mode);

/* Small batches (up to 32 images per call; one stereo pair per call is Tracking's rhythm) are launch-latency bound and run a launch form of
 * their own: the blur strips and the FAST cells in ONE launch on one stream (large batches: two launches on two streams, i.e. a fork and a
 * join of ~6 us each).  on = 1 (default) / 0 = the large-batch form at every batch size (tests compare them: the outputs are bit-identical).
 * The profiling modes (orbx_profile_enable) time stages and always run the large-batch form. */
int orbx_set_small_batch_forms(orbx_extractor* h, int on);

/* Replay the extraction pipeline as one hipGraph (captured on first use, re-captured when the batch size, geometry, input
 * pointer or lapping area change).  Pays off at small batches, where the ~17 kernel launches are latency-bound. */
int orbx_set_graph_replay(orbx_extractor* h, int on);

/* Per-stage GPU time of the last batch, HIP events on the launching streams.  on = 1: normal two-stream schedule (the blur
 * overlaps FAST + quadtree, so their times overlap too); on = 2: serial schedule, every kernel alone on one stream. */
#define ORBX_NSTAGES 8
int orbx_profile_enable(orbx_extractor* h, int on);
int orbx_profile_get(orbx_extractor* h, float ms[ORBX_NSTAGES]);
const char* orbx_stage_name(int i);

/* Stage probes for parity tests (level-ordered intermediate results of image `image_index`). */
int orbx_debug_candidates(orbx_extractor* h, int image_index, int level, int* xys, int cap);       /* returns count; (x,y,score) rel. to the 16-px border, reference order */
int orbx_debug_level_keys(orbx_extractor* h, int image_index, int level, int* xys, int cap);       /* quadtree output in list order */
/* test switches of the stereo row search (orbm_stereo_match): bit 0 visits the candidates of a row band in the opposite order (the result
 * must not change), bit 2 makes every lane walk all candidates from the last to the first (every tie then meets inside one lane; the
 * result must not change either), bit 1 restores the distance-only tie rule of an earlier revision (the tie tests must then fail);
 * bit 3: the 2-NN of orbm_knn2 / orbm_stereo_fisheye on the vector units (one wave per query) instead of the matrix cores (the result must
 * not change); bit 4: orbm_project_points and orbm_search_by_projection_lastframe_batch evaluate `Tcw * p` as q.toRotationMatrix() * p + t
 * (an earlier revision's form) instead of Sophus' quaternion action (tests/test_sophus_action.py: the reference must then catch it) */
int orbx_debug_stereo_flags(orbx_extractor* h, int flags);
int orbx_debug_quadtree_profile(orbx_extractor* h, long long out[16]);   /* phase timestamps of one quadtree workgroup (profiling mode 2) */
/* test hook: the largest quadtree (nodes per level) that is given the LDS form; levels above it keep their node lists in the global node pool
 * (the form the 5 x nFeatures extractor of the monocular initialisation takes for its first levels).  0 = every level in the pool; the default and
 * the largest value is 4 000.  Bit-identical outputs. */
int orbx_debug_quadtree_lds_nodes(orbx_extractor* h, int max_nodes);
int orbx_debug_quadtree_pool_levels(orbx_extractor* h);                   /* number of levels (0 .. n-1) the last extraction ran in the pool form */
/* test hook: the byte / packed-16-bit instruction wrappers of the kernels (csrc/orbx_simd.h) applied to n operand triples (n a multiple of 256);
 * out = 22 x n results in the order mul24, mul24 (forced), v_perm_b32, v_alignbyte_b32, v_dot4_u32_u8, v_dot2_u32_u16, packed max3, packed min3,
 * packed sub, packed xor(a, c), wave inclusive scan / wave sum of a & 0xFFFF, wave minimum of b (per 64 consecutive elements), v_sad_u8, v_mul_u32_u24,
 * the 64-bit wave scan (two words), the 64-bit workgroup scan (two words), the wave OR, v_mul_hi_u32_u24, v_add3_u32 */
int orbx_debug_simd_selftest(orbx_extractor* h, const uint32_t* a, const uint32_t* b, const uint32_t* c, int n, uint32_t* out);
/* what the library holds at this moment, process-wide: out = { device allocations, page-locked host allocations, streams, events }.  The lifetime
 * tests create, use and destroy every kind of handle (extractor, key frames, map points, vocabulary, communicator, caller buffers) and expect the
 * four counts back where they started (tests/test_lifetime.py). */
int orbx_debug_live_resources(long long out[4]);

/* ---------------------------------------------------------------------------------------------------------- */
/* ORBmatcher::DescriptorDistance (include/ORBmatcher.h:44, src/ORBmatcher.cc:2383-2403), all pairs:
 * out[i*nb + j] = Hamming(a[i], b[j]).  Host buffers; runs on the handle's GPU. */
int orbm_hamming_matrix(orbx_extractor* h, const uint8_t* a, int na, const uint8_t* b, int nb, int* out);

/* Frame::ComputeStereoMatches (src/Frame.cc:1102-1358) for B rectified pairs: pair p uses image
 * left_first+p of `left`'s last batch and image right_first+p of `right`'s last batch (left == right is allowed:
 * one handle that extracted [L0..LB-1, R0..RB-1]).  bf = mbf, b = mb (include/Frame.h:209-212).  Asynchronous. */
int orbm_stereo_match(orbx_extractor* left, int left_first, orbx_extractor* right, int right_first,
                      int B, float bf, float b);
/* uRight/depth: [B][cap] (mvuRight / mvDepth, -1 = none), n_matches: [B].  Blocking. */
int orbm_stereo_fetch(orbx_extractor* left, int B, float* uRight, float* depth, int cap, int* n_matches);

/* Matching part of Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1553-1562): brute-force Hamming 2-NN of
 * left[monoLeft:] against right[monoRight:] + Lowe ratio (0.7).  Outputs [B][cap], indexed by query row
 * relative to monoLeft; idx relative to monoRight (DMatch.queryIdx/trainIdx).  Blocking fetch. */
int orbm_knn2(orbx_extractor* left, int left_first, orbx_extractor* right, int right_first, int B);
int orbm_knn2_fetch(orbx_extractor* left, int B, int* idx0, int* dist0, int* idx1, int* dist1,
                    uint8_t* ratio_ok, int cap);

/* Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1530-1587) for B fisheye pairs whose two images were extracted with the cameras' lapping
 * areas: brute-force 2-NN + ratio test on the lapping keypoints, then KannalaBrandt8::TriangulateMatches (parallax, DLT triangulation,
 * positive depth in both cameras, reprojection chi-square; src/CameraModels/KannalaBrandt8.cpp:439-523) on every survivor; accepted when
 * the depth is > 1e-4.  Asynchronous; orbm_stereo_fisheye_fetch returns, per pair, mvLeftToRightMatch [cap], mvRightToLeftMatch [cap],
 * mvDepth [cap] (-1 = none), mvStereo3Dpoints [cap][3] and the number of matches.  The null vector of the triangulation matrix is computed
 * in fp64 instead of Eigen's fp32 JacobiSVD: depths agree with the reference to ~1e-6 relative, not bit for bit. */
typedef struct OrbmKB8Stereo {
    float cam1[8], cam2[8];               /* KannalaBrandt8::mvParameters (fx, fy, cx, cy, k0..k3) of mpCamera and mpCamera2 */
    float R12[9], t12[3];                 /* Frame::mRlr (row-major), mtlr */
} OrbmKB8Stereo;
int orbm_stereo_fisheye(orbx_extractor* left, int left_first, orbx_extractor* right, int right_first, int B, const OrbmKB8Stereo* cams);
int orbm_stereo_fisheye_fetch(orbx_extractor* left, int B, int* l2r, int* r2l, float* depth, float* p3d, int* n_matches, int cap);

/* ---------------------------------------------------------------------------------------------------------- */
/* Guided searches.  The reference methods take Frame / KeyFrame / MapPoint objects (pointer graphs with mutexes);
 * across the C ABI they are passed as read-only structure-of-arrays VIEWS of exactly the fields the method reads.
 * The facade (include/orb_slam3_amd/ORBmatcher.h) fills the views from the real classes and applies the results.
 * Geometry that the reference computes with Eigen/Sophus before matching (Tcw * x3Dw, GeometricCamera::project,
 * the fundamental matrix) stays on the caller's side and enters here as numbers, so its float op order is the
 * reference's own.  The functions below are the Frame::Nleft == -1 paths; the two-camera (fisheye rig) branches of the two
 * SearchByProjection overloads are the *_fisheye entry points further down. */
typedef struct OrbmFrameView {            /* fields of ORB_SLAM3::Frame, include/Frame.h */
    int N;                                /* number of keypoints */
    const OrbxKeyPoint* keys_un;          /* mvKeysUn (:232) */
    const uint8_t* desc;                  /* mDescriptors, N x 32 (:244) */
    const float* u_right;                 /* mvuRight, negative = monocular point (:234); NULL = all monocular */
    const uint8_t* occupied;              /* mvpMapPoints[i] != NULL && mvpMapPoints[i]->Observations() > 0 */
    float min_x, min_y, max_x, max_y;     /* mnMinX .. mnMaxY (:287-290) */
    float grid_w_inv, grid_h_inv;         /* mfGridElementWidthInv / HeightInv (:250-251) */
    float mbf;                            /* (:209) */
    int nlevels; const float* scale_factors;   /* mvScaleFactors (:281) */
} OrbmFrameView;

typedef struct OrbmMapPointView {         /* tracking fields of ORB_SLAM3::MapPoint, include/MapPoint.h:171-179 */
    int M;
    const uint8_t* in_view;               /* mbTrackInView */
    const float* proj_x; const float* proj_y; const float* proj_xr;   /* mTrackProjX / Y / XR */
    const int* scale_level;               /* mnTrackScaleLevel */
    const float* view_cos;                /* mTrackViewCos */
    const float* track_depth;             /* mTrackDepth */
    const uint8_t* is_bad;                /* isBad() */
    const uint8_t* has_obs;               /* Observations() > 0 */
    const uint8_t* desc;                  /* GetDescriptor(), M x 32 */
} OrbmMapPointView;

/* Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (src/Frame.cc:859-951): indices in the reference's order.
 * Returns the count (>= 0) or a negative status. */
int orbm_get_features_in_area(orbx_extractor* h, const OrbmFrameView* F, float x, float y, float r,
                              int min_level, int max_level, int* indices, int cap);

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)
 * (src/ORBmatcher.cc:45-239).  assigned[i] = index of the map point written to F.mvpMapPoints[i], or -1 if the call
 * leaves that entry untouched.  *nmatches = the reference's return value. */
int orbm_search_by_projection_mappoints(orbx_extractor* h, const OrbmFrameView* F, const OrbmMapPointView* P,
                                        float th, int far_points, float th_far_points, float nnratio,
                                        int* assigned, int* nmatches);

typedef struct OrbmLastFrameView {        /* what SearchByProjection(CurrentFrame, LastFrame, ...) reads of LastFrame */
    int N;
    const uint8_t* valid;                 /* mvpMapPoints[i] != NULL && !mvbOutlier[i] && invzc >= 0 && projection inside bounds */
    const float* proj_u; const float* proj_v;   /* CurrentFrame.mpCamera->project(Tcw * pMP->GetWorldPos()) */
    const float* inv_z;                   /* 1 / x3Dc(2) */
    const int* octave;                    /* LastFrame.mvKeys[i].octave */
    const float* angle;                   /* LastFrame.mvKeysUn[i].angle */
    const uint8_t* has_obs;               /* pMP->Observations() > 0 */
    const uint8_t* desc;                  /* pMP->GetDescriptor(), N x 32 */
} OrbmLastFrameView;

/* C1 on the device: Frame::isInFrustum (src/Frame.cc:667-773, one camera) + Pinhole / KannalaBrandt8::project + MapPoint::PredictScale
 * (src/MapPoint.cc:688-731) for M map points at once, in the reference's fp32 operation order (mRcw * P + mtcw as a matrix product, as the
 * reference writes it; Eigen >= 3.3 - which the vendored Sophus requires - sums the three terms of a product coefficient, a dot() or a
 * norm() as a0 + (a1 + a2); no fused multiply-adds).  The outputs are the fields the reference stores in the
 * MapPoint: mbTrackInView, mTrackProjX / Y / XR, mTrackDepth, mTrackViewCos, mnTrackScaleLevel (`out` itself is required; any array pointer in
 * it may be NULL).  MapPoint::PredictScale's log is glibc's logf (the reference's std::log(float)), reproduced bit for bit on the device. */
typedef struct OrbmFrustumView {
    float Rcw[9], tcw[3], Ow[3];          /* Frame::mRcw (row-major) = mTcw.rotationMatrix(), mtcw, mOw = mTcw.inverse().translation() (src/Frame.cc:594-598) */
    float qcw[4];                         /* mTcw.unit_quaternion().coeffs() (x, y, z, w): read by orbm_search_by_projection_lastframe_batch only, which
                                           * evaluates `Tcw * x3Dw` (src/ORBmatcher.cc:1987) as Sophus does - on the quaternion, not on mRcw */
    int camera_type;                      /* GeometricCamera::GetType(): 0 pinhole (cam = fx, fy, cx, cy), 1 Kannala-Brandt (8 parameters) */
    float cam[8];
    float min_x, max_x, min_y, max_y;     /* mnMinX .. mnMaxY */
    float mbf, log_scale_factor;          /* mbf, mfLogScaleFactor */
    int nlevels; const float* scale_factors;
} OrbmFrustumView;
typedef struct OrbmWorldPointView {       /* map points by position */
    int M;
    const float* pos;                     /* GetWorldPos(), M x 3 */
    const float* normal;                  /* GetNormal(), M x 3 */
    const float* min_distance;            /* mfMinDistance (GetMinDistanceInvariance() / 0.8) */
    const float* max_distance;            /* mfMaxDistance (GetMaxDistanceInvariance() / 1.2) */
    const uint8_t* is_bad;                /* isBad() (NULL = none) */
    const uint8_t* has_obs;               /* Observations() > 0 (NULL = all) */
    const uint8_t* desc;                  /* GetDescriptor(), M x 32 (searches only) */
} OrbmWorldPointView;
typedef struct OrbmTrackOut { uint8_t* in_view; float *proj_x, *proj_y, *proj_xr, *depth, *view_cos; int* scale_level; } OrbmTrackOut;
int orbm_is_in_frustum(orbx_extractor* h, const OrbmFrustumView* frame, const OrbmWorldPointView* points, float viewing_cos_limit, const OrbmTrackOut* out);
/* isInFrustum for every point, then ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints) (src/ORBmatcher.cc:45-167)
 * on the points in view, as Tracking::SearchLocalPoints does (src/Tracking.cc:4009-4067): the window queries are produced and consumed on
 * the device.  assigned[i] = index of the map point given to keypoint i, or -1; out (may be NULL) receives the tracking fields. */
int orbm_search_local_points(orbx_extractor* h, const OrbmFrameView* F, const OrbmFrustumView* frame, const OrbmWorldPointView* points,
                             float viewing_cos_limit, float th, int far_points, float th_far, float nnratio, const OrbmTrackOut* out,
                             int* assigned, int* nmatches);

/* The same with the map points resident on the device: Tracking::SearchLocalPoints visits the local map (2 000 - 6 000 points,
 * Tracking::UpdateLocalPoints, src/Tracking.cc:4070-4110) every frame, and a map point's position, normal, distance limits and descriptor
 * change only when LocalMapping / LoopClosing touch it.  orbm_points_create uploads those fields of a point set once (is_bad / has_obs of the
 * view are ignored: they are call-time state); orbm_search_local_points_resident then moves the frame, the two flag arrays (NULL: no point bad /
 * every point observed) and the results only.  Same results as orbm_search_local_points. */
typedef struct orbm_points orbm_points;
int orbm_points_create(orbx_extractor* h, const OrbmWorldPointView* points, orbm_points** out);
void orbm_points_destroy(orbm_points* p);
int orbm_search_local_points_resident(orbx_extractor* h, const OrbmFrameView* F, const OrbmFrustumView* frame, const orbm_points* points,
                                      const uint8_t* is_bad, const uint8_t* has_obs, float viewing_cos_limit, float th, int far_points, float th_far,
                                      float nnratio, const OrbmTrackOut* out, int* assigned, int* nmatches);

/* Frame::ComputeStereoFromRGBD (src/Frame.cc:1361-1391) for frames [first, first + B) of the handle's last extraction: mvDepth[i] = the depth
 * image (CV_32F, already scaled by Tracking::mDepthMapFactor, src/Tracking.cc:1617-1618) at the keypoint, mvuRight[i] = x - mbf / d where d > 0,
 * both -1 elsewhere.  depth image b at depth + b * image_stride, rows `stride` floats apart (host memory, or device memory of this GPU).
 * uRight uses mvKeysUn (orbx_set_undistort; mvKeys without distortion), the depth is read at mvKeys, as the reference does.  Results like orbm_stereo_match's: fetch with
 * orbm_stereo_fetch (n_matches = keypoints with a depth), or leave them on the device for orbm_search_local_points_batch.  Asynchronous. */
int orbm_stereo_from_depth(orbx_extractor* h, int first, int B, const float* depth, int stride, size_t image_stride, int depth_on_device, float mbf);

/* Tracking::SearchLocalPoints (src/Tracking.cc:3979-4067 -> Frame::isInFrustum src/Frame.cc:667-773 + ORBmatcher::SearchByProjection
 * src/ORBmatcher.cc:45-167) for a BATCH of frames without leaving the device: frames = images [first, first + B) of the handle's last
 * extraction, read where the extractor left them (mvKeysUn - undistorted on the device when orbx_set_undistort is set -, mDescriptors; mvuRight = the results of
 * orbm_stereo_match / orbm_stereo_from_depth for the same range when use_u_right != 0, else every keypoint is monocular); frames[b] = pose,
 * camera and bounds of frame b (the grid constants are derived from the bounds as Frame does, src/Frame.cc:190-191); the local map is resident
 * (orbm_points); is_bad / has_obs: call-time flags of the M points (NULL = none bad / all observed); occupied: [B][orbx_max_keypoints()] bytes,
 * keypoints that already hold a map point with observations (NULL = none).  The sequential accept loop of the reference (a keypoint that has
 * received a point is skipped by every later point) runs on the device, one wave per frame.  Asynchronous; orbm_search_local_points_fetch
 * returns assigned [B][cap] (index of the map point written to F.mvpMapPoints[i], -1 = untouched), the reference's return value per frame
 * and - when requested here - mbTrackInView [B][M].  ORBX_E_CAPACITY from the fetch = the candidate pool was too small for this scene: it has
 * been enlarged, enqueue the same call again. */
int orbm_search_local_points_batch(orbx_extractor* h, int first, int B, const OrbmFrustumView* frames, const orbm_points* points,
                                   const uint8_t* is_bad, const uint8_t* has_obs, const uint8_t* occupied, int use_u_right,
                                   float viewing_cos_limit, float th, int far_points, float th_far, float nnratio, int want_in_view);
int orbm_search_local_points_fetch(orbx_extractor* h, int* assigned, int cap, int* nmatches, uint8_t* in_view);

/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1950-2184, one camera) - the search of
 * Tracking::TrackWithMotionModel - for a BATCH of frames on the device: current frames = images [first, first + B) of the handle's last
 * extraction (mvKeysUn, mDescriptors, mvuRight as in orbm_search_local_points_batch); cur[b] = pose, camera, bounds, mbf and scale factors of
 * current frame b; `last` = per frame the map points of ITS last frame, cap_last rows per frame: world position, valid
 * (mvpMapPoints[i] != NULL && !mvbOutlier[i]), octave and angle of the last frame's keypoint i, Observations() > 0, descriptor.
 * forward / backward: [B] bytes, bForward / bBackward of each pair of poses (:1973-1975).  Projection, image test, level window, window search,
 * right-coordinate gate, Hamming distances, the sequential accept loop and the rotation histogram with its three maxima all run on the
 * device.  Asynchronous; orbm_search_local_points_fetch returns assigned [B][cap] (index into the last frame's points, -1 untouched, -2 reset
 * to NULL by the rotation check) and the return value per frame.
 * ONE pending batch per handle: orbm_search_local_points_batch and this call share the result block; enqueueing either of them discards a
 * batch that has not been fetched yet, and a call that fails leaves nothing to fetch (orbm_search_local_points_fetch then returns ORBX_E_ARG). */
typedef struct OrbmLastFrameBatch {
    int cap_last;                          /* rows per frame in the arrays below */
    const int* n;                          /* [B] LastFrame.N */
    const float* pos;                      /* [B][cap_last][3] pMP->GetWorldPos() */
    const uint8_t* valid;                  /* [B][cap_last] */
    const int* octave;                     /* [B][cap_last] */
    const float* angle;                    /* [B][cap_last] */
    const uint8_t* has_obs;                /* [B][cap_last] (NULL = all observed) */
    const uint8_t* desc;                   /* [B][cap_last][32] */
} OrbmLastFrameBatch;
int orbm_search_by_projection_lastframe_batch(orbx_extractor* h, int first, int B, const OrbmFrustumView* cur, const OrbmLastFrameBatch* last,
                                              float th, const uint8_t* forward, const uint8_t* backward, int check_orientation,
                                              const uint8_t* occupied, int use_u_right);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:2196-2324) -
 * the search of Tracking::Relocalization (src/Tracking.cc:4480, :4500) - for a BATCH of frames on the device: current frames = images
 * [first, first + B) of the handle's last extraction, cur[b] = pose (quaternion + translation + mOw), camera, bounds and scale factors of frame b;
 * `kf` = per frame the map points of ITS candidate key frame, cap_kf rows per frame: world position, valid (pMP != NULL && !pMP->isBad() &&
 * !sAlreadyFound.count(pMP)), mfMinDistance / mfMaxDistance (the device applies the 0.8 / 1.2 of Get*DistanceInvariance and evaluates
 * MapPoint::PredictScale(dist3D, &CurrentFrame) itself), the angle of the key frame's keypoint i (pKF->mvKeysUn[i].angle), the descriptor.
 * occupied [B][cap] = CurrentFrame.mvpMapPoints[i] != NULL beforehand (NULL = none).  Projection (Tcw * x3Dw on the quaternion), image test, distance
 * range, predicted level, window search, Hamming distances, the sequential accept loop (best distance <= orb_dist; every accepted point occupies its
 * keypoint) and the rotation histogram run on the device.  Asynchronous; orbm_search_local_points_fetch returns assigned [B][cap] (index into the
 * key frame's rows, -1 untouched, -2 reset by the rotation check) and the return value per frame.  Shares the one pending batch of the handle. */
typedef struct OrbmKeyFramePointBatch {
    int cap_kf;                            /* rows per frame in the arrays below */
    const int* n;                          /* [B] pKF->GetMapPointMatches().size() */
    const float* pos;                      /* [B][cap_kf][3] */
    const uint8_t* valid;                  /* [B][cap_kf] */
    const float* min_distance;             /* [B][cap_kf] mfMinDistance */
    const float* max_distance;             /* [B][cap_kf] mfMaxDistance */
    const float* angle;                    /* [B][cap_kf] */
    const uint8_t* desc;                   /* [B][cap_kf][32] */
} OrbmKeyFramePointBatch;
int orbm_search_by_projection_keyframe_batch(orbx_extractor* h, int first, int B, const OrbmFrustumView* cur, const OrbmKeyFramePointBatch* kf,
                                             float th, int orb_dist, int check_orientation, const uint8_t* occupied);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBmatcher.cc:1950-2184).
 * forward/backward = bForward/bBackward (:1973-1975, computed from the two poses by the caller).
 * assigned[i] = index into LastFrame of the point written to CurrentFrame.mvpMapPoints[i]; -1 untouched;
 * -2 = written and then reset to NULL by the rotation-consistency check. */
int orbm_search_by_projection_frame(orbx_extractor* h, const OrbmFrameView* Cur, const OrbmLastFrameView* Last,
                                    float th, int forward, int backward, int check_orientation,
                                    int* assigned, int* nmatches);

typedef struct OrbmKeyFrameView {         /* what SearchForTriangulation reads of a KeyFrame, include/KeyFrame.h */
    int N;
    const OrbxKeyPoint* keys_un;          /* mvKeysUn */
    const uint8_t* desc;                  /* mDescriptors */
    const float* u_right;                 /* mvuRight (NULL = all monocular) */
    const uint8_t* has_map_point;         /* GetMapPoint(i) != NULL */
    int fv_nodes;                         /* mFeatVec (DBoW2::FeatureVector = map<NodeId, vector<unsigned>>) as CSR: */
    const uint32_t* fv_node_id;           /*   node ids, ascending */
    const int* fv_start;                  /*   fv_nodes + 1 offsets into fv_feat */
    const uint32_t* fv_feat;              /*   feature indices, in each node's insertion order */
    int nlevels; const float* scale_factors; const float* level_sigma2;
} OrbmKeyFrameView;

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (src/ORBmatcher.cc:1045-1323),
 * pinhole / no second camera.  F12 = K1^-T [t12]x R12 K2^-1 row-major and ep = the epipole, as the reference computes them
 * (:1061-1063, Pinhole.cpp:186-192).  matches12[i] = idx2 matched to feature i of KF1, or -1. */
int orbm_search_for_triangulation(orbx_extractor* h, const OrbmKeyFrameView* K1, const OrbmKeyFrameView* K2,
                                  const float F12[9], const float ep[2], int only_stereo, int coarse,
                                  int check_orientation, int* matches12, int* nmatches);

/* The same for one key frame against n2 neighbours in one launch (LocalMapping::CreateNewMapPoints loops over 10-30 neighbour key frames,
 * src/LocalMapping.cc:510-540): F12s = n2 x 9, eps = n2 x 2, matches12 = n2 rows of K1->N entries, nmatches = n2 counts. */
int orbm_search_for_triangulation_batch(orbx_extractor* h, const OrbmKeyFrameView* K1, int n2, const OrbmKeyFrameView* const* K2s,
                                        const float* F12s, const float* eps, int only_stereo, int coarse, int check_orientation,
                                        int* matches12, int* nmatches);

/* The same for key frames with Kannala-Brandt cameras - one fisheye camera, or a fisheye rig (mpCamera2 != NULL, features [0, NLeft) from
 * camera 1 and [NLeft, N) from camera 2): the epipolar test is KannalaBrandt8::epipolarConstrain = TriangulateMatches > 1e-4
 * (src/CameraModels/KannalaBrandt8.cpp:322-328), with the relative pose chosen per pair of cameras (src/ORBmatcher.cc:1203-1240).
 * K1 / K2->keys_un hold mvKeys followed by mvKeysRight for a rig (mvKeysUn for one camera), u_right = NULL for a rig. */
typedef struct OrbmKB8Pair {
    int nleft1, nleft2;                   /* KeyFrame::NLeft of KF1 / KF2, -1 = one camera */
    float cam1[2][8], cam2[2][8];         /* mvParameters of mpCamera, mpCamera2 of KF1 and of KF2 (the second is ignored for one camera) */
    float R[4][9], t[4][3];               /* [bRight1 * 2 + bRight2]: rotation (row-major) / translation of Tll, Tlr, Trl, Trr (:1067-1083);
                                             one camera: entry 0 = T12 = T1w * Tw2 */
} OrbmKB8Pair;
int orbm_search_for_triangulation_kb8(orbx_extractor* h, const OrbmKeyFrameView* K1, const OrbmKeyFrameView* K2, const OrbmKB8Pair* cams,
                                      const float ep[2], int only_stereo, int coarse, int check_orientation, int* matches12, int* nmatches);

/* Batched Frame::GetFeaturesInArea + Hamming distance: the building block the remaining projection-type searches of the
 * reference (SearchByProjection(KeyFrame*, Sim3, ...) src/ORBmatcher.cc:495-732, SearchByProjection(Frame&, KeyFrame*, ...) :2196-2324,
 * Fuse :1325-1675, SearchBySim3 :1689-1932) share: for every query (x, y, r, minLevel, maxLevel, descriptor) the keypoints
 * of F in the window, in the reference's GetFeaturesInArea order, with their distance to the query descriptor and their octave.
 * None of those callers has an order-dependent accept loop, so they take the minimum over their own gates on these lists.
 * CSR output: start[Q], count[Q]; entry k of query q is idx[start[q]+k], dist[...], level[...].  `cap` = capacity of the entry
 * arrays; returns the total number of entries (> cap: nothing is copied, call again with a larger cap). */
typedef struct OrbmAreaQuery { float x, y, r; int min_level, max_level; } OrbmAreaQuery;
int orbm_area_search_batch(orbx_extractor* h, const OrbmFrameView* F, const OrbmAreaQuery* queries, const uint8_t* query_desc, int Q,
                           int* start, int* count, int* idx, int* dist, int* level, int cap);

/* ---- Input pre-step: what the reference does to a camera frame between the driver and ORBextractor ----
 * geometry 1 = cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR), the stereo rectification of System::TrackStereo (src/System.cc:286-293,
 * maps of cv::initUndistortRectifyMap(..., CV_32F, ...), src/Settings.cc:549-574); geometry 2 = cv::resize(im, imToFeed, newImSize)
 * (src/System.cc:295-297, e.g. 752x480 -> 600x350 in Examples/Monocular/EuRoC.yaml:36-37); then, for 3/4-channel frames,
 * cv::cvtColor(..., COLOR_RGB2GRAY | BGR2GRAY | RGBA2GRAY | BGRA2GRAY) of Tracking::GrabImage* (src/Tracking.cc:1532-1560, rgb = mbRGB).
 * With a spec set, orbx_extract / orbx_extract_batch take the raw frames (width, height, stride of the source, `channels` bytes per
 * pixel) and extract at out_w x out_h (geometry != 0) or at the source size.  NULL restores plain 8UC1 input. */
typedef struct OrbxInputSpec {
    int channels;                        /* 1, 3 or 4 */
    int rgb;                             /* Tracking::mbRGB: 1 = red first, 0 = blue first */
    int gray_variant;                    /* cvtColor 8U coefficients: 0 = OpenCV 4.x (9798, 19235, 3735 >> 15), 1 = 3.x (4899, 9617, 1868 >> 14) */
    int geometry;                        /* 0 none, 1 remap, 2 resize */
    int out_w, out_h;                    /* size fed to the extractor when geometry != 0 (map size / Settings::newImSize()) */
    const float* map_x; const float* map_y;   /* geometry 1: M1, M2 (CV_32FC1, out_h x out_w, dense); copied to the device */
} OrbxInputSpec;
int orbx_set_input(orbx_extractor* h, const OrbxInputSpec* spec);

/* ---- two-camera (fisheye rig, Frame::Nleft != -1) branches of the SearchByProjection overloads ----
 * left  = the view of camera 1: keys_un = mvKeys, desc = descriptor rows [0, Nleft), occupied = mvpMapPoints[0, Nleft) (with observations);
 * right = the view of camera 2: keys_un = mvKeysRight, desc = rows [Nleft, N), occupied = mvpMapPoints[Nleft, N); u_right is not read.
 * Both views carry the frame's bounds / grid parameters (the right grid is mGridRight). */
typedef struct OrbmFisheyeFrameView {
    OrbmFrameView left, right;
    const int* left_to_right;             /* mvLeftToRightMatch [Nleft]  (read by the map-point overload only; NULL = all -1) */
    const int* right_to_left;             /* mvRightToLeftMatch [Nright] (read by the map-point overload only; NULL = all -1) */
} OrbmFisheyeFrameView;
typedef struct OrbmMapPointRightView {    /* the *R tracking fields of MapPoint, include/MapPoint.h:175-179 */
    const uint8_t* in_view_r;             /* mbTrackInViewR */
    const float* proj_xr; const float* proj_yr;   /* mTrackProjXR / mTrackProjYR */
    const int* scale_level_r;             /* mnTrackScaleLevelR (-1 = not set) */
    const float* view_cos_r;              /* mTrackViewCosR */
} OrbmMapPointRightView;
/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) with F.Nleft != -1
 * (src/ORBmatcher.cc:45-239 including :170-236).  assigned has Nleft + Nright entries, indexed like F.mvpMapPoints. */
int orbm_search_by_projection_mappoints_fisheye(orbx_extractor* h, const OrbmFisheyeFrameView* F, const OrbmMapPointView* P,
                                                const OrbmMapPointRightView* PR, float th, int far_points, float th_far_points,
                                                float nnratio, int* assigned, int* nmatches);
/* Frame::isInFrustum for a two-camera rig (Nleft != -1, src/Frame.cc:754-766): Frame::isInFrustumChecks (:1592-1650) once per camera on the
 * device.  `left` = camera 1 as in OrbmFrustumView (mRcw, mtcw, mOw, mpCamera, the image bounds); camera 2 is derived exactly as the reference
 * derives it: mR = Rrl * mRcw, mt = Rrl * mtcw + trl, twc = mRwc * tlr + mOw (Eigen >= 3.3's sums: a0 + (a1 + a2)), projected with mpCamera2.  A point
 * that fails any test of a camera leaves that camera's fields untouched in the reference; here its in-view flag is 0, its level -1 and the
 * other fields unspecified. */
typedef struct OrbmFrustumRigView {
    OrbmFrustumView left;                 /* camera 1 */
    float Rrl[9], trl[3];                 /* mTrl.rotationMatrix() (row-major), mTrl.translation() */
    float tlr[3];                         /* mTlr.translation() */
    float Rwc[9];                         /* mRwc (row-major) */
    int camera2_type; float cam2[8];      /* mpCamera2: GetType(), parameters */
} OrbmFrustumRigView;
typedef struct OrbmTrackOutRight { uint8_t* in_view_r; float *proj_xr, *proj_yr, *depth_r, *view_cos_r; int* scale_level_r; } OrbmTrackOutRight;   /* the *R fields, include/MapPoint.h:175-179 */
int orbm_is_in_frustum_rig(orbx_extractor* h, const OrbmFrustumRigView* frame, const OrbmWorldPointView* points, float viewing_cos_limit,
                           const OrbmTrackOut* left, const OrbmTrackOutRight* right);
/* Tracking::SearchLocalPoints for a rig frame (src/Tracking.cc:3979-4067 with Nleft != -1): the two-camera isInFrustum on the device, then
 * ORBmatcher::SearchByProjection(F, points, th, bFarPoints, thFarPoints) with its right-camera branch (src/ORBmatcher.cc:45-239).  assigned has
 * Nleft + Nright entries; left / right (may be NULL) receive the tracking fields. */
int orbm_search_local_points_fisheye(orbx_extractor* h, const OrbmFisheyeFrameView* F, const OrbmFrustumRigView* frame, const OrbmWorldPointView* points,
                                     float viewing_cos_limit, float th, int far_points, float th_far, float nnratio, const OrbmTrackOut* left,
                                     const OrbmTrackOutRight* right, int* assigned, int* nmatches);
/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) with CurrentFrame.Nleft != -1
 * (src/ORBmatcher.cc:1950-2184 including :2090-2150).  proj_ur / proj_vr [Last->N] = project(GetRelativePoseTrl() * x3Dc).
 * assigned has Nleft + Nright entries (-1 untouched, -2 reset by the rotation check). */
int orbm_search_by_projection_frame_fisheye(orbx_extractor* h, const OrbmFisheyeFrameView* Cur, const OrbmLastFrameView* Last,
                                            const float* proj_ur, const float* proj_vr, float th, int forward, int backward,
                                            int check_orientation, int* assigned, int* nmatches);

/* ---- the geometry in front of the projection-type searches, on the device ----
 * SearchByProjection(Frame, LastFrame) (src/ORBmatcher.cc:1993-2010), (Frame, KeyFrame) (:2228-2256), (KeyFrame, Sim3, ...) x2 (:525-560, :640-690),
 * Fuse x2 (:1388-1430, :1590-1625) and SearchBySim3 (:1745-1790, :1830-1875) all start with the same per-map-point chain: transform, depth test,
 * projection, image test, distance range, viewing angle.  orbm_project_points evaluates it for M points at once in the reference's fp32
 * operation order.  The reference multiplies `Tcw * p3Dw` with Sophus types, and Sophus rotates a point by the unit quaternion WITHOUT forming
 * the matrix (Thirdparty/Sophus/sophus/so3.hpp:357-367: uv = 2 q.vec x p; p + q.w uv + q.vec x uv; se3.hpp:321-324 adds the translation;
 * rxso3.hpp:265-273 / sim3.hpp:226-229 for a similarity): poses therefore enter as quaternion coefficients in Eigen's coeffs() order
 * (x, y, z, w) + translation, and the device evaluates exactly those statements (csrc/sophus_action.h; no fused multiply-adds).  R * p + t
 * with R = q.toRotationMatrix() rounds differently in the last bit, which is enough to move a keypoint across a search-window edge.
 * The spec says which tests the method at hand applies and how it writes its projection.  MapPoint::PredictScale stays with the caller's
 * MapPoint (it reads a protected member): `dist` is its argument.  Blocking. */
typedef struct OrbmProjection {
    float q[4], t[3];                     /* p1 = T * p_w: the SE3 pose in front (Tcw, T1w, ...): T.unit_quaternion().coeffs() (x, y, z, w), T.translation() */
    int second; float q2[4], t2[3], s2;   /* a second transform behind it, p_c = X * p1: 0 none; 1 a Sim3 (SearchBySim3's S21 / S12): q2 = X.quaternion().coeffs()
                                           * (not unit: |q2|^2 = scale), t2 = X.translation(), s2 = X.scale(); 2 an SE3 (GetRelativePoseTrl()): unit q2, t2 */
    float Ow[3];                          /* camera centre for the distance / angle tests (dist_mode 0) */
    int dist_mode;                        /* 0: dist = |p_w - Ow|, 1: dist = |p_c| (SearchBySim3, :1771) */
    int depth_test;                       /* 0 none (:2228-2256), 1: p_c.z < 0 rejects, 2: 1 / p_c.z < 0 rejects (:1997-2000) */
    int camera_type; float cam[8];        /* GeometricCamera::GetType(): 0 pinhole (fx, fy, cx, cy), 1 Kannala-Brandt */
    int inline_pinhole;                   /* 1: x = X * (1 / Z), u = fx * x + cx as written out in :656-660 / :1760-1766 instead of mpCamera->project */
    float min_x, max_x, min_y, max_y;     /* mnMinX .. mnMaxY */
    int bounds_mode;                      /* 0: the Frame test (u < min || u > max rejects), 1: KeyFrame::IsInImage (u >= min && u < max), 2: none */
    int distance_test;                    /* dist outside [min_inv, max_inv] rejects */
    int angle_test;                       /* PO . Pn < 0.5 dist rejects (:551, :1416, :1612) */
    float bf;                             /* ur = u - bf / z (Fuse, :1400); 0 otherwise */
} OrbmProjection;
typedef struct OrbmProjectIn {
    int M;
    const float* pos; const float* normal;            /* GetWorldPos(), GetNormal(): M x 3 (normal may be NULL without the angle test) */
    const float* min_inv; const float* max_inv;       /* GetMinDistanceInvariance(), GetMaxDistanceInvariance() (NULL without the distance test) */
    const uint8_t* skip;                              /* 1 = rejected by the caller's own tests (NULL = none) */
} OrbmProjectIn;
typedef struct OrbmProjectOut { uint8_t* valid; float *u, *v, *ur, *inv_z, *dist; } OrbmProjectOut;   /* any array may be NULL */
int orbm_project_points(orbx_extractor* h, const OrbmProjection* spec, const OrbmProjectIn* in, const OrbmProjectOut* out);

/* ---- remaining projection-type searches ----
 * The caller evaluates the geometry in front of GetFeaturesInArea with the reference's own Sophus/Eigen code (Tcw * p3Dw, project,
 * IsInImage, min/max distance, viewing angle, PredictScale) and hands the survivors over as numbers; window search, level window,
 * chi-square gate and every Hamming distance run on the device, the (order-dependent) accept loop is replayed in order. */
typedef struct OrbmProjectedPointView {
    int M;
    const uint8_t* valid;                 /* 1 = the point reaches GetFeaturesInArea in the reference's loop */
    const float* u; const float* v;       /* its projection */
    const float* ur;                      /* uv(0) - bf*invz: read by the Fuse chi-square gate only (may be NULL otherwise) */
    const int* pred_level;                /* pMP->PredictScale(dist, pKF) */
    const float* angle;                   /* pKF->mvKeysUn[i].angle: read by orbm_search_by_projection_keyframe with check_orientation only */
    const uint8_t* desc;                  /* pMP->GetDescriptor(), M x 32 */
} OrbmProjectedPointView;

/* ORBmatcher::SearchByProjection(KeyFrame* pKF, Sim3f& Scw, vpPoints, vpMatched, th, ratioHamming) (src/ORBmatcher.cc:495-606) and
 * the overload that also returns the key frame of each point (:608-732; vpMatchedKF[idx] = vpPointsKFs[assigned[idx]]).
 * KF->occupied[idx] = (vpMatched[idx] != NULL) on entry.  assigned[idx] = index of the point written to vpMatched[idx], else -1. */
int orbm_search_by_projection_sim3(orbx_extractor* h, const OrbmFrameView* KF, const OrbmProjectedPointView* P, float th,
                                   float ratio_hamming, int* assigned, int* nmatches);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:2196-2324).
 * Cur->occupied[i] = (CurrentFrame.mvpMapPoints[i] != NULL); P has one entry per feature of pKF.  assigned[i] = index into pKF of the
 * map point written to CurrentFrame.mvpMapPoints[i]; -1 untouched; -2 written and reset by the rotation-consistency check. */
int orbm_search_by_projection_keyframe(orbx_extractor* h, const OrbmFrameView* Cur, const OrbmProjectedPointView* P, float th,
                                       int orb_dist, int check_orientation, int* assigned, int* nmatches);

/* Candidate search of ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight) (src/ORBmatcher.cc:1325-1528; chi2_gate = 1, inv_level_sigma2 =
 * pKF->mvInvLevelSigma2) and of Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:1543-1660; chi2_gate = 0).  best_idx[i] = the key-frame
 * feature the reference would fuse point i with (bestDist <= TH_LOW), else -1; best_dist may be NULL.  The map surgery that follows
 * (Replace / AddObservation / AddMapPoint) mutates the caller's map and is done by the caller, in point order, from best_idx.
 * For bRight build KF from mvKeysRight / the right descriptors and add NLeft to the returned indices. */
int orbm_fuse_candidates(orbx_extractor* h, const OrbmFrameView* KF, const OrbmProjectedPointView* P, float th, int chi2_gate,
                         const float* inv_level_sigma2, int* best_idx, int* best_dist);

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (src/ORBmatcher.cc:1689-1932).  P1in2 has one entry per feature of KF1 (its
 * map point transformed by S21 and projected into KF2; valid = present, good, not already matched, passes the depth / image / distance
 * tests), P2in1 the reverse.  matches12[i1] = idx2 of the mutually consistent new match, else -1; *nfound = the reference's return value. */
int orbm_search_by_sim3(orbx_extractor* h, const OrbmFrameView* KF1, const OrbmFrameView* KF2, const OrbmProjectedPointView* P1in2,
                        const OrbmProjectedPointView* P2in1, float th, int* matches12, int* nfound);

/* ---- "next" rows of SURVEY.md §8f, built on the same kernels ---- */
/* ORBmatcher::SearchByBoW.  K1 = the key frame whose map points are searched (has_map_point[i] = map point present and not bad),
 * K2 = the other side.  th_inclusive = 1: SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:259-493: accept
 * bestDist1 <= TH_LOW, every frame feature is a candidate — K2->has_map_point may be NULL).  th_inclusive = 0:
 * SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (:892-1043: bestDist1 < TH_LOW, only K2 features with a good map point).
 * matches12[i] = index in K2 matched to feature i of K1, or -1 (after the rotation-consistency pruning). */
int orbm_search_by_bow(orbx_extractor* h, const OrbmKeyFrameView* K1, const OrbmKeyFrameView* K2, float nnratio, int th_inclusive,
                       int check_orientation, int* matches12, int* nmatches);

/* n independent (K1s[p], K2s[p]) pairs in one launch: every relocalisation candidate against the current frame (src/Tracking.cc:4360-4380),
 * a key frame against its loop / merge candidates (src/LoopClosing.cc:840-850).  matches12[p] has K1s[p]->N entries, nmatches n entries. */
int orbm_search_by_bow_batch(orbx_extractor* h, int n, const OrbmKeyFrameView* const* K1s, const OrbmKeyFrameView* const* K2s, float nnratio,
                             int th_inclusive, int check_orientation, int* const* matches12, int* nmatches);

/* Device-resident key frames.  What the vocabulary-bucket searches read of a key frame or frame - mvKeysUn, mDescriptors, mvuRight, mFeatVec
 * (include/KeyFrame.h:  the fields of OrbmKeyFrameView) - is uploaded once by orbm_keyframe_create (has_map_point of the view is ignored:
 * map points change while a key frame lives, the searches take the flags per call).  The resident searches give the same results as
 * orbm_search_for_triangulation_batch / orbm_search_by_bow_batch and move only flags, poses and results across the bus; the accept loop of
 * SearchByBoW runs on the device (one wave per vocabulary node).  A key frame may be used with any
 * extractor handle of the device it was created on.  Limit: at most 2048 features of one key frame per vocabulary node (ORBX_E_CAPACITY). */
typedef struct orbm_keyframe orbm_keyframe;
int orbm_keyframe_create(orbx_extractor* h, const OrbmKeyFrameView* K, orbm_keyframe** out);
void orbm_keyframe_destroy(orbm_keyframe* kf);
/* SearchForTriangulation of K1 against n2 neighbours (src/ORBmatcher.cc:1045-1323).  has_map_point1 [K1 N] / has_map_point2[j] [K2s[j] N]:
 * GetMapPoint(i) != NULL at call time (NULL = none).  Other arguments and results as orbm_search_for_triangulation_batch. */
int orbm_search_for_triangulation_resident(orbx_extractor* h, orbm_keyframe* K1, const uint8_t* has_map_point1, int n2, orbm_keyframe* const* K2s,
                                           const uint8_t* const* has_map_point2, const float* F12s, const float* eps, int only_stereo, int coarse,
                                           int check_orientation, int* matches12, int* nmatches);
/* The same for key frames with Kannala-Brandt cameras (one fisheye camera or the rig; the views given to orbm_keyframe_create list mvKeys followed
 * by mvKeysRight and no u_right for a rig, as for orbm_search_for_triangulation_kb8): cams[j] describes the pair (K1, K2s[j]). */
int orbm_search_for_triangulation_resident_kb8(orbx_extractor* h, orbm_keyframe* K1, const uint8_t* has_map_point1, int n2, orbm_keyframe* const* K2s,
                                               const uint8_t* const* has_map_point2, const OrbmKB8Pair* cams, const float* eps, int only_stereo, int coarse,
                                               int check_orientation, int* matches12, int* nmatches);
/* SearchByBoW for n pairs (src/ORBmatcher.cc:259-493, :892-1043).  has_map_point1[p]: K1s[p] features with a good map point (NULL = none, no
 * matches); eligible2[p]: K2s[p] features that may be matched (NULL = all: the Frame overload).  Results as orbm_search_by_bow_batch. */
int orbm_search_by_bow_resident(orbx_extractor* h, int n, orbm_keyframe* const* K1s, const uint8_t* const* has_map_point1, orbm_keyframe* const* K2s,
                                const uint8_t* const* eligible2, float nnratio, int th_inclusive, int check_orientation, int* const* matches12,
                                int* nmatches);

/* SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) for a fisheye-rig frame (F.Nleft != -1; src/ORBmatcher.cc:259-493 incl. :343-372, :414-446).
 * K1 / K2 list ALL features by index (camera 1 first: keys = mvKeys followed by mvKeysRight, descriptor rows as stored); nleft2 = F.Nleft.
 * assigned2[j] = feature of K1 whose map point is written to vpMapPointMatches[j], -1 = NULL (after the rotation-consistency pruning). */
int orbm_search_by_bow_fisheye(orbx_extractor* h, const OrbmKeyFrameView* K1, const OrbmKeyFrameView* K2, int nleft2, float nnratio,
                               int check_orientation, int* assigned2, int* nmatches);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBmatcher.cc:734-880).
 * prev_matched: N1 x 2 floats (vbPrevMatched), updated in place like the reference does (:876-878). */
int orbm_search_for_initialization(orbx_extractor* h, const OrbmFrameView* F1, const OrbmFrameView* F2, float* prev_matched,
                                   int window_size, float nnratio, int check_orientation, int* matches12, int* nmatches);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:438-529) for P map points at once.  Point p owns the
 * descriptors desc[32*start[p] .. 32*start[p+1]) — its observations' descriptors in the reference's iteration order (:451-470).
 * best[p] = index, relative to start[p], of the descriptor the reference would keep (smallest median distance to the others, first on
 * ties); -1 for a point without descriptors. */
int orbm_distinctive_descriptors(orbx_extractor* h, const uint8_t* desc, const int* start, int P, int* best);

/* ---- Vocabulary: the consumer right behind the extractor, Frame::ComputeBoW (src/Frame.cc:984-997) ->
 * ORBVocabulary::transform(features, mBowVec, mFeatVec, 4) (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1195).  Produces the
 * FeatureVector CSR that OrbmKeyFrameView (SearchByBoW / SearchForTriangulation) consumes. ---- */
typedef struct orbv_vocabulary orbv_vocabulary;

/* Vocabulary from flat arrays in ORBvoc.txt line order (loadFromTextFile, TemplatedVocabulary.h:1338-1430): entry i describes node
 * i + 1; parent[i] (0 = root) must be an earlier node; is_leaf[i] = the file's leaf flag (word ids are handed out to flagged nodes in
 * line order); desc[32*i..]; weight[i].  scoring / weighting = DBoW2::ScoringType / WeightingType (BowVector.h:37-56).
 * The vocabulary lives on h's device. */
int orbv_create(orbx_extractor* h, int k, int L, int scoring, int weighting, int n_nodes, const int* parent, const uint8_t* is_leaf,
                const uint8_t* desc, const double* weight, orbv_vocabulary** out);
/* ORBVocabulary::loadFromTextFile(path) */
int orbv_load_text(orbx_extractor* h, const char* path, orbv_vocabulary** out);
void orbv_destroy(orbv_vocabulary* v);
int orbv_words(const orbv_vocabulary* v);                                       /* ORBVocabulary::size() */

/* transform(features, BowVector&, FeatureVector&, levelsup) for n host descriptors (n x 32).  Any output pointer may be NULL.
 * word_id/node_id [n]: per-feature word and ancestor at level L - levelsup (transform(feature, id, weight, &nid, levelsup), :1218-1259).
 * One case differs from the reference by necessity: a word whose leaf lies ABOVE level L - levelsup (a ragged vocabulary) leaves the reference's `nid` an
 * uninitialised local (:1150, undefined); here the node is then the leaf itself.  ORBvoc.txt at levelsup = 4 has no such leaf.
 * BowVector: bow_id/bow_val (capacity n), ascending word ids, *n_bow entries.  FeatureVector as CSR: fv_node (capacity n, ascending),
 * fv_start (capacity n + 1), fv_feat (capacity n; feature indices in insertion order), *n_fv nodes. */
int orbv_transform(orbv_vocabulary* v, orbx_extractor* h, const uint8_t* desc, int n, int levelsup, uint32_t* word_id, uint32_t* node_id,
                   uint32_t* bow_id, double* bow_val, int* n_bow, uint32_t* fv_node, int* fv_start, uint32_t* fv_feat, int* n_fv);
/* The same for images [first, first + B) of h's last orbx_extract_batch, on the descriptors that are still on the device (no copy);
 * asynchronous on h's stream.  orbv_fetch copies the vectors of image b (relative to `first`) out; capacities = orbx_max_keypoints(h). */
int orbv_transform_extracted(orbv_vocabulary* v, orbx_extractor* h, int first, int B, int levelsup);
int orbv_fetch(orbv_vocabulary* v, orbx_extractor* h, int b, uint32_t* word_id, uint32_t* node_id, int n_features, uint32_t* bow_id, double* bow_val,
               int* n_bow, uint32_t* fv_node, int* fv_start, uint32_t* fv_feat, int* n_fv);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vpMapPointMatches) (src/ORBmatcher.cc:259-493; Tracking::TrackReferenceKeyFrame
 * src/Tracking.cc:3183, Relocalization :4371) for a BATCH of frames on the device: frame b = image first + b of h's last extraction, whose
 * FeatureVector is read where orbv_transform_extracted(v, h, first, B, levelsup) left it (call that first, on the same images); KFs[b] = the
 * device-resident key frame it is searched against (one key frame may serve many frames); has_map_point[b] [KFs[b]->N] = its features with a good
 * map point.  The accept loop per vocabulary node, the ratio and TH_LOW tests and the rotation histogram with its three maxima run on the device
 * (two launches per batch).  matches12[b][i] = frame feature matched to key-frame feature i, or -1, i.e. vpMapPointMatches[matches12[b][i]] =
 * the key frame's map point i; nmatches[b] = the reference's return value.  Blocking. */
int orbm_search_by_bow_frames_batch(orbx_extractor* h, const orbv_vocabulary* v, int first, int B, orbm_keyframe* const* KFs,
                                    const uint8_t* const* has_map_point, float nnratio, int check_orientation, int* const* matches12, int* nmatches);

const char* orbx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
