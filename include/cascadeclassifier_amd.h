/*
 * cascadeclassifier_amd.h — C ABI of the MI355X-native cascade-classifier hot path.
 *
 * Drop-in boundary for ONE path of vladiant/CascadeClassifier: integral-image construction and
 * Haar / LBP per-window feature evaluation over the sliding-window / scale pyramid, as used by the
 * detection tool and by CvCascadeBoost stage training. Everything behind these entry points runs as
 * hand-written HIP kernels on gfx950; there is NO CPU fallback: without a usable HIP device every
 * compute entry point fails with CC_ERR_NO_DEVICE.
 *
 * Each group names the reference interface it replaces (paths relative to the reference repo root).
 *
 * Conventions: plain C types; opaque handles; caller-owned buffers; every function returns a
 * cc_status (0 = OK, < 0 = error) unless documented otherwise; cc_last_error() returns a
 * thread-local, human-readable message for the last failing call on the calling thread. No
 * exceptions cross this boundary. Handles are thread-compatible (one thread at a time per handle);
 * cc_eval_calc* are safe for concurrent callers once cc_eval_set_image(s) has returned, as the
 * reference's const operator() is (traincascade/lib/src/o_cvcascadeboosttraindata.cpp:586-594).
 */
#ifndef CASCADECLASSIFIER_AMD_H_
#define CASCADECLASSIFIER_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define CC_API __attribute__((visibility("default")))
#else
#define CC_API
#endif

typedef int cc_status;
enum {
  CC_OK = 0,
  CC_ERR_INVALID_ARG = -1,      /* CV_Assert-class failures in the reference (features.cpp:75,85-87) */
  CC_ERR_NO_DEVICE = -2,        /* HIP runtime / gfx950 device not usable: the product has no CPU path */
  CC_ERR_HIP = -3,              /* a HIP call failed; message carries hipGetErrorString */
  CC_ERR_IO = -4,               /* file cannot be opened / read */
  CC_ERR_PARSE = -5,            /* malformed cascade XML */
  CC_ERR_UNSUPPORTED = -6,      /* valid input the GPU path does not implement yet (message says which) */
  CC_ERR_BUFFER_TOO_SMALL = -7, /* output capacity too small; the required size is reported */
  CC_ERR_OUT_OF_RANGE = -8
};

CC_API const char* cc_last_error(void);
CC_API int cc_version(void);
/* Number of usable HIP devices (0 when there is none or the runtime cannot initialise). */
CC_API int cc_device_count(void);

typedef struct cc_rect {
  int32_t x, y, width, height;
} cc_rect;

/* ============================================================================================
 * 1. Cascade model (cascade.xml, new format).
 *    Replaces: cv::CascadeClassifier(const String&) / load() / empty() as called at
 *    tools/detection/Cpp/main.cpp:42 and tools/detection/Python/detect.py:16; the format is the one
 *    written by CvCascadeClassifier::save (traincascade/lib/src/cascadeclassifier.cpp:439-456,
 *    tags traincascade/lib/include/cascadeclassifier.h:27-73).
 * ============================================================================================ */
typedef struct cc_cascade cc_cascade;

enum { CC_FEATURE_HAAR = 0, CC_FEATURE_LBP = 1, CC_FEATURE_HOG = 2 };

typedef struct cc_cascade_info {
  int32_t feature_type; /* CC_FEATURE_* */
  int32_t win_w, win_h;
  int32_t n_stages, n_weak, n_nodes, n_leaves, n_features;
  int32_t max_cat_count, subset_size;
  int32_t max_nodes_per_tree; /* 1 => stump cascade */
  int32_t has_tilted;
} cc_cascade_info;

CC_API cc_status cc_cascade_load_xml(const char* path, cc_cascade** out);
CC_API cc_status cc_cascade_load_xml_mem(const char* text, size_t len, cc_cascade** out);
CC_API void cc_cascade_destroy(cc_cascade* c);
CC_API cc_status cc_cascade_info_get(const cc_cascade* c, cc_cascade_info* info);
/* Flat views of the parsed model (for inspection / parity tests). Arrays are owned by the cascade. */
CC_API cc_status cc_cascade_stages(const cc_cascade* c, const int32_t** first_weak, const int32_t** n_weak,
                                   const float** threshold /* (float)stageThreshold - 1e-5f */);
/* Stump view (valid when max_nodes_per_tree == 1): per weak classifier feature index, threshold, leaf values and
 * (LBP) subset_size int32 words of category mask. */
CC_API cc_status cc_cascade_stumps(const cc_cascade* c, const int32_t** feature_idx, const float** threshold,
                                   const float** left, const float** right, const int32_t** subsets);
/* Haar: rects as int32[n_features][3][4] (x y w h), weights float[n_features][3], tilted int32[n_features].
 * LBP:  rects as int32[n_features][4]; weights / tilted are NULL. */
CC_API cc_status cc_cascade_features(const cc_cascade* c, const int32_t** rects, const float** weights,
                                     const int32_t** tilted);

/* Writes the model back as a new-format cascade.xml (the layout of CvCascadeClassifier::save,
 * traincascade/lib/src/cascadeclassifier.cpp:439-456; stages boost.cpp:520-532, trees o_cvcascadeboosttree.cpp:41-93,
 * features haarfeatures.cpp:311-320 / lbpfeatures.cpp:65-68). Reading the written file yields an identical model. */
CC_API cc_status cc_cascade_save_xml(const cc_cascade* c, const char* path);

/* The legacy "baseFormat" layout of CvCascadeClassifier::save(filename, true) (cascadeclassifier.cpp:421-437 tag names,
 * :457-531 writer): <cascade type_id="opencv-haar-classifier"> with <size>, per stage <trees> (nodes breadth-first from
 * the root, inner children numbered as they are queued; <feature> = Feature::write, haarfeatures.cpp:311-320; <threshold>,
 * <left_val>/<left_node>, <right_val>/<right_node>), <stage_threshold>, <parent> = stage - 1, <next> = -1.
 * Haar cascades only: anything else returns CC_ERR_UNSUPPORTED with the reference's message (:461-462). Write-only, as
 * in the reference; cc_cascade_load_xml refuses this layout. */
CC_API cc_status cc_cascade_save_xml_legacy(const cc_cascade* c, const char* path);

/* .vec sample files (positives of the trainer): header int32 count, int32 width*height, 2 x int16 0; per sample one zero
 * byte + width*height int16 pixels. Reader: CvCascadeImageReader::PosReader (traincascade/lib/src/imagestorage.cpp:138-182);
 * writer: icvWriteVecHeader / icvWriteVecSample (tools/createsamples/utility.cpp:128-152). Host-side file IO.
 * cc_vec_read: *count / *vec_size are always set when the header parses; pixels (count * vec_size bytes, may be NULL to
 * query the sizes) receives at most cap_samples samples, each pixel narrowed to 8 bits as the reader does. */
CC_API cc_status cc_vec_read(const char* path, int32_t* count, int32_t* vec_size, uint8_t* pixels, int cap_samples);
CC_API cc_status cc_vec_write(const char* path, const uint8_t* pixels, int count, int width, int height);

/* ============================================================================================
 * 2. Detector.
 *    Replaces: cv::CascadeClassifier::detectMultiScale(gray, objects, scaleFactor, minNeighbors, flags,
 *    minSize, maxSize) as called at tools/detection/Cpp/main.cpp:45 (gray, objects, 4, 50) and
 *    tools/detection/Python/detect.py:22; internally cv::resize(INTER_LINEAR_EXACT), cv::integral and
 *    cv::groupRectangles (OpenCV 4.6.0, external/CMakeLists.txt:11).
 * ============================================================================================ */
typedef struct cc_detector cc_detector;

typedef struct cc_detect_params {
  double scale_factor;   /* > 1 */
  int32_t min_neighbors; /* groupRectangles threshold; <= 0 returns ungrouped candidates */
  int32_t min_w, min_h;  /* 0 = no limit */
  int32_t max_w, max_h;  /* 0 = image size */
} cc_detect_params;

/* device: HIP device ordinal. max_batch: frames processed per pass (workspace is sized for it). */
CC_API cc_status cc_detector_create(const cc_cascade* c, int device, int max_batch, cc_detector** out);
CC_API void cc_detector_destroy(cc_detector* d);
/* Use the caller's HIP stream (hipStream_t) for all work of this detector; NULL = detector's own stream. */
CC_API cc_status cc_detector_set_stream(cc_detector* d, void* hip_stream);

/* One frame, host memory. out receives at most cap rectangles; *n = number found (if *n > cap the call returns
 * CC_ERR_BUFFER_TOO_SMALL and fills the first cap). Rectangle order: class order of cv::groupRectangles for
 * candidates sorted (scale, y, x), i.e. OpenCV's single-threaded order. */
CC_API cc_status cc_detect_multiscale(cc_detector* d, const uint8_t* gray, int width, int height, size_t row_stride,
                                      const cc_detect_params* p, cc_rect* out, int cap, int* n);

/* Batch of equally sized frames. frames points to HOST memory (on_device = 0) or to DEVICE memory of the detector's
 * device (on_device = 1; e.g. a torch tensor's data_ptr): frame f starts at frames + f * frame_stride.
 * Output: rectangles of frame f are out[offsets[f] .. offsets[f+1]); offsets has n_frames + 1 entries. */
CC_API cc_status cc_detect_batch(cc_detector* d, const uint8_t* frames, int on_device, int n_frames, int width,
                                 int height, size_t row_stride, size_t frame_stride, const cc_detect_params* p,
                                 cc_rect* out, int cap, int32_t* offsets);

/* Same, but only the device pipeline (pyramid, integrals, cascade evaluation, skip-rule filter): candidates stay on
 * the device, nothing is copied back or grouped. For benchmarking the kernels. */
CC_API cc_status cc_detect_batch_device_only(cc_detector* d, const uint8_t* frames, int on_device, int n_frames,
                                             int width, int height, size_t row_stride, size_t frame_stride,
                                             const cc_detect_params* p);

/* cc_detect_batch in two halves, for streams of batches (video): submit launches the batch and returns while its last
 * pass still runs; collect waits for it and returns what cc_detect_batch would have returned (same rectangles, same
 * order, same status codes). Submitting batch k + 1 BEFORE collecting batch k lets the pyramid / integral work of the new
 * batch run under the cascade kernel of the old one and the host side of the old batch's last pass (copy-back, grouping)
 * under the device side of the new one -- the two parts of a synchronous call that nothing overlaps. At most one batch
 * is left unfetched inside the detector: a later submit (or any other detection call) fetches it first; results stay in
 * the ticket until collect. DEVICE frames must stay valid until the batch is collected; HOST frames are copied into the
 * detector's pinned staging area inside submit (three slots; the copy to the device is asynchronous on the front stream and a
 * pass's frames are staged one pass ahead of its kernels) and may be reused as soon as submit returns -- unless they already
 * live in pinned memory (hipHostMalloc / hipHostRegister): those are copied straight from the caller's buffer and must stay valid
 * until the batch is collected. collect frees
 * the ticket, except when it returns CC_ERR_BUFFER_TOO_SMALL (offsets[n_frames] then holds the count: collect again with
 * room for it). cc_detect_batch_discard ends a ticket whose results are not wanted (it waits for the batch's last pass;
 * NULL is accepted); a ticket that is neither collected nor discarded leaks. A ticket is ended only by the detector that
 * issued it: collect / discard with another detector return CC_ERR_INVALID_ARG and leave the ticket valid (collect it
 * with its own detector); after its detector has been destroyed a ticket can only be discarded (with any detector
 * argument, NULL included), collecting it is CC_ERR_INVALID_ARG. No reference counterpart (the reference handles one
 * image per call, tools/detection/Cpp/main.cpp:42-45). */
typedef struct cc_batch_ticket cc_batch_ticket;
CC_API cc_status cc_detect_batch_submit(cc_detector* d, const uint8_t* frames, int on_device, int n_frames, int width,
                                        int height, size_t row_stride, size_t frame_stride, const cc_detect_params* p,
                                        cc_batch_ticket** ticket);
CC_API cc_status cc_detect_batch_collect(cc_detector* d, cc_batch_ticket* ticket, cc_rect* out, int cap, int32_t* offsets);
CC_API cc_status cc_detect_batch_discard(cc_detector* d, cc_batch_ticket* ticket);

/* The outputRejectLevels overload of cv::CascadeClassifier::detectMultiScale (objects, rejectLevels, levelWeights,
 * ..., outputRejectLevels = true; OpenCV 4.6.0 objdetect, no call site in the reference: SURVEY.md 8f-4). Windows that
 * pass every stage are reported with level = number of stages and weight = the stage sum of the last stage; the
 * grouping keeps, per class, the highest level and among its members the largest weight (cv::groupRectangles with
 * weights). Parity unpinned like the rest of the detection side. out / reject_levels / level_weights hold cap entries. */
CC_API cc_status cc_detect_multiscale_levels(cc_detector* d, const uint8_t* gray, int width, int height, size_t row_stride,
                                             const cc_detect_params* p, cc_rect* out, int32_t* reject_levels,
                                             double* level_weights, int cap, int* n);

/* Ungrouped candidates of one frame, as int32[7] = {scale_idx, gx, gy, x, y, w, h}, sorted (scale, gy, gx). */
CC_API cc_status cc_detect_raw(cc_detector* d, const uint8_t* gray, int width, int height, size_t row_stride,
                               const cc_detect_params* p, int32_t* cand, int cap, int* n);

/* Parity instrumentation: per grid window result of runAt for one frame, all scales concatenated scale-major,
 * row-major [gy][gx]: codes: 1 pass, -k rejected at stage k (0 = stage 0), -1 window rejected before stage 0
 * (variance test); sums: stage accumulator (double) at exit; visited: 1 when OpenCV's scan loop visits the window
 * (stage-0 skip rule). Any of the three may be NULL. n_windows = capacity / total count. */
CC_API cc_status cc_detect_debug_windows(cc_detector* d, const uint8_t* gray, int width, int height, size_t row_stride,
                                         const cc_detect_params* p, int32_t* codes, double* sums, uint8_t* visited,
                                         int64_t cap, int64_t* n_windows);

/* Scale pyramid geometry for an image size (host computation only; no device needed). */
typedef struct cc_scale_info {
  float scale;
  int32_t width, height; /* resized image */
  int32_t ystep;
  int32_t nx, ny;        /* grid windows enumerated along x / y */
  int32_t win_w, win_h;  /* emitted rectangle size */
} cc_scale_info;
CC_API cc_status cc_scale_plan(int win_w, int win_h, int width, int height, const cc_detect_params* p,
                               cc_scale_info* out, int cap, int* n);

/* Run-time specialisation of the cascade kernel for THIS detector's cascade (Haar or LBP stump cascades): the first n_stages
 * stages (whole stages, capped by a code-size budget) are compiled with hiprtc into straight-line code whose LDS offsets,
 * weights, thresholds and leaf values are immediates; later stages stay table-driven. Same arithmetic, identical results;
 * the cascade kernel runs about 17 % faster on the bench cascade. Takes a few seconds the first time; code objects are
 * cached per process and on disk ($CCAMD_CACHE_DIR, default ~/.cache/cascadeclassifier_amd; set it empty to disable),
 * keyed by architecture, options and generated source. libhiprtc is loaded on demand and a missing library is CC_ERR_UNSUPPORTED, in which case the detector keeps
 * using the table-driven kernel. n_stages <= 0 switches back. cc_detector_specialized_stages reports the stages in
 * effect (0 = none). cc_cascade_compile_specialized only compiles (no device needed; arch e.g. "gfx950") and returns
 * the code-object size: the build check of the generated source. */
CC_API cc_status cc_detector_specialize(cc_detector* d, int n_stages);
/* Same, without waiting: generation and compilation run on a background host thread while detection continues on the
 * table-driven kernel; the first detection call after the build has finished loads the module and switches over
 * (cc_detector_specialized_stages tells when). CCAMD_AUTO_SPECIALIZE=<stages> in the environment does this for every
 * detector at creation, i.e. without any change to the calling code. */
CC_API cc_status cc_detector_specialize_async(cc_detector* d, int n_stages);
CC_API int cc_detector_specialized_stages(const cc_detector* d);
CC_API cc_status cc_cascade_compile_specialized(const cc_cascade* c, int n_stages, const char* arch, size_t* code_bytes);

/* Per-kernel device time accumulated since the last reset, measured with HIP events on the detector's streams.
 * Profiling is off by default (no events are recorded). A synchronous detection call reads its events before it
 * returns; a submitted batch (cc_detect_batch_submit) does not wait for them: cc_detector_get_timings first waits for the
 * detector's streams and reads what is outstanding. */
typedef struct cc_detector_timings {
  double resize_ms, integral_ms, eval_ms, finalize_ms;
  int64_t resize_launches, integral_launches, eval_launches, finalize_launches;
  int64_t frames;         /* frames processed */
  int64_t grid_windows;   /* grid windows evaluated */
  int64_t integral_elems; /* integral entries per channel produced */
  /* eval_ms spans ALL cascade-kernel launches of a pass (eval_launches counts passes). A run-time specialised Haar kernel is
   * one module per step (k_eval_spec_step2 over the tiles of STEP-2 scales, then k_eval_spec_step1): eval_step1_ms is the
   * part of eval_ms the STEP-1 module's launches took (0 when the pass is a single launch). */
  double eval_step1_ms;
} cc_detector_timings;
CC_API cc_status cc_detector_set_profiling(cc_detector* d, int enabled);
CC_API cc_status cc_detector_get_timings(cc_detector* d, cc_detector_timings* t, int reset);
/* Single-image calls (cc_detect_multiscale on a host image; the call shape of tools/detection/Cpp/main.cpp:45) replay
 * the whole device pass from one hipGraph once a first ordinary call has sized the buffers. Returns 1 if the LAST such
 * call was a graph launch, 0 if it used ordinary launches (first call, changed geometry, graphs switched off, or a
 * capture that did not come out: then the detector stays on ordinary launches), negative cc_status on a null handle. */
CC_API int cc_detector_graph_active(const cc_detector* d);

/* ============================================================================================
 * 3. Building blocks exposed for parity tests and roofline measurement (all run on the device).
 *    Replace: cv::integral (traincascade/lib/src/haarfeatures.cpp:109,112, lbpfeatures.cpp:27),
 *    cv::resize INTER_LINEAR_EXACT (traincascade/lib/src/imagestorage.cpp:86,117), cv::groupRectangles.
 * ============================================================================================ */
/* sum / sqsum / tilted: (height+1) x (width+1) int32, densely packed; sqsum is the detector's CV_32S variant
 * (wrap-around accumulation); any output may be NULL. */
CC_API cc_status cc_integral_u8(int device, const uint8_t* img, int width, int height, size_t row_stride, int32_t* sum,
                                int32_t* sqsum, int32_t* tilted);
CC_API cc_status cc_resize_linear_exact_u8(int device, const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst,
                                           int dw, int dh, size_t dstride);
/* Profiling aid: streams n_bytes of device memory once with the cascade kernel's load shape (one dword per lane,
 * 64 consecutive lanes) so that the FETCH_SIZE counter can be calibrated on a known byte count
 * (MI355X_MICROARCH.md, HBM section). *checksum receives the wrapped 32-bit sum of the words read. */
CC_API cc_status cc_debug_stream_dwords(int device, size_t n_bytes, int repeats, uint32_t* checksum);
/* Parity instrumentation: the bulk evaluator divides by a sample's norm factor with the operand-independent half of the
 * IEEE division hoisted out of the feature loop (cc_eval.hip: div_by_refined). Compares it with the division operator
 * on the device for n_pairs pseudo-random operand pairs of the evaluator's value ranges (x2: arbitrary floats and
 * integer-valued dividends over sqrt-shaped divisors); *mismatches must come back 0. */
CC_API cc_status cc_debug_division_check(int device, uint64_t n_pairs, uint64_t seed, uint64_t* mismatches);
/* Same kind of check for the cascade kernel's variance normalisation: counts values nf = area * valsqsum - valsum^2 (drawn
 * as the kernels form them) for which the refined-rsqrt path differs from (float)(1.0 / sqrt(nf)), the CPU's two correctly
 * rounded operations (cv::CascadeClassifier setWindow). Must report 0. */
CC_API cc_status cc_debug_vnf_check(int device, uint64_t n_values, uint64_t seed, uint64_t* mismatches);
/* Host-side (tiny, serial in the reference too): cv::groupRectangles(rects, group_threshold, eps). */
CC_API cc_status cc_group_rectangles(const cc_rect* rects, int n, int group_threshold, double eps, cc_rect* out, int cap,
                                     int* n_out);

/* ============================================================================================
 * 4. Training-side feature evaluator.
 *    Replaces: CvFeatureEvaluator / CvHaarEvaluator / CvLBPEvaluator
 *    (traincascade/lib/include/traincascade_features.h:155-188, haarfeatures.h:61-122, lbpfeatures.h:37-83):
 *    create+init  -> cc_eval_create           (features.cpp:72-81,91-97; haarfeatures.cpp:89-98; lbpfeatures.cpp:15-20)
 *    setImage     -> cc_eval_set_image(s)     (features.cpp:83-89; haarfeatures.cpp:100-114; lbpfeatures.cpp:22-28)
 *    operator()   -> cc_eval_calc / cc_eval_calc_batch (haarfeatures.h:108-122; lbpfeatures.h:44-45,70-83)
 *    getNumFeatures/getMaxCatCount/getFeatureSize/getCls -> cc_eval_* getters
 *    writeFeatures needs only the geometry -> cc_eval_feature_geometry (haarfeatures.cpp:311-320, lbpfeatures.cpp:65-68)
 *    bulk consumer CvCascadeBoostTrainData::precalculate (o_cvcascadeboosttraindata.cpp:490-596) -> cc_eval_calc_batch.
 * ============================================================================================ */
typedef struct cc_evaluator cc_evaluator;
enum { CC_HAAR_BASIC = 0, CC_HAAR_CORE = 1, CC_HAAR_ALL = 2 };

CC_API cc_status cc_eval_create(int feature_type, int haar_mode, int win_w, int win_h, int max_samples, int device,
                                cc_evaluator** out);
CC_API void cc_eval_destroy(cc_evaluator* e);
CC_API int cc_eval_num_features(const cc_evaluator* e);
CC_API int cc_eval_max_cat_count(const cc_evaluator* e); /* 0 Haar, 256 LBP */
CC_API int cc_eval_feature_size(const cc_evaluator* e);  /* 1 */
/* Haar: rects int32[3][4], weights float[3], *tilted; LBP: rects[0] = one cell (x y w h), weights/tilted untouched. */
CC_API cc_status cc_eval_feature_geometry(const cc_evaluator* e, int fi, int32_t* rects, float* weights, int* tilted);
/* img: win_h rows of win_w bytes, row_stride bytes apart. idx < max_samples.
 * The call does no device work (the trainer makes it per candidate window, cascadeclassifier.cpp:340-347): the pixels are
 * queued -- a later image for the same idx replaces the queued one; the queue reaches the device, runs of consecutive indices
 * per launch, before anything reads stored samples there -- and the integral(s) and norm factor of THIS window are mirrored
 * on the host, from which cc_eval_calc / cc_eval_calc_list answer for idx until the next cc_eval_set_image(s) (SURVEY.md 8b:
 * "scalar, host-mirror fast path"). The mirror's values are bit-identical to the device's for every catalog feature
 * (tests/test_gpu_eval.py::test_host_mirror_of_the_last_set_window_equals_the_device). */
CC_API cc_status cc_eval_set_image(cc_evaluator* e, const uint8_t* img, size_t row_stride, uint8_t cls_label, int idx);
/* n images of win_w x win_h, densely packed, stored at idx first_idx..first_idx+n-1; labels may be NULL (labels kept). */
CC_API cc_status cc_eval_set_images(cc_evaluator* e, const uint8_t* imgs, int n, int first_idx, const uint8_t* labels);
/* Host array of max_samples floats (the reference's `cls` Mat; o_cvcascadeboosttraindata.cpp:238-239 wraps it). */
CC_API const float* cc_eval_labels(const cc_evaluator* e);
/* Scalar operator()(featureIdx, sampleIdx). For the sample set last by cc_eval_set_image: answered from its host mirror, no
 * launch (~20 ns). Any other sample: one device evaluation (slow; prefer the batch / list forms). */
CC_API cc_status cc_eval_calc(cc_evaluator* e, int fi, int si, float* out);
/* out[k] = evaluator(feature_idx[k], si) for an arbitrary list of features and ONE stored sample, in one launch: the call
 * shape of the trainer's stage prediction on a freshly set window (CvCascadeBoostTree::predict ->
 * CvCascadeBoostTrainData::getVarValue, o_cvcascadeboosttree.cpp:16-39, o_cvcascadeboosttraindata.cpp:484-488), which asks
 * for the cascade's features one by one. For the sample set last by cc_eval_set_image the values come from its host mirror;
 * the C++ adaptor batches scalar calls for other samples through this entry point. */
CC_API cc_status cc_eval_calc_list(cc_evaluator* e, const int32_t* feature_idx, int n_feats, int si, float* out);
/* out[(fi - fi_begin) * n_samples + s] = evaluator(fi, sample_idx ? sample_idx[s] : s); out is HOST memory unless
 * out_on_device != 0 (then it is memory of the evaluator's device). */
CC_API cc_status cc_eval_calc_batch(cc_evaluator* e, int fi_begin, int fi_end, const int32_t* sample_idx, int n_samples,
                                    float* out, int out_on_device);
/* The same into DEVICE memory with a row pitch: d_out[(fi - fi_begin) * pitch + s], pitch in elements (0 = n_samples).
 * For consumers that stay on the device (presort, split search, a trainer holding valCache in HBM): the pitch lets rows
 * start on any alignment the consumer likes; measured in round 3, the store rate of the bulk evaluator does not depend
 * on it (profiles/r03_training_kernels.txt). */
CC_API cc_status cc_eval_calc_batch_device(cc_evaluator* e, int fi_begin, int fi_end, const int32_t* sample_idx, int n_samples,
                                           float* d_out, size_t pitch);
/* cc_eval_calc_batch plus, per feature, the argsort of the samples by value: the "sorted index" half of
 * CvCascadeBoostTrainData::precalculate (FeatureValAndIdxPrecalc / FeatureIdxOnlyPrecalc,
 * o_cvcascadeboosttraindata.cpp:490-556: `buf` rows of unsigned short when sample_count < 65536, else int).
 * idx[(fi - fi_begin) * n_samples + k] = index of the sample with the k-th smallest value of feature fi; equal values
 * keep increasing sample order (the reference's std::sort leaves their order unspecified). idx_bytes is 2 or 4
 * (2 requires n_samples <= 65536). vals (float[(fi_end - fi_begin) * n_samples], unsorted, as cc_eval_calc_batch) may be
 * NULL. Host outputs. Samples are 0..n_samples-1. */
CC_API cc_status cc_eval_calc_batch_sorted(cc_evaluator* e, int fi_begin, int fi_end, int n_samples, float* vals, void* idx,
                                           int idx_bytes);
/* Feature::calc (un-normalised) of caller-supplied Haar features on stored samples: the shape of the reference KATs
 * test_features.cpp:462-560. feats: n x {tilted, rects[3][4], weights[3]} as below. */
typedef struct cc_haar_feature {
  int32_t tilted;
  int32_t r[3][4];
  float w[3];
} cc_haar_feature;
CC_API cc_status cc_eval_calc_custom_haar(cc_evaluator* e, const cc_haar_feature* feats, int n_feats, int normalized,
                                          const int32_t* sample_idx, int n_samples, float* out);
/* Feature::calc (haarfeatures.h:114-122) on caller-held flattened integral images: sum / tilted are n_rows rows of
 * row_len int32 (one integral image per row, as the reference's `sum` / `tilted` Mats; either may be NULL if no feature
 * needs it), `step` is the integral's own row stride (window width + 1) used to form the corner offsets exactly as
 * CV_SUM_OFFSETS / CV_TILTED_OFFSETS do. out[f * n_rows + r]. Offsets outside [0, row_len) are rejected. */
CC_API cc_status cc_haar_feature_calc(int device, const cc_haar_feature* feats, int n_feats, int step, const int32_t* sum,
                                      const int32_t* tilted, int n_rows, int row_len, float* out);
/* Cached per-sample data copied back for parity tests: sum / tilted are (win_w+1)*(win_h+1) int32. */
CC_API cc_status cc_eval_get_sample(cc_evaluator* e, int idx, int32_t* sum, int32_t* tilted, float* normfactor);
/* Training-side cascade predict (CvCascadeClassifier::predict, cascadeclassifier.cpp:297-306 ->
 * boost.cpp:461-477 -> o_cvcascadeboosttree.cpp:16-39) of a stump cascade over stored samples:
 * out[s] = 1 if every stage passes else 0. Feature indices of `c` index the cascade's own <features> list. */
CC_API cc_status cc_eval_predict_cascade(cc_evaluator* e, const cc_cascade* c, const int32_t* sample_idx, int n_samples,
                                         uint8_t* out);
CC_API cc_status cc_eval_last_kernel_ms(cc_evaluator* e, double* ms);

/* ============================================================================================
 * 5. Batched negative mining over a background image.
 *    Replaces the per-window loop of CvCascadeClassifier::fillPassedSamples for negatives
 *    (traincascade/lib/src/cascadeclassifier.cpp:329-357): NegReader::get (imagestorage.cpp:90-126, one window of
 *    the image's sqrt(2) scale ladder at half-window steps) -> featureEvaluator->setImage (haarfeatures.cpp:100-114)
 *    -> CvCascadeClassifier::predict (cascadeclassifier.cpp:297-306 -> boost.cpp:461-477 ->
 *    o_cvcascadeboosttree.cpp:16-39). One call enumerates the reader's whole window stream for ONE image
 *    (NegReader::nextImg, imagestorage.cpp:57-88, has already chosen the image and the offset), builds each
 *    ladder level and its integral images once, and evaluates the trained stages on every window on the device.
 *    Training-side arithmetic: norm factor sqrt(area*sqsum - sum^2) as float, value = calc / nf (0 if nf == 0),
 *    ordered splits go left on `<=`, a stage passes iff sum >= threshold - 1e-5f.
 * ============================================================================================ */
typedef struct cc_negminer cc_negminer;
CC_API cc_status cc_negminer_create(const cc_cascade* trained_stages, int device, cc_negminer** out);
CC_API void cc_negminer_destroy(cc_negminer* m);
/* Ladder of one image: level l has size lw[l] x lh[l] and nx[l] x ny[l] window positions (x = ox + i*(win_w/2),
 * y = oy + j*(win_h/2)); the stream visits level 0 first, rows top to bottom, windows left to right. Host only. */
CC_API cc_status cc_negminer_plan(const cc_negminer* m, int width, int height, int ox, int oy, int32_t* lw, int32_t* lh,
                                  int32_t* nx, int32_t* ny, int cap, int* n_levels, int64_t* n_windows);
/* pass[i] = 1 iff stream window i passes every trained stage (i < *n_windows <= cap). If pixels != NULL the first
 * min(max_keep, #passing) passing windows are also copied out, win_w*win_h bytes each, in stream order, with their
 * stream indices in keep_index; *n_keep = how many were written. */
CC_API cc_status cc_negminer_run(cc_negminer* m, const uint8_t* gray, int width, int height, size_t row_stride, int ox, int oy,
                                 uint8_t* pass, int64_t cap, int64_t* n_windows, uint8_t* pixels, int64_t* keep_index,
                                 int max_keep, int* n_keep);
/* The same for n_images (<= 256) images of ONE size consumed with ONE offset -- what NegReader::nextImg hands out between two
 * wraps of its `round` counter over a background set of equal-sized images (imagestorage.cpp:57-88: the offset depends on
 * `round` and the image size only). One copy to the device, one launch of every kernel over all images, one copy back: a call
 * costs about what a single-image call costs. *n_windows = windows PER image; pass[k * *n_windows + i] for image k (cap >=
 * n_images * *n_windows); the kept windows are the first max_keep passing ones in stream order, image 0 first, with
 * keep_index[j] = k * *n_windows + i. A trainer that needs `need` more negatives walks pass[] in this order and stops where
 * the per-window loop would have stopped (cascadeclassifier.cpp:329-357): results behind that point are simply not consumed. */
CC_API cc_status cc_negminer_run_batch(cc_negminer* m, const uint8_t* const* images, int n_images, int width, int height,
                                       size_t row_stride, int ox, int oy, uint8_t* pass, int64_t cap, int64_t* n_windows,
                                       uint8_t* pixels, int64_t* keep_index, int max_keep, int* n_keep);

/* ============================================================================================
 * 6. Best-split search of a boosted-tree node on the device.
 *    Replaces CvDTree::find_best_split (traincascade/lib/src/o_cvdtree.cpp:313-357) and the per-variable searches it
 *    calls, CvBoostTree::find_split_ord_class / find_split_cat_class / find_split_ord_reg / find_split_cat_reg
 *    (o_cvboostree.cpp:151-247, 249-359, 361-426, 428-516), on the variable data CvCascadeBoostTrainData::
 *    get_ord_var_data / get_cat_var_data deliver (o_cvcascadeboosttraindata.cpp:403-482).
 *
 *    cc_eval_presort evaluates EVERY feature on stored samples [0, n_samples) once and keeps the result resident in
 *    HBM: for Haar the per-feature sorted order (values + sample indices; the reference's FeatureValAndIdxPrecalc,
 *    o_cvcascadeboosttraindata.cpp:535-556, limited there to the -precalcValBufSize / -precalcIdxBufSize budgets), for
 *    LBP the category codes (1 byte) and the samples of every feature ordered by (code, sample) (4 bytes). Haar: 6 bytes
 *    per (feature, sample): 19.5 GB for BASIC 24x24 x 20 000 samples; LBP: 5 bytes, 0.85 GB for 8 464 x 20 000. Call it after
 *    the stage's samples are in place (setImage / cc_eval_set_images) and again whenever they change.
 *
 *    cc_eval_find_best_split then searches one node: sample_idx are the node's stored-sample slots in node order (NULL
 *    = 0..n-1; no duplicates; any order is exact, LBP nodes listed in increasing order take the 7x faster kernel), weights the n + 2 "subtree weights" of CvBoostTree::calc_node_value
 *    (o_cvboostree.cpp:657-732: w[i], then the totals w[n], w[n+1]), responses the ordered responses (regression
 *    trees: LOGIT / GENTLE boost) or class_labels the 0/1 labels (DISCRETE / REAL boost), node_value = node->value.
 *    boost_type / split_criteria take CvBoost's values (boost.h: DISCRETE 0, REAL 1, LOGIT 2, GENTLE 3; DEFAULT 0,
 *    GINI 1, MISCLASS 3, SQERR 4) with the reference's defaulting rule (o_cvboostree.cpp:188-190).
 *    The arithmetic is the reference's, operation by operation, in double: one thread walks one variable's samples in
 *    sorted order. Equal feature values are taken in increasing sample-index order (the reference's std::sort leaves
 *    their order unspecified); categories are ordered with the same std::sort call as the reference. The winner over
 *    variables is chosen in variable order with the reference's float comparisons (o_cvdtree.cpp:340-341,351).
 *    out->found = 0 when no split has quality > 0 (find_best_split returns NULL). Optional per-variable results (NULL to
 *    skip): per_var_quality[vi] = best quality of variable vi on its own (-1 if it has no split),
 *    per_var_point[vi] = split_point (ordered) or number of categories sent left - 1 (categorical).
 * ============================================================================================ */
typedef struct cc_split {
  int32_t found;       /* 1 if a split was found */
  int32_t var_idx;     /* feature index (catalog order) */
  float quality;
  float ord_c;         /* ordered variables: threshold, samples with value <= ord_c go left */
  int32_t split_point; /* ordered variables: position of the last left sample in the node's sorted order */
  int32_t subset[8];   /* categorical variables: bit c set = category c goes left */
} cc_split;
/*    Multi-GPU: variables shard across processes: cc_eval_presort_range keeps only variables
 *    [fi_begin, fi_end) on this device; cc_eval_find_best_split then searches that range (var_idx stays a catalog index,
 *    per_var_* arrays hold fi_end - fi_begin entries). Because the reference's winner is the first variable with the
 *    largest float quality, the global winner is the shard result with the largest quality, lowest shard first.
 */
CC_API cc_status cc_eval_presort(cc_evaluator* e, int n_samples);
CC_API cc_status cc_eval_presort_range(cc_evaluator* e, int fi_begin, int fi_end, int n_samples);
CC_API cc_status cc_eval_find_best_split(cc_evaluator* e, const int32_t* sample_idx, int n, const double* weights,
                                         const float* responses, const int32_t* class_labels, double node_value,
                                         int boost_type, int split_criteria, cc_split* out, double* per_var_quality,
                                         int32_t* per_var_point);

/* ============================================================================================
 * 7. Multi-GPU: the gather of detections (BASELINE configs[3]).
 *    Detection shards by frame, one process per GPU, and needs no data-path collective: rank r runs cc_detect_batch on
 *    frames [lo, hi) = cc_shard_range(n_frames, r, world). The only exchange is this gather of the per-frame rectangle
 *    lists (a few KB per rank) over RCCL: one ncclAllGather of {frames, rectangles} headers and one padded
 *    ncclAllGather of the payload (per-frame counts, then rectangles), so every rank ends up with all frames'
 *    rectangles in global frame order. librccl is loaded on first use (an RCCL the process has already loaded, e.g.
 *    PyTorch's, is reused); world == 1 needs no RCCL at all.
 *    Bootstrap like any NCCL program: rank 0 calls cc_comm_unique_id and hands the CC_COMM_ID_BYTES bytes to the other
 *    ranks by whatever channel the host program has (MPI_Bcast, a file, a socket, torch.distributed); every rank then
 *    calls cc_comm_create (collective). There is no reference counterpart: the reference is single-process
 *    (tools/detection/Cpp/main.cpp:42-45 handles one image).
 * ============================================================================================ */
typedef struct cc_comm cc_comm;
#define CC_COMM_ID_BYTES 128
CC_API void cc_shard_range(int n_items, int rank, int world, int* lo, int* hi);
/* (An id starts RCCL's bootstrap listener: create one only for a communicator that all ranks then really build.) */
CC_API cc_status cc_comm_unique_id(void* id /* CC_COMM_ID_BYTES bytes */);
CC_API cc_status cc_comm_create(int device, int rank, int world, const void* id /* may be NULL when world == 1 */, cc_comm** out);
CC_API void cc_comm_destroy(cc_comm* c);
CC_API int cc_comm_rank(const cc_comm* c);
CC_API int cc_comm_world(const cc_comm* c);
/* Collective over the communicator. rects / offsets: this rank's frames as cc_detect_batch returns them (offsets has
 * n_frames + 1 entries). On return *n_frames_all / *n_rects_all hold the totals over all ranks. If the caller's buffers
 * (out: cap_rects rectangles, offsets_out: cap_frames + 1 entries) are too small the call returns
 * CC_ERR_BUFFER_TOO_SMALL, but only after the collectives have completed (the ranks stay in step) and with the result
 * kept in the communicator: cc_gather_fetch copies it out later without communicating (never call the gather again on
 * one rank alone). */
CC_API cc_status cc_gather_detections(cc_comm* c, const cc_rect* rects, const int32_t* offsets, int n_frames, cc_rect* out,
                                      int cap_rects, int32_t* offsets_out, int cap_frames, int* n_frames_all, int* n_rects_all);
CC_API cc_status cc_gather_fetch(const cc_comm* c, cc_rect* out, int cap_rects, int32_t* offsets_out, int cap_frames);

#ifdef __cplusplus
}
#endif
#endif /* CASCADECLASSIFIER_AMD_H_ */
