/* libpadel_hip.so — C-ABI of the MI355X (gfx950) inference engine behind the reference's trackers/.
 *
 * The reference is pure Python and has no FFI; the boundary this library replaces is the call the
 * tracker plugins make into their numeric engines:
 *
 *   - `self.model.predict(sample, conf, iou, imgsz, device, classes)` on an `ultralytics.YOLO`
 *       detect model  — trackers/players_tracker/players_tracker.py:351-359   -> pa_yolo_infer()
 *       pose model    — trackers/players_keypoints_tracker/players_keypoints_tracker.py:285-292
 *                       (preceded by the PIL resize of :260-266)               -> pa_yolo_infer()
 *   - `YOLO(model_path)` / `self.model.to(device)` — players_tracker.py:303,338-339 -> pa_model_create()
 *   - `tracknet(x)` on the 27-channel window stack — trackers/ball_tracker/ball_tracker.py:445-446
 *                                                                              -> pa_tracknet_infer()
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer and the
 * library never keeps a caller pointer past return; every call is synchronous at this boundary (the one
 * exception to both: the pa_yolo_submit / pa_yolo_wait pair below); one
 * engine per GPU, not thread-safe; return 0 on success, non-zero on failure with the message in
 * pa_last_error().  The Python binding (`padel_analytics_amd/engine.py`, ctypes) is the only caller.
 */
#ifndef PADEL_HIP_H
#define PADEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pa_engine pa_engine;
typedef struct pa_model pa_model;

#define PA_ABI_VERSION 5

/* ---- graph description (built on the host from a state_dict; see padel_analytics_amd/graph.py) ---- */

enum pa_op_kind {
    PA_OP_STEM = 1,       /* model.0: Conv 3x3 s2 + SiLU straight from the u8 network input      */
    PA_OP_CONV = 2,       /* Conv kxk (k in 1,3; stride 1,2) + bias + act (+ residual)           */
    PA_OP_SPPF_POOL = 3,  /* three chained MaxPool2d(5,1,2): slice c -> slices c+1..c+3          */
    PA_OP_UPSAMPLE2X = 4, /* nearest x2 of a slice into a slice of a buffer one level finer      */
    PA_OP_MAXPOOL2 = 5    /* MaxPool2d(2,2) into a buffer one level coarser                      */
};

enum pa_act { PA_ACT_NONE = 0, PA_ACT_SILU = 1, PA_ACT_RELU = 2, PA_ACT_SIGMOID = 3, PA_ACT_LEAKY = 4 /* nn.LeakyReLU(0.01): InpaintNet, reference trackers/ball_tracker/models.py:83-93 */ };

/* activation buffer: fp32 NHWC, spatial size = network input >> level, `channels` floats per pixel */
typedef struct pa_buf_desc {
    int32_t level;
    int32_t channels;
} pa_buf_desc;

typedef struct pa_op_desc {
    int32_t kind;
    int32_t in_buf, in_choff, cin;      /* slice read  (cin multiple of 16 for PA_OP_CONV)          */
    int32_t out_buf, out_choff, cout;   /* slice written (cout = real channels)                     */
    int32_t ksize, stride, act;
    int32_t res_buf, res_choff;         /* residual slice added after the activation; res_buf < 0: none */
    int32_t npad;                       /* PA_OP_CONV: rows of the packed weight matrix (multiple of 16) */
    int32_t reserved;                   /* PA_OP_CONV, fp32 models: offset (in floats, > 0) of the same weights pre-split
                                           into three bf16 planes for the bf16x3 kernels, [npad][k-step][hi|mid|lo][32];
                                           0: not provided.  PA_DTYPE_H2 models: offset (> 0) of the conv's npad per-output-
                                           channel inverse weight-row scales (fp32)                                  */
    int64_t w_off, b_off;               /* offsets (in floats) of packed weights / bias in the blob */
    int32_t flags;                      /* PA_OP_CONV of PA_DTYPE_H2 models: PA_CONV_W_SINGLE = the correction (m) plane of the packed
                                           weights is all zero — the weights are fp16 numbers (what an Ultralytics checkpoint stores)
                                           times the row's power of two, BatchNorm's scale travels in the per-channel output scale
                                           instead — so the kernels skip the wm x ah product: TWO MFMAs per operand pair, not three.
                                           Results do not depend on the flag (the skipped product is exactly zero).  ABI v4 */
    int32_t pad_;
} pa_op_desc;
#define PA_CONV_W_SINGLE 1

enum pa_task { PA_TASK_DETECT = 0, PA_TASK_POSE = 1, PA_TASK_TRACKNET = 2 };

/* PA_DTYPE_F32: the parity path (fp32 storage, fp32 MFMA: what the reference computes with half=False).
 * PA_DTYPE_F16 (detect / pose only; BASELINE configs[4]): activations and conv weights are fp16, accumulation fp32
 * (v_mfma_f32_16x16x32_f16), biases / stem weights / the Detect-Pose head maps (head_buf) stay fp32; conv weights sit
 * in the blob as fp16 [npad][Ktot] with K order (64-channel chunk, tap, 32-channel half), cin % 32 == 0,
 * w_off still counts 4-byte blob words.
 * PA_DTYPE_H2 (the fast fp32-equivalent path): every activation is an fp16 PAIR (x ~ h + m / 2048, 22-23 significant
 * bits) stored in 16-channel groups of 64 bytes [h x 16 | m x 16] — 4 bytes per channel like fp32 — written once by its
 * producer; conv weights sit at w_off as pre-split planes [npad][k-step][h | m][32 fp16] of the row-scaled weights, the
 * inverse row scales at `reserved`; a product is three f16 MFMAs instead of the six bf16 ones of the PA_DTYPE_F32 default
 * (csrc/h2_common.h).  Biases, stem weights, head maps and the TrackNet heat map stay fp32.  Values beyond the fp16
 * range (|x| > 65504) raise the model's overflow flag (pa_model_take_overflow): the caller repeats the call on a
 * PA_DTYPE_F32 model.                                                                                         */
enum pa_dtype { PA_DTYPE_F32 = 0, PA_DTYPE_F16 = 1, PA_DTYPE_H2 = 2 };

typedef struct pa_model_desc {
    int32_t task;
    int32_t nc;                 /* classes (detect / pose)                                          */
    int32_t nk, kpt_dim;        /* pose: nk = K * kpt_dim                                           */
    int32_t n_bufs;
    const pa_buf_desc* bufs;
    int32_t n_ops;
    const pa_op_desc* ops;
    int32_t head_buf[3];        /* detect/pose: per-level head maps, channels >= 64 + nc + nk (pixel stride);
                                   tracknet: head_buf[0] = output heat-map buffer                   */
    int32_t in_channels;        /* tracknet: channels of the fp32 input buffer (buffer 0)           */
    int32_t dtype;              /* enum pa_dtype: storage type of activations and conv weights      */
} pa_model_desc;

/* ---- engine ---- */
int pa_abi_version(void);
int pa_device_count(void);
int pa_engine_create(int device_id, pa_engine** out);
void pa_engine_destroy(pa_engine* eng);
/* message of the last failure on this engine (eng may be NULL for creation failures) */
const char* pa_last_error(pa_engine* eng);
int pa_engine_synchronize(pa_engine* eng);

/* raw device memory helpers (bench keeps frames resident in HBM with these) */
int pa_device_malloc(pa_engine* eng, size_t nbytes, void** out_dev);
int pa_device_free(pa_engine* eng, void* dev);
int pa_memcpy_h2d(pa_engine* eng, void* dst_dev, const void* src_host, size_t nbytes);
int pa_memcpy_d2h(pa_engine* eng, void* dst_host, const void* src_dev, size_t nbytes);
/* host -> HBM on the engine's copy stream: does not queue behind inference launched from another host
 * thread, so the next batch of frames uploads while the current one computes (the reference decodes and
 * uploads per tracker, trackers/runner.py:215-220; the runner's fan-out mode uploads once per batch)   */
int pa_upload(pa_engine* eng, void* dst_dev, const void* src_host, size_t nbytes);

/* page-lock a caller-owned host range (hipHostRegister) so that uploads from it run at PCIe speed and overlap compute:
 * the frame source of the end-to-end-from-host path (the reference decodes into host memory, trackers/runner.py:215-220).
 * The caller keeps ownership and unregisters before freeing.                                                  */
int pa_host_register(pa_engine* eng, void* ptr, size_t nbytes);
int pa_host_unregister(pa_engine* eng, void* ptr);

/* tuning knobs (tests / tools only).  Defaults come from the environment ONCE at pa_engine_create
 * (PADEL_CONV_IMPL=tap|lds, PADEL_CONV_VARIANT, PADEL_CONV_TUNE, PADEL_CONV_TAP_PD, PADEL_GRAPH, PADEL_ALIAS).
 * keys: "impl" (2 bf16x3 kernels = default, 0 fp32-MFMA tap kernels, 1 fp32-MFMA LDS cross-check kernel),
 * "variant" (forced tile id, -1 auto), "tune",
 * "tap_pd" (2|3), "graph" (hipGraph replay of the op list), "alias" (liveness-shared activation arena),
 * "timeline" (s_memtime-instrumented 3x3 kernel, dump to pa_engine_set_timeline_path)                 */
int pa_engine_set_tuning(pa_engine* eng, const char* key, int value);
int pa_engine_set_timeline_path(pa_engine* eng, const char* path);

/* ---- models ---- */
/* weights: packed blob produced by the host graph builder; copied to HBM, caller keeps ownership.
 * weights == NULL: the blob is allocated zero-filled and filled by pa_engine_bcast_weights (ranks that did
 * not load the checkpoint)                                                                             */
int pa_model_create(pa_engine* eng, const pa_model_desc* desc, const float* weights, size_t n_floats,
                    pa_model** out);
void pa_model_destroy(pa_model* m);
/* PA_DTYPE_H2 models: *out = 1 if any activation written since the last call did not fit the fp16 range (results of
 * those inferences are invalid: repeat them on a PA_DTYPE_F32 model), then clears the flag; always 0 for other dtypes */
int pa_model_take_overflow(pa_model* m, int* out);
/* frames / windows processed per replay of the graph (activation buffers are sized for it); default 64 */
int pa_model_set_max_batch(pa_model* m, int max_batch);
/* activation bytes of the current plan: the liveness-shared arena actually allocated, and the sum of the
 * logical buffers it replaces */
int pa_model_plan_bytes(pa_model* m, size_t* arena_bytes, size_t* logical_bytes);

/* tests: overwrite every byte of the planned activation arena (0xFF = NaN patterns in fp32 and fp16).  A following
 * inference must be unaffected: no kernel may read a byte nobody wrote since (h2 / fp32 graphs; fp16 graphs rely on
 * zero-initialised pad channels and are excluded)                                                            */
int pa_model_fill_arena(pa_model* m, int byte_value);

enum pa_pre_mode {
    PA_PRE_LETTERBOX = 0,   /* ultralytics LetterBox(auto, stride 32), cv2 INTER_LINEAR, pad 114    */
    PA_PRE_PIL_STRETCH = 1  /* PIL Image.resize((imgsz, imgsz)) bicubic, then LetterBox == identity */
};

typedef struct pa_yolo_params {
    int32_t imgsz;            /* 640 / 1280                                                         */
    int32_t pre_mode;         /* enum pa_pre_mode                                                   */
    int32_t channel_reverse;  /* 1: network channel c = frame channel 2-c      */
    int32_t letterbox_auto;   /* LetterBox auto flag (1 when all frames of the call share a shape)  */
    float conf, iou;
    int32_t max_det;          /* <= 300 rows are returned per image                                 */
    int32_t n_classes;        /* 0: keep every class                                                */
    const int32_t* classes;   /* host array                                                         */
    int32_t frames_on_device; /* 1: `frames` is an HBM pointer from pa_device_malloc                */
} pa_yolo_params;

/* frames: n x h x w x 3 uint8 (HWC).  out_boxes: n x max_det x 6 {x1,y1,x2,y2,conf,cls} in source
 * pixels; out_kpts: n x max_det x nk (NULL for detect); out_counts: n.                              */
int pa_yolo_infer(pa_model* m, const uint8_t* frames, int n, int h, int w, const pa_yolo_params* p,
                  float* out_boxes, float* out_kpts, int32_t* out_counts);

/* The same call in two halves, for pipelined batch loops (round 4): pa_yolo_submit enqueues everything pa_yolo_infer does
 * — preprocessing, network, decode, NMS, the result copies — behind whatever the engine's stream still holds and returns a
 * ticket WITHOUT waiting; pa_yolo_wait blocks until that ticket's results are in the caller's arrays.  Submitting batch
 * k + 1 before waiting for batch k keeps the GPU busy while the host collects results and prepares the next call (the
 * reference's `predict` is synchronous per batch, players_tracker.py:351-359: its GPU idles there).
 * Requirements: frames in HBM (frames_on_device = 1), n <= max_batch, no profiling; out_* must be page-locked
 * (pa_host_register) and must not be touched between submit and wait — the ONE place where the library keeps caller
 * pointers past return.  At most PA_MAX_INFLIGHT tickets per model; tickets complete in submission order; the plan
 * (source size, imgsz, batch) must not change while tickets are in flight.  `overflow` (may be NULL): 1 if an activation of
 * an h2 model has left the fp16 range by the time this ticket finished (sticky until pa_model_take_overflow): the
 * results of this and of later tickets are then not to be used (see PA_DTYPE_H2).                                   */
#define PA_MAX_INFLIGHT 4
int pa_yolo_submit(pa_model* m, const uint8_t* frames, int n, int h, int w, const pa_yolo_params* p,
                   float* out_boxes, float* out_kpts, int32_t* out_counts, int* ticket);
int pa_yolo_wait(pa_model* m, int ticket, int* overflow);

/* decode + NMS + rescale alone, on CALLER-SUPPLIED head maps (tests: hand-derived known answers for the Detect / Pose
 * inference branch, ops.non_max_suppression, scale_boxes — players_tracker.py:351-359 [upstream]): heads[l] =
 * n x H_l x W_l x c fp32 in the pa_yolo_head_shape layout of the plan for source size h x w (n <= max_batch); outputs as
 * pa_yolo_infer                                                                                               */
int pa_yolo_postprocess(pa_model* m, const float* const* heads, int n, int h, int w, const pa_yolo_params* p,
                        float* out_boxes, float* out_kpts, int32_t* out_counts);

/* raw head maps of the last pa_yolo_infer call (level 0..2): n x H_l x W_l x c fp32 (c from
 * pa_yolo_head_shape; channels [0,64) box DFL logits, [64,64+nc) class logits, then nk keypoint values) */
int pa_yolo_head_shape(pa_model* m, int level, int* h, int* w, int* c);
int pa_yolo_read_head(pa_model* m, int level, int n, float* out);

/* the u8 network input (NHWC4: R,G,B,0 as the network sees them) the last pa_yolo_infer call built from its
 * first n frames — byte-exact check of the letterbox (cv2 INTER_LINEAR) / Pillow-bicubic preprocessing */
int pa_yolo_netin_shape(pa_model* m, int* h, int* w);
int pa_yolo_read_netin(pa_model* m, int n, uint8_t* out);

/* generic graph forward (TrackNet): x = n x H x W x C_in fp32 NHWC (C_in = channels of buffer 0) ->
 * contents of buffer head_buf[0]: n x (H>>level) x (W>>level) x channels fp32 */
int pa_tracknet_infer(pa_model* m, const float* x, int n, int h, int w, int x_on_device, float* out,
                      int out_on_device);

/* ---- ball path: streaming TrackNet session (trackers/ball_tracker/ball_tracker.py:373-523) ----
 * Frames are fed in stream order; every frame is Pillow-bicubic-resized to 512x288 on the device once
 * (iterable.py:188), windows of 8 consecutive frames + the background are run through the TrackNet
 * model, the 8 overlapping window outputs of each frame are ensembled (ball_tracker.py:449-509) and
 * thresholded at 0.5 (predict.py:184).  The caller gets one 288x512 uint8 mask (255/0) per frame, in
 * frame order, and turns it into coordinates (predict.py:7-39) on the host.                          */
typedef struct pa_ball pa_ball;
int pa_ball_create(pa_model* tracknet, int src_h, int src_w, pa_ball** out);
void pa_ball_destroy(pa_ball* b);
/* background: src_h x src_w x 3 uint8 RGB (np.median(...).astype(uint8), iterable.py:70-78); also resets
 * the stream state                                                                                     */
int pa_ball_set_background(pa_ball* b, const uint8_t* median_rgb);
/* background = per-pixel median of the first n BGR frames, computed on the device with np.median + uint8
 * truncation semantics (iterable.py:59-78); out_median_rgb (optional) receives the src_h x src_w x 3 RGB
 * median; also resets the stream state                                                                 */
int pa_ball_background_from_frames(pa_ball* b, const uint8_t* frames_bgr, int n, int frames_on_device,
                                   uint8_t* out_median_rgb);
/* frames: n x src_h x src_w x 3 uint8 BGR (n <= the model's max_batch).  flush != 0 after the last
 * frames of the clip emits the 7 tail frames.  Outputs, each optional (NULL) but at least one of
 * masks / rects: out_masks (n + 7) x 288 x 512 bytes; out_heat (n + 7) x 288 x 512 fp32 ensembled heat
 * maps; out_rects (n + 7) x 4 int32 {x, y, w, h} = predict_location (predict.py:7-39) of each mask
 * (largest bounding rectangle among the 8-connected components; all zero: empty mask; w = -1: too many
 * foreground pixels for the device list, use the mask).                                               */
int pa_ball_feed(pa_ball* b, const uint8_t* frames_bgr, int n, int frames_on_device, int flush,
                 uint8_t* out_masks, float* out_heat, int32_t* out_rects, int* out_count);

/* predict_location on caller-supplied masks (n x 288 x 512 uint8, n <= max_batch + 7) -> n x 4 {x, y, w, h} */
int pa_ball_locate(pa_ball* b, const uint8_t* masks, int n, int32_t* out_rects);

/* ---- multi-GPU: one process per GPU, frames shard by batch, the ONLY collective is the one-time broadcast
 * of the packed weight blob from the rank that loaded the checkpoint; RCCL over xGMI,
 * HBM to HBM, no host bounce.  The reference has no distributed code (single `.to(device)`,
 * trackers/tracker.py:172-174) — this replaces "load the .pt in every process".
 * Bootstrap: rank 0 calls pa_comm_unique_id (128 bytes), the host side ships the bytes to the other ranks
 * out of band (torch.distributed store / any channel), every rank calls pa_engine_comm_init.          */
int pa_comm_unique_id(void* out, size_t cap);
int pa_engine_comm_init(pa_engine* eng, const void* unique_id, size_t id_bytes, int nranks, int rank);
void pa_engine_comm_destroy(pa_engine* eng);
int pa_engine_bcast_weights(pa_engine* eng, pa_model* m, int root);
/* same, with a separate source on the root: the root sends src's blob, EVERY rank (the root too) receives into dst, a
 * model created from a NULL blob; other ranks pass src = NULL                                               */
int pa_engine_bcast_weights_from(pa_engine* eng, pa_model* src, pa_model* dst, int root);
/* in-place broadcast of nbytes of device memory (e.g. the ball tracker's background median) */
int pa_engine_bcast(pa_engine* eng, void* dev_ptr, size_t nbytes, int root);
/* max over ranks of one double (bench: step time of the slowest rank) */
int pa_engine_allreduce_max(pa_engine* eng, double* value);
/* ABI v5 (round 6) — the sharded runner's result gather on the communicator the library owns, instead of torch.distributed
 * (trackers/runner.py has no counterpart: the reference runs one process).  Every rank contributes `nbytes` bytes of HOST memory
 * (its packed partial results; lengths differ per rank, 0 allowed).  Two calls, two collectives, no padding to the longest rank:
 * pa_engine_gather_sizes — ncclAllGather of the lengths: sizes[r] = rank r's nbytes, on every rank;
 * pa_engine_gather       — one ncclSend per rank / nranks - 1 ncclRecv on the root inside a group: on `root`, `recv` (capacity
 *                          recv_cap >= sum of sizes) receives the buffers back to back in rank order; other ranks pass recv = NULL.
 * `sizes` of the second call are the first call's.  nranks == 1 (or no communicator): host copies.                          */
int pa_engine_gather_sizes(pa_engine* eng, size_t nbytes, uint64_t* sizes);
int pa_engine_gather(pa_engine* eng, const void* send, size_t nbytes, void* recv, size_t recv_cap, const uint64_t* sizes, int root);

/* ---- host-native ByteTrack: `self.byte_track.update_with_detections(detections)`,
 * trackers/players_tracker/players_tracker.py:367-369 (constructed at :311 with frame_rate = fps; supervision
 * defaults track_activation_threshold .25, lost_track_buffer 30, minimum_matching_threshold .8).  Stateful and
 * sequential in frame order; pure host code (no GPU needed).  One call consumes a batch of frames in order:
 * boxes n_frames x stride x 6 {x1,y1,x2,y2,conf,cls} (pa_yolo_infer layout, stride = max_det), counts n_frames,
 * keep (optional) n_frames x stride bytes — 0 drops the box before tracking (polygon-zone filter,
 * players_tracker.py:364-365); out_ids n_frames x stride: public track id of each box, -1 = dropped
 * (unmatched / unconfirmed / filtered).  Thresholds are doubles: the same values the Python twin compares with.     */
typedef struct pa_bytetrack pa_bytetrack;
int pa_bytetrack_create(double track_activation_threshold, int lost_track_buffer, double minimum_matching_threshold,
                        int frame_rate, pa_bytetrack** out);
void pa_bytetrack_destroy(pa_bytetrack* b);
void pa_bytetrack_reset(pa_bytetrack* b);
int pa_bytetrack_update_batch(pa_bytetrack* b, const float* boxes, const int32_t* counts, const uint8_t* keep,
                              int n_frames, int stride, int32_t* out_ids);

/* ---- profiling (bench.py roofline): per-op device times of the LAST inference, HIP events on the
 * engine's stream.  kinds/ms/flops are host arrays of capacity cap; returns the number of records. */
int pa_engine_set_profiling(pa_engine* eng, int enable);
int pa_model_last_profile(pa_model* m, int cap, int32_t* kinds, float* ms, double* flops, int32_t* ksizes);
/* same records as CSV text: kind,ksize,M,cout,cin,stride,mf,nf,ms,flops,res per line (mf, nf: the workgroup tile's pixels and
 * channels; res: 1 if the conv adds a residual input it has to read); returns bytes written */
int pa_model_profile_text(pa_model* m, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PADEL_HIP_H */
