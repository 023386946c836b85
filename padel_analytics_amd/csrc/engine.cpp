// libpadel_hip.so — engine: owns one HIP stream per GPU, the packed weights and every activation
// buffer in HBM, plans a graph for a (source size, imgsz, batch) and replays it per batch.
// C-ABI declared in include/padel_hip.h.
#include "../../include/padel_hip.h"
#include "kernels.h"
#include <rccl/rccl.h>      // types only: librccl is dlopen'ed on first use (see the RCCL section)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <string>
#include <vector>

using namespace padel;

static thread_local std::string g_err;

// Tuning knobs: read from the environment ONCE at pa_engine_create (PADEL_CONV_IMPL=tap|bx3, PADEL_CONV_VARIANT,
// PADEL_CONV_TUNE, PADEL_CONV_TAP_PD, PADEL_GRAPH, PADEL_ALIAS), changed afterwards only through
// pa_engine_set_tuning — the replay loop never touches getenv.
struct Tuning {
    int impl = 2;        // 2 (default): bf16x3 kernels — fp32 values split exactly into 3 bf16, 6 products on the bf16
                         // pipe, fp32 accumulate (admitted by the same parity criteria as the fp32-MFMA kernels);
                         // 0: tap-unrolled LDS-DMA fp32-MFMA kernels, 1: LDS kernel (their bitwise cross-check)
    int variant = -1;    // forced tile id (tests / tools), -1: per-layer heuristic
    int tune = 1;        // bit 0: s_setprio around MFMA clusters
    int tap_pd = 2;      // prefetch distance of the 1x1 tap kernel
    int graph = 0;       // 1: replay the op list of a (model, batch) from a captured hipGraph
    int timeline = 0;    // 1: 3x3 tap launches of the 64x96 tile run the s_memtime-instrumented instantiation
    int alias = 1;       // 1: activation buffers share one arena by liveness, 0: disjoint ranges
    int fuse_stem = 1;   // 1: h2 YOLOv8 graphs run model.0 (stem) + model.1 (3x3 stride 2) as ONE kernel (stem_l1_h2.hip; default since round 4, 0 = two kernels)
    int w_single = 1;    // 1 (default): h2 convs whose packed weights have an all-zero m plane (PA_CONV_W_SINGLE) skip the wm x ah product;
                         // 0 (tests): all three products everywhere — bitwise the same results
    int fuse_sppf = 1;   // 1: h2 graphs run the three chained 5x5 max-pools of SPPF as ONE kernel (sppf_h2_kernel: keys in LDS, separable passes); 0 = three launches.  Bitwise the same maps
    int fold_up = 1;     // 1: an nn.Upsample(2) whose only reader is a bf16x3 1x1 conv is never materialised (the conv
                         // fetches those channels at [y >> 1][x >> 1] of the coarse map), 0: run the upsample kernel
};

struct pa_comm;

struct pa_engine {
    int dev = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // uploads that must not queue behind compute (pa_upload)
    std::string err;
    bool profiling = false;
    float* zeros = nullptr;   // 256 B of zeros: source of padded conv taps
    Tuning t;
    int tuning_epoch = 0;     // bumped by pa_engine_set_tuning: captured graphs of an older epoch are discarded
    std::string timeline_path;
    pa_comm* comm = nullptr;  // RCCL communicator (pa_engine_comm_init), optional
};

static int env_int(const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; }

#define PA_FAIL(eng, ...)                                        \
    do {                                                         \
        char _b[512];                                            \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                   \
        if (eng) (eng)->err = _b; else g_err = _b;               \
        return 1;                                                \
    } while (0)

#define PA_HIP(eng, call)                                                                  \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            (void)hipGetLastError();   /* reported here: must not surface again at the next launch's hipGetLastError() */ \
            PA_FAIL(eng, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                  \
    } while (0)

struct ProfRec { int kind; int ksize; double flops; hipEvent_t e0, e1; float ms; int M, cout, cin, stride, mf, nf, res; };

struct pa_model {
    pa_engine* e = nullptr;
    pa_model_desc d{};
    std::vector<pa_buf_desc> bufs;
    std::vector<pa_op_desc> ops;
    std::vector<int> fold_src;             // conv op i -> index of the upsample op it can absorb (-1: none), find_upsample_folds
    std::vector<int> fold_dst;             // upsample op j -> its absorbing conv (-1: none)
    float* d_w = nullptr;
    size_t n_w = 0;
    // h2 models: the h planes of the two-product stride-1 3x3 convs once more in MFMA operand order (conv_patch_h2r.hip), built on
    // the device from d_w before the first replay and again after a weight broadcast (ensure_operand_copies)
    char* d_wr = nullptr;
    std::vector<long long> wr_off;         // op -> byte offset into d_wr, -1: no copy
    bool wr_valid = false;
    unsigned* d_ovf = nullptr;             // h2 models: sticky "a value did not fit fp16" flag (pa_model_take_overflow)
    float* d_stage = nullptr; size_t stage_cap = 0;   // h2 generic graphs: fp32 input staged here before it is encoded
    int max_batch = 64;

    // plan
    bool planned = false;
    int p_h0 = 0, p_w0 = 0, p_imgsz = 0, p_pre = 0, p_auto = 0, p_batch = 0;
    int net_h = 0, net_w = 0;
    int rw = 0, rh = 0, top = 0, left = 0, lb_mode = 0;
    std::vector<float*> bptr;
    void* arena = nullptr;                 // what bptr points into
    size_t arena_bytes = 0, logical_bytes = 0;   // bytes of the plan with / without liveness aliasing
    unsigned h_ovf = 0;                    // overflow flag as read back by the last pa_yolo_infer calls (h2 models)
    bool ovf_cached = false;               // h_ovf is current: no kernel of this model has run since it was read
    // pa_yolo_submit / pa_yolo_wait: tickets in flight.  slot = ticket % PA_MAX_INFLIGHT; h_pin[slot]: the overflow flag as the
    // stream read it back after that ticket's kernels (page-locked: a pageable destination would make the copy block the host)
    unsigned* h_pin = nullptr;
    hipEvent_t tk_ev[PA_MAX_INFLIGHT]{};
    bool tk_busy[PA_MAX_INFLIGHT]{};
    int next_ticket = 0, n_inflight = 0;
    std::vector<int32_t> classes_host;     // what d_classes holds (uploaded again only when the caller's list changes)
    std::map<int, hipGraphExec_t> graphs;  // op-list replay per batch size (tuning "graph")
    int graph_epoch = -1;                  // engine tuning epoch the graphs were captured under
    uint8_t* d_frames = nullptr; size_t frames_cap = 0;
    uint8_t* d_netin = nullptr;
    uint8_t* d_tmp = nullptr;
    int32_t *d_xtab = nullptr, *d_ytab = nullptr;                 // cv2 bilinear tables
    int32_t *d_hb = nullptr, *d_hk = nullptr, *d_vb = nullptr, *d_vk = nullptr;   // PIL tables
    int hks = 0, vks = 0;
    // post
    int A = 0, P2 = 0;
    HeadLevel lv[3]{};
    float* d_cand = nullptr; int32_t* d_cidx = nullptr; int32_t* d_ccnt = nullptr;
    uint64_t* d_keys = nullptr; int32_t* d_order = nullptr; uint8_t* d_supp = nullptr;
    float* d_oboxes = nullptr; float* d_okpts = nullptr; int32_t* d_ocnt = nullptr;
    int32_t* d_classes = nullptr; int classes_cap = 0;
    int last_n = 0;
    std::vector<ProfRec> prof;
    size_t n_prof = 0;
};

static void find_upsample_folds(pa_model* m);

extern "C" {

int pa_abi_version(void) { return PA_ABI_VERSION; }

int pa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* pa_last_error(pa_engine* eng) { return eng ? eng->err.c_str() : g_err.c_str(); }

int pa_engine_create(int device_id, pa_engine** out) {
    if (!out) PA_FAIL((pa_engine*)nullptr, "pa_engine_create: out is NULL");
    int n = 0;
    PA_HIP((pa_engine*)nullptr, hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) PA_FAIL((pa_engine*)nullptr, "pa_engine_create: device %d of %d", device_id, n);
    PA_HIP((pa_engine*)nullptr, hipSetDevice(device_id));
    pa_engine* e = new pa_engine();
    e->dev = device_id;
    hipError_t r = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (r != hipSuccess) { delete e; PA_FAIL((pa_engine*)nullptr, "hipStreamCreate: %s", hipGetErrorString(r)); }
    if (r == hipSuccess) r = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking);
    if (r != hipSuccess) { delete e; PA_FAIL((pa_engine*)nullptr, "hipStreamCreate: %s", hipGetErrorString(r)); }
    r = hipMalloc((void**)&e->zeros, 256);
    if (r == hipSuccess) r = hipMemset(e->zeros, 0, 256);
    if (r != hipSuccess) { delete e; PA_FAIL((pa_engine*)nullptr, "zero page: %s", hipGetErrorString(r)); }
    r = padel::init_misc_kernels();
    if (r != hipSuccess) { delete e; PA_FAIL((pa_engine*)nullptr, "kernel attributes: %s", hipGetErrorString(r)); }
    if (const char* v = getenv("PADEL_CONV_IMPL")) {       // tap | bx3 ("lds", the kernel retired in round 5, is refused — it used to select tap silently)
        if (v[0] == 'l') { delete e; PA_FAIL((pa_engine*)nullptr, "PADEL_CONV_IMPL=%s: the LDS kernel was retired in round 5; tap or bx3", v); }
        e->t.impl = (v[0] == 'b') ? 2 : 0;
    }
    e->t.variant = env_int("PADEL_CONV_VARIANT", -1);
    e->t.tune = env_int("PADEL_CONV_TUNE", 1);
    e->t.tap_pd = env_int("PADEL_CONV_TAP_PD", 2) == 3 ? 3 : 2;
    e->t.graph = env_int("PADEL_GRAPH", 0);
    e->t.fuse_stem = env_int("PADEL_FUSE_STEM", 1);
    e->t.fuse_sppf = env_int("PADEL_FUSE_SPPF", 1);
    e->t.alias = env_int("PADEL_ALIAS", 1);
    e->t.fold_up = env_int("PADEL_FOLD_UP", 1);
    *out = e;
    return 0;
}

int pa_engine_set_tuning(pa_engine* e, const char* key, int value) {
    if (!e || !key) return 1;
    const std::string k = key;
    if (k == "impl") {                                      // 0: fp32-input MFMA tap kernels, 2: bf16x3
        if (value != 0 && value != 2) PA_FAIL(e, "tuning impl = %d: 0 (fp32-input MFMA tap kernels) or 2 (bf16x3); 1 was the LDS kernel, retired in round 5 (tools/legacy_conv)", value);
        e->t.impl = value;
    }
    else if (k == "variant") e->t.variant = value;
    else if (k == "tune") e->t.tune = value;
    else if (k == "tap_pd") e->t.tap_pd = (value == 3) ? 3 : 2;
    else if (k == "graph") e->t.graph = value ? 1 : 0;
    else if (k == "timeline") e->t.timeline = value ? 1 : 0;
    else if (k == "alias") e->t.alias = value ? 1 : 0;
    else if (k == "fold_up") e->t.fold_up = value ? 1 : 0;
    else if (!strcmp(key, "fuse_stem")) e->t.fuse_stem = value;
    else if (!strcmp(key, "fuse_sppf")) e->t.fuse_sppf = value;
    else if (!strcmp(key, "w_single")) e->t.w_single = value ? 1 : 0;
    else PA_FAIL(e, "pa_engine_set_tuning: unknown key '%s'", key);
    e->tuning_epoch++;
    return 0;
}

int pa_engine_set_timeline_path(pa_engine* e, const char* path) {
    if (!e) return 1;
    e->timeline_path = path ? path : "";
    return 0;
}

void pa_engine_destroy(pa_engine* e) {
    if (!e) return;
    hipSetDevice(e->dev);
    pa_engine_comm_destroy(e);
    if (e->zeros) hipFree(e->zeros);
    if (e->stream) hipStreamDestroy(e->stream);
    if (e->copy_stream) hipStreamDestroy(e->copy_stream);
    delete e;
}

int pa_engine_synchronize(pa_engine* e) {
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

int pa_engine_set_profiling(pa_engine* e, int enable) { e->profiling = enable != 0; return 0; }

int pa_device_malloc(pa_engine* e, size_t nbytes, void** out) {
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipMalloc(out, nbytes));
    return 0;
}
int pa_device_free(pa_engine* e, void* p) {
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipFree(p));
    return 0;
}
int pa_memcpy_h2d(pa_engine* e, void* dst, const void* src, size_t n) {
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, e->stream));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}
int pa_upload(pa_engine* e, void* dst, const void* src, size_t n) {
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, e->copy_stream));
    PA_HIP(e, hipStreamSynchronize(e->copy_stream));
    return 0;
}
int pa_host_register(pa_engine* e, void* ptr, size_t nbytes) {
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipHostRegister(ptr, nbytes, hipHostRegisterDefault));
    return 0;
}
int pa_host_unregister(pa_engine* e, void* ptr) {
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipHostUnregister(ptr));
    return 0;
}
int pa_memcpy_d2h(pa_engine* e, void* dst, const void* src, size_t n) {
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, e->stream));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

// ------------------------------------------------------------------------------- model
static int validate_desc(pa_engine* e, const pa_model_desc* d, size_t n_floats) {
    if (d->n_bufs <= 0 || d->n_ops <= 0 || !d->bufs || !d->ops) PA_FAIL(e, "model desc: empty graph");
    const bool f16 = d->dtype == PA_DTYPE_F16, h2 = d->dtype == PA_DTYPE_H2;
    if (d->dtype != PA_DTYPE_F32 && d->dtype != PA_DTYPE_F16 && d->dtype != PA_DTYPE_H2) PA_FAIL(e, "model desc: dtype %d", d->dtype);
    // (a TASK_TRACKNET graph in fp16 is a generic op list run through pa_tracknet_infer — conv unit tests; the ball
    // session itself is fp32 only, see pa_ball_create)
    const int kalign = f16 ? 31 : 15, valign = f16 ? 7 : 3;      // conv K granularity, 16-byte vector granularity (elements)
    for (int i = 0; i < d->n_bufs; ++i)
        if (d->bufs[i].level < 0 || d->bufs[i].level > 6 || d->bufs[i].channels <= 0 || (d->bufs[i].channels & 3))
            PA_FAIL(e, "model desc: buffer %d (level %d, channels %d)", i, d->bufs[i].level, d->bufs[i].channels);
    auto okslice = [&](int b, int off, int c) {
        return b >= 0 && b < d->n_bufs && off >= 0 && c > 0 && off + c <= d->bufs[b].channels;
    };
    auto is_head_buf = [&](int b) { return b == d->head_buf[0] || b == d->head_buf[1] || b == d->head_buf[2]; };
    if (h2)          // h2 buffers are made of whole 16-channel groups (the fp32 head maps excepted)
        for (int i = 0; i < d->n_bufs; ++i)
            if (!is_head_buf(i) && (d->bufs[i].channels & 15)) PA_FAIL(e, "model desc: h2 buffer %d has %d channels", i, d->bufs[i].channels);
    for (int i = 0; i < d->n_ops; ++i) {
        const pa_op_desc& o = d->ops[i];
        if (!okslice(o.out_buf, o.out_choff, o.cout)) PA_FAIL(e, "op %d: bad output slice", i);
        if (h2 && o.kind != PA_OP_STEM && (((o.in_choff | o.cin) & 15) || (!is_head_buf(o.out_buf) && (o.out_choff & 3))))
            PA_FAIL(e, "op %d: h2 slices must start on a 16-channel group", i);
        if (o.kind != PA_OP_STEM && !okslice(o.in_buf, o.in_choff, o.cin)) PA_FAIL(e, "op %d: bad input slice", i);
        if (o.kind == PA_OP_CONV) {
            if ((o.cin & kalign) || (o.in_choff & valign) || (o.ksize != 1 && o.ksize != 3) || (o.stride != 1 && o.stride != 2))
                PA_FAIL(e, "op %d: unsupported conv (cin %d choff %d k %d s %d)", i, o.cin, o.in_choff, o.ksize, o.stride);
            if (o.npad < o.cout || (o.npad & 15)) PA_FAIL(e, "op %d: npad %d for cout %d", i, o.npad, o.cout);
            const size_t ksteps3 = o.ksize == 3 ? (size_t)(o.cin / 32) * 9 + ((o.cin & 16) ? 5 : 0) : (size_t)(o.cin + 31) / 32;
            const size_t wn = h2 ? (size_t)o.npad * ksteps3 * 32 : (size_t)o.npad * o.cin * o.ksize * o.ksize / (f16 ? 2 : 1);
            if (o.w_off < 0 || (o.w_off & 3) || (size_t)o.w_off + wn > n_floats || o.b_off < 0 ||
                (size_t)o.b_off + o.npad > n_floats)
                PA_FAIL(e, "op %d: weights outside the blob", i);
            if (o.res_buf >= 0 && !okslice(o.res_buf, o.res_choff, o.cout)) PA_FAIL(e, "op %d: bad residual slice", i);
            if (h2) {
                if (o.reserved <= 0 || (o.reserved & 3) || (size_t)o.reserved + o.npad > n_floats) PA_FAIL(e, "op %d: h2 row scales outside the blob", i);
                if (o.res_buf >= 0 && (o.res_choff & 3)) PA_FAIL(e, "op %d: h2 residual slice alignment", i);
            } else if (o.reserved < 0 || (o.reserved & 3) || (o.reserved > 0 && (size_t)o.reserved + (size_t)o.npad * 48 * ksteps3 > n_floats))
                PA_FAIL(e, "op %d: bf16x3 weights outside the blob", i);
            const int lin = d->bufs[o.in_buf].level, lout = d->bufs[o.out_buf].level;
            if (lout != lin + (o.stride == 2 ? 1 : 0)) PA_FAIL(e, "op %d: level mismatch", i);
        } else if (o.kind == PA_OP_STEM) {
            if ((o.cout & 15) || (size_t)o.w_off + (size_t)o.cout * 27 > n_floats || (size_t)o.b_off + o.cout > n_floats)
                PA_FAIL(e, "op %d: bad stem", i);
            if (d->bufs[o.out_buf].level != 1) PA_FAIL(e, "op %d: stem output must be level 1", i);
        } else if (o.kind == PA_OP_SPPF_POOL) {
            if ((o.cin & valign) || (o.in_choff & valign) || o.in_buf != o.out_buf || !okslice(o.in_buf, o.in_choff, 4 * o.cin))
                PA_FAIL(e, "op %d: bad sppf slices", i);
        } else if (o.kind == PA_OP_UPSAMPLE2X) {
            if (d->bufs[o.out_buf].level != d->bufs[o.in_buf].level - 1 || o.cin != o.cout || ((o.cin | o.in_choff | o.out_choff) & valign))
                PA_FAIL(e, "op %d: bad upsample", i);
        } else if (o.kind == PA_OP_MAXPOOL2) {
            if (d->bufs[o.out_buf].level != d->bufs[o.in_buf].level + 1 || o.cin != o.cout || ((o.cin | o.in_choff | o.out_choff) & valign))
                PA_FAIL(e, "op %d: bad maxpool", i);
        } else {
            PA_FAIL(e, "op %d: unknown kind %d", i, o.kind);
        }
    }
    if (d->task == PA_TASK_DETECT || d->task == PA_TASK_POSE) {
        for (int l = 0; l < 3; ++l) {
            const int b = d->head_buf[l];
            if (b < 0 || b >= d->n_bufs || d->bufs[b].channels < 64 + d->nc + d->nk || d->bufs[b].level != 3 + l ||
                d->bufs[b].channels != d->bufs[d->head_buf[0]].channels)
                PA_FAIL(e, "model desc: head buffer %d", l);
        }
        if (d->nk && (d->kpt_dim < 2 || d->kpt_dim > 3 || d->nk % d->kpt_dim)) PA_FAIL(e, "model desc: kpt shape");
    }
    return 0;
}

int pa_model_create(pa_engine* e, const pa_model_desc* desc, const float* weights, size_t n_floats, pa_model** out) {
    // weights == NULL: the blob is allocated zero-filled and arrives through pa_engine_bcast_weights
    if (!e || !desc || !out) PA_FAIL(e, "pa_model_create: NULL argument");
    if (validate_desc(e, desc, n_floats)) return 1;
    PA_HIP(e, hipSetDevice(e->dev));
    pa_model* m = new pa_model();
    m->e = e;
    m->d = *desc;
    m->bufs.assign(desc->bufs, desc->bufs + desc->n_bufs);
    m->ops.assign(desc->ops, desc->ops + desc->n_ops);
    find_upsample_folds(m);
    m->d.bufs = m->bufs.data();
    m->d.ops = m->ops.data();
    m->n_w = n_floats;
    hipError_t r = hipMalloc((void**)&m->d_w, n_floats * sizeof(float) + kConvReadSlack);
    if (r == hipSuccess)
        r = weights ? hipMemcpyAsync(m->d_w, weights, n_floats * sizeof(float), hipMemcpyHostToDevice, e->stream)
                    : hipMemsetAsync(m->d_w, 0, n_floats * sizeof(float), e->stream);
    if (r == hipSuccess) r = hipMemsetAsync(m->d_w + n_floats, 0, kConvReadSlack, e->stream);
    if (r == hipSuccess) r = hipMalloc((void**)&m->d_ovf, 256);
    if (r == hipSuccess) r = hipMemsetAsync(m->d_ovf, 0, 256, e->stream);
    if (r == hipSuccess) r = hipHostMalloc((void**)&m->h_pin, 64 * sizeof(unsigned), hipHostMallocDefault);
    if (r == hipSuccess) memset(m->h_pin, 0, 64 * sizeof(unsigned));
    for (int k = 0; k < PA_MAX_INFLIGHT && r == hipSuccess; ++k) r = hipEventCreateWithFlags(&m->tk_ev[k], hipEventDisableTiming);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    if (r != hipSuccess) { delete m; PA_FAIL(e, "weights upload: %s", hipGetErrorString(r)); }
    *out = m;
    return 0;
}

static void free_plan(pa_model* m) {
    for (auto& g : m->graphs) hipGraphExecDestroy(g.second);
    m->graphs.clear();
    if (m->arena) hipFree(m->arena);
    m->arena = nullptr;
    m->bptr.clear();
    void* ptrs[] = {m->d_netin, m->d_tmp, m->d_xtab, m->d_ytab, m->d_hb, m->d_hk, m->d_vb, m->d_vk, m->d_cand,
                    m->d_cidx, m->d_ccnt, m->d_keys, m->d_order, m->d_supp, m->d_oboxes, m->d_okpts, m->d_ocnt};
    for (void* p : ptrs) if (p) hipFree(p);
    m->d_netin = m->d_tmp = nullptr;
    m->d_xtab = m->d_ytab = m->d_hb = m->d_hk = m->d_vb = m->d_vk = nullptr;
    m->d_cand = nullptr; m->d_cidx = m->d_ccnt = nullptr; m->d_keys = nullptr; m->d_order = nullptr; m->d_supp = nullptr;
    m->d_oboxes = m->d_okpts = nullptr; m->d_ocnt = nullptr;
    m->planned = false;
}

void pa_model_destroy(pa_model* m) {
    if (!m) return;
    hipSetDevice(m->e->dev);
    hipStreamSynchronize(m->e->stream);
    free_plan(m);
    for (auto& r : m->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    if (m->d_frames) hipFree(m->d_frames);
    if (m->d_classes) hipFree(m->d_classes);
    if (m->d_w) hipFree(m->d_w);
    if (m->d_wr) hipFree(m->d_wr);
    if (m->d_ovf) hipFree(m->d_ovf);
    if (m->d_stage) hipFree(m->d_stage);
    if (m->h_pin) hipHostFree(m->h_pin);
    for (int k = 0; k < PA_MAX_INFLIGHT; ++k) if (m->tk_ev[k]) hipEventDestroy(m->tk_ev[k]);
    delete m;
}

int pa_model_take_overflow(pa_model* m, int* out) {
    if (!m || !out) return 1;
    pa_engine* e = m->e;
    *out = 0;
    if (m->d.dtype != PA_DTYPE_H2) return 0;
    PA_HIP(e, hipSetDevice(e->dev));
    unsigned v = 0;
    if (m->ovf_cached) {                 // pa_yolo_infer read the flag back with its results (nothing has run on the model since)
        v = m->h_ovf;
    } else {
        PA_HIP(e, hipMemcpyAsync(&v, m->d_ovf, sizeof(v), hipMemcpyDeviceToHost, e->stream));
        PA_HIP(e, hipStreamSynchronize(e->stream));
    }
    m->ovf_cached = false;
    m->h_ovf = 0;
    if (v) {
        PA_HIP(e, hipMemsetAsync(m->d_ovf, 0, sizeof(v), e->stream));
        PA_HIP(e, hipStreamSynchronize(e->stream));
    }
    *out = v ? 1 : 0;
    return 0;
}

int pa_model_set_max_batch(pa_model* m, int max_batch) {
    if (max_batch < 1) PA_FAIL(m->e, "max_batch %d", max_batch);
    if (max_batch != m->max_batch) { hipSetDevice(m->e->dev); hipStreamSynchronize(m->e->stream); free_plan(m); m->max_batch = max_batch; }
    return 0;
}

// ---- host-side coefficient tables --------------------------------------------------------------
// cv2.resize INTER_LINEAR u8 (see oracle/yolov8_ref.py:cv2_resize_linear_u8)
static void cv2_linear_table(int src, int dst, std::vector<int32_t>& tab) {
    tab.resize((size_t)dst * 3);
    const double scale = (double)src / dst;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src - 1) { s = src - 1; f = 0.f; }
        tab[d * 3 + 0] = s;
        tab[d * 3 + 1] = (int)std::nearbyint((1.f - f) * 2048.f);
        tab[d * 3 + 2] = (int)std::nearbyint(f * 2048.f);
    }
}

// Pillow ImagingResample precompute_coeffs + normalize_coeffs_8bpc, bicubic a = -0.5
static double pil_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static int pil_coeffs(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& kk) {
    const double scale = (double)in_size / out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { k[x] = pil_bicubic((x + xmin - center + 0.5) * ss); ww += k[x]; }
        for (int x = 0; x < xmax; ++x) {
            double v = ww != 0.0 ? k[x] / ww : k[x];
            kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << 22)) : (int)(0.5 + v * (1 << 22));
        }
        bounds[xx * 2] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
    return ksize;
}

static hipError_t upload(pa_engine* e, int32_t** dptr, const std::vector<int32_t>& v) {
    hipError_t r = hipMalloc((void**)dptr, v.size() * sizeof(int32_t));
    if (r != hipSuccess) return r;
    r = hipMemcpyAsync(*dptr, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
    if (r != hipSuccess) return r;
    return hipStreamSynchronize(e->stream);
}

// Activation memory plan.  Every logical buffer of the graph needs batch * H * W * channels floats, but most
// are dead most of the time (a C2f's scratch dies with the C2f): buffers whose live ranges on the op list do not
// overlap share bytes of ONE arena (first-fit by offset over the buffers ordered by first use).  A buffer is
// live from the first op that touches it to the last one; the network input of a TrackNet graph (buffer 0) is
// live from before op 0, head buffers stay live past the last op (decode / NMS / pa_yolo_read_head read them).
// Aliased bytes always hold finite fp32 activations, so a zero-weighted pad channel still contributes exactly 0.
// nn.Upsample(scale_factor=2) + torch.cat is never materialised where the consumer allows it.  Upsample op j
// (coarse slice S[so, so + c) -> fine slice X[xo, xo + c)) is absorbed by conv i when: i is a stride-1 conv with bf16x3
// weights — 1x1 (YOLOv8's FPN joins) or 3x3 with cin % 32 == 0 (TrackNet's decoder blocks) — whose input slice starts
// at X[xo] and covers the c channels (c % 32 == 0), nothing else reads those channels of X, and nothing overwrites the
// source slice between j and i.  Whether the absorption is USED is decided per launch (fp32 model, impl bx3, tuning
// fold_up, and for a 3x3 consumer the patch kernel being the tile chosen: conv_launch_args); the liveness plan keeps S
// alive until i either way.
static void find_upsample_folds(pa_model* m) {
    const int nops = (int)m->ops.size();
    m->fold_src.assign(nops, -1);
    m->fold_dst.assign(nops, -1);
    auto overlap = [](int a0, int an, int b0, int bn) { return a0 < b0 + bn && b0 < a0 + an; };
    for (int j = 0; j < nops; ++j) {
        const pa_op_desc& u = m->ops[j];
        if (u.kind != PA_OP_UPSAMPLE2X || (u.cin & 31)) continue;
        bool head = false;
        for (int l = 0; l < 3; ++l) head |= m->d.head_buf[l] == u.out_buf;
        if (head) continue;
        int reader = -1, readers = 0;
        for (int k = 0; k < nops; ++k) {
            const pa_op_desc& o = m->ops[k];
            if (k == j) continue;
            bool reads = false;
            if (o.kind == PA_OP_CONV) {
                reads = (o.in_buf == u.out_buf && overlap(o.in_choff, o.cin, u.out_choff, u.cin)) ||
                        (o.res_buf == u.out_buf && overlap(o.res_choff, o.cout, u.out_choff, u.cin));
            } else if (o.kind == PA_OP_SPPF_POOL) {
                reads = o.in_buf == u.out_buf && overlap(o.in_choff, 4 * o.cin, u.out_choff, u.cin);
            } else if (o.kind != PA_OP_STEM) {
                reads = o.in_buf == u.out_buf && overlap(o.in_choff, o.cin, u.out_choff, u.cin);
            }
            if (reads) { reader = k; ++readers; }
        }
        if (readers != 1 || reader < j) continue;
        const pa_op_desc& c = m->ops[reader];
        const bool shape_ok = c.kind == PA_OP_CONV && c.stride == 1 && (c.ksize == 1 || (c.ksize == 3 && (c.cin & 31) == 0));
        if (!shape_ok || c.reserved <= 0 || c.in_buf != u.out_buf || c.in_choff != u.out_choff || c.cin < u.cin ||
            m->fold_src[reader] >= 0)
            continue;
        bool clobbered = false;
        for (int k = j + 1; k < reader && !clobbered; ++k) {
            const pa_op_desc& o = m->ops[k];
            const int wc = o.kind == PA_OP_SPPF_POOL ? 4 * o.cin : o.cout;
            clobbered = o.out_buf == u.in_buf && overlap(o.out_choff, wc, u.in_choff, u.cin);
        }
        if (clobbered) continue;
        m->fold_src[reader] = j;
        m->fold_dst[j] = reader;
    }
}
static bool fold_active(const pa_model* m, int conv_op) {
    if (m->d.dtype == PA_DTYPE_H2) return m->e->t.fold_up && m->fold_src[conv_op] >= 0;
    return m->e->t.fold_up && m->e->t.impl == 2 && m->d.dtype != PA_DTYPE_F16 && m->fold_src[conv_op] >= 0;
}

static int plan_buffers(pa_model* m, int batch) {
    pa_engine* e = m->e;
    int maxl = 0;
    for (const auto& b : m->bufs) maxl = std::max(maxl, b.level);
    const int mask = (1 << maxl) - 1;
    if ((m->net_h & mask) || (m->net_w & mask)) PA_FAIL(e, "network input %dx%d is not a multiple of %d", m->net_h, m->net_w, mask + 1);
    const int nb = (int)m->bufs.size(), nops = (int)m->ops.size();
    std::vector<int> first(nb, nops + 1), last(nb, -2);
    auto touch = [&](int b, int i) { if (b >= 0 && b < nb) { first[b] = std::min(first[b], i); last[b] = std::max(last[b], i); } };
    for (int i = 0; i < nops; ++i) {
        const pa_op_desc& o = m->ops[i];
        if (o.kind != PA_OP_STEM) touch(o.in_buf, i);
        touch(o.out_buf, i);
        if (o.kind == PA_OP_CONV && o.res_buf >= 0) touch(o.res_buf, i);
        if (o.kind == PA_OP_CONV && m->fold_src[i] >= 0) touch(m->ops[m->fold_src[i]].in_buf, i);   // an absorbed upsample's source
    }
    if (m->d.task == PA_TASK_TRACKNET) touch(0, -1);
    const bool f16 = m->d.dtype == PA_DTYPE_F16;
    auto is_head = [&](int b) { return b == m->d.head_buf[0] || b == m->d.head_buf[1] || b == m->d.head_buf[2]; };
    for (int l = 0; l < 3; ++l) touch(m->d.head_buf[l], nops + 1);
    // fp16 models: the head maps are the only fp32 buffers; they never share bytes with fp16 buffers, so a pad
    // channel that is read under a zero weight always holds a finite fp16 value, never reinterpreted fp32 bits
    if (f16) for (int l = 0; l < 3; ++l) touch(m->d.head_buf[l], -1);
    std::vector<size_t> bytes(nb), off(nb, 0);
    size_t logical = 0;
    for (int i = 0; i < nb; ++i) {
        const size_t H = m->net_h >> m->bufs[i].level, W = m->net_w >> m->bufs[i].level;
        const size_t es = (f16 && !is_head(i)) ? 2 : 4;
        bytes[i] = ((size_t)batch * H * W * m->bufs[i].channels * es + kConvReadSlack + 255) & ~(size_t)255;
        logical += bytes[i];
    }
    size_t total = 0;
    if (e->t.alias) {
        std::vector<int> order(nb);
        for (int i = 0; i < nb; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
        std::vector<int> act;                                     // placed buffers still live, sorted by offset
        for (int b : order) {
            if (last[b] < first[b]) continue;                     // never touched: offset 0, never accessed
            act.erase(std::remove_if(act.begin(), act.end(), [&](int a) { return last[a] < first[b]; }), act.end());
            size_t o = 0;
            for (int a : act) {
                if (o + bytes[b] <= off[a]) break;
                o = std::max(o, off[a] + bytes[a]);
            }
            off[b] = o;
            act.insert(std::upper_bound(act.begin(), act.end(), b, [&](int x, int y) { return off[x] < off[y]; }), b);
            total = std::max(total, o + bytes[b]);
        }
    } else {
        for (int i = 0; i < nb; ++i) { off[i] = total; total += bytes[i]; }
    }
    PA_HIP(e, hipMalloc(&m->arena, total + kConvReadSlack));
    PA_HIP(e, hipMemsetAsync(m->arena, 0, total + kConvReadSlack, e->stream));
    m->bptr.assign(nb, nullptr);
    for (int i = 0; i < nb; ++i) m->bptr[i] = reinterpret_cast<float*>(static_cast<char*>(m->arena) + off[i]);
    m->arena_bytes = total;
    m->logical_bytes = logical;
    m->p_batch = batch;
    return 0;
}

static int plan_yolo(pa_model* m, int h0, int w0, const pa_yolo_params* p) {
    pa_engine* e = m->e;
    PA_HIP(e, hipStreamSynchronize(e->stream));
    free_plan(m);
    const int S = p->imgsz;
    if (S <= 0 || (S & 31)) PA_FAIL(e, "imgsz %d must be a positive multiple of 32", S);
    if (p->pre_mode == PA_PRE_LETTERBOX) {
        const double r = std::min((double)S / h0, (double)S / w0);
        m->rw = (int)std::nearbyint(w0 * r);
        m->rh = (int)std::nearbyint(h0 * r);
        double dw = S - m->rw, dh = S - m->rh;
        if (p->letterbox_auto) { dw = std::fmod(dw, 32.0); dh = std::fmod(dh, 32.0); }
        dw /= 2; dh /= 2;
        m->top = (int)std::nearbyint(dh - 0.1);
        const int bottom = (int)std::nearbyint(dh + 0.1);
        m->left = (int)std::nearbyint(dw - 0.1);
        const int right = (int)std::nearbyint(dw + 0.1);
        m->net_h = m->rh + m->top + bottom;
        m->net_w = m->rw + m->left + right;
        if (w0 == m->rw && h0 == m->rh) m->lb_mode = 0;
        else if (w0 == 2 * m->rw && h0 == 2 * m->rh) m->lb_mode = 1;
        else {
            m->lb_mode = 2;
            std::vector<int32_t> xt, yt;
            cv2_linear_table(w0, m->rw, xt);
            cv2_linear_table(h0, m->rh, yt);
            PA_HIP(e, upload(e, &m->d_xtab, xt));
            PA_HIP(e, upload(e, &m->d_ytab, yt));
        }
    } else if (p->pre_mode == PA_PRE_PIL_STRETCH) {
        m->net_h = m->net_w = S;
        m->rw = m->rh = S; m->top = m->left = 0; m->lb_mode = 0;
        std::vector<int32_t> b, k;
        if (w0 != S) { m->hks = pil_coeffs(w0, S, b, k); PA_HIP(e, upload(e, &m->d_hb, b)); PA_HIP(e, upload(e, &m->d_hk, k)); }
        if (h0 != S) { m->vks = pil_coeffs(h0, S, b, k); PA_HIP(e, upload(e, &m->d_vb, b)); PA_HIP(e, upload(e, &m->d_vk, k)); }
        if (w0 != S && h0 != S) PA_HIP(e, hipMalloc((void**)&m->d_tmp, (size_t)m->max_batch * h0 * S * 3));
    } else {
        PA_FAIL(e, "unknown pre_mode %d", p->pre_mode);
    }
    const int B = m->max_batch;
    PA_HIP(e, hipMalloc((void**)&m->d_netin, (size_t)B * m->net_h * m->net_w * 4));
    if (plan_buffers(m, B)) return 1;
    int a0 = 0;
    for (int l = 0; l < 3; ++l) {
        const int b = m->d.head_buf[l];
        m->lv[l].buf = m->bptr[b];
        m->lv[l].H = m->net_h >> (3 + l);
        m->lv[l].W = m->net_w >> (3 + l);
        m->lv[l].stride = 8 << l;
        m->lv[l].anchor0 = a0;
        a0 += m->lv[l].H * m->lv[l].W;
    }
    m->A = a0;
    if (m->A >= 65536) PA_FAIL(e, "%d anchors per image exceed the 16-bit sort key", m->A);
    m->P2 = 1;
    while (m->P2 < m->A) m->P2 <<= 1;
    PA_HIP(e, hipMalloc((void**)&m->d_cand, (size_t)B * m->A * 6 * sizeof(float)));
    PA_HIP(e, hipMalloc((void**)&m->d_cidx, (size_t)B * m->A * sizeof(int32_t)));
    PA_HIP(e, hipMalloc((void**)&m->d_ccnt, (size_t)B * sizeof(int32_t)));
    PA_HIP(e, hipMalloc((void**)&m->d_keys, (size_t)B * m->P2 * sizeof(uint64_t)));
    PA_HIP(e, hipMalloc((void**)&m->d_order, (size_t)B * m->A * sizeof(int32_t)));
    PA_HIP(e, hipMalloc((void**)&m->d_supp, (size_t)B * m->A));
    PA_HIP(e, hipMalloc((void**)&m->d_oboxes, (size_t)B * 300 * 6 * sizeof(float)));
    PA_HIP(e, hipMalloc((void**)&m->d_ocnt, (size_t)B * sizeof(int32_t)));
    if (m->d.nk) PA_HIP(e, hipMalloc((void**)&m->d_okpts, (size_t)B * 300 * m->d.nk * sizeof(float)));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    m->p_h0 = h0; m->p_w0 = w0; m->p_imgsz = S; m->p_pre = p->pre_mode; m->p_auto = p->letterbox_auto;
    m->planned = true;
    return 0;
}

static ProfRec* prof_begin(pa_model* m, size_t idx, int kind, int ksize, double flops) {
    if (!m->e->profiling) return nullptr;
    if (m->prof.size() <= idx) {
        m->prof.resize(idx + 1);
        hipEventCreate(&m->prof[idx].e0);
        hipEventCreate(&m->prof[idx].e1);
    }
    ProfRec* r = &m->prof[idx];
    r->kind = kind; r->ksize = ksize; r->flops = flops; r->ms = 0.f;
    r->M = r->cout = r->cin = r->stride = r->mf = r->nf = r->res = 0;
    hipEventRecord(r->e0, m->e->stream);
    return r;
}
static void prof_end(pa_model* m, ProfRec* r) { if (r) hipEventRecord(r->e1, m->e->stream); }

// ConvArgs of conv op i for `n` images and the tile id it will be launched with (kernel choice: bf16x3 by default,
// tap kernels / the LDS cross-check kernel by tuning, conv_tap16 for fp16 models; a forced variant picks the tile of
// whichever kernel is selected).  An absorbed upsample (find_upsample_folds) is attached here: always for a 1x1
// consumer, for a 3x3 consumer only when the tile chosen is the patch kernel's.
static int conv_launch_args(const pa_model* m, size_t i, int n, ConvArgs& a) {
    const pa_engine* e = m->e;
    const pa_op_desc& o = m->ops[i];
    const pa_buf_desc& ob = m->bufs[o.out_buf];
    const pa_buf_desc& ib = m->bufs[o.in_buf];
    const int Ho = m->net_h >> ob.level, Wo = m->net_w >> ob.level;
    a.in = m->bptr[o.in_buf]; a.in_cs = ib.channels; a.in_choff = o.in_choff;
    a.out = m->bptr[o.out_buf]; a.out_cs = ob.channels; a.out_choff = o.out_choff;
    a.res = o.res_buf >= 0 ? m->bptr[o.res_buf] : nullptr;
    a.res_cs = o.res_buf >= 0 ? m->bufs[o.res_buf].channels : 0; a.res_choff = o.res_choff;
    a.zeros = e->zeros;
    a.w = m->d_w + o.w_off; a.bias = m->d_w + o.b_off;
    a.H = m->net_h >> ib.level; a.W = m->net_w >> ib.level; a.Ho = Ho; a.Wo = Wo;
    a.cin = o.cin; a.cout = o.cout; a.n16 = o.npad / 16; a.ksize = o.ksize; a.stride = o.stride; a.act = o.act;
    a.M = n * Ho * Wo;
    fill_fastdiv((unsigned)(Ho * Wo), &a.howo_magic, &a.howo_shift);
    fill_fastdiv((unsigned)Wo, &a.wo_magic, &a.wo_shift);
    a.tune = e->t.tune; a.tap_pd = e->t.tap_pd;
    const bool f16 = m->d.dtype == PA_DTYPE_F16, h2 = m->d.dtype == PA_DTYPE_H2;
    const bool use_tap = e->t.impl == 0 || f16;
    const bool use_bx3 = !f16 && !h2 && e->t.impl == 2 && o.reserved > 0;
    a.w3 = use_bx3 ? (const void*)(m->d_w + o.reserved) : nullptr;
    a.out_f32 = (f16 || h2) && (o.out_buf == m->d.head_buf[0] || o.out_buf == m->d.head_buf[1] || o.out_buf == m->d.head_buf[2]);
    if (h2) {
        a.oscale = m->d_w + o.reserved;
        a.ovf_flag = m->d_ovf;
        a.w_single = (o.flags & PA_CONV_W_SINGLE) && e->t.w_single ? 1 : 0;
        a.wr = (m->wr_valid && i < m->wr_off.size() && m->wr_off[i] >= 0) ? (const void*)(m->d_wr + m->wr_off[i]) : nullptr;
        const int lv = e->t.variant >= 0 ? e->t.variant : choose_conv_h2_variant(a);
        if (fold_active(m, (int)i) && (o.ksize == 1 || (lv >= 300 && lv < 400 && conv_h2p_supported(a)))) {
            const pa_op_desc& u = m->ops[m->fold_src[i]];       // the first up_c channels come from the coarse map
            a.in2 = m->bptr[u.in_buf]; a.in2_cs = m->bufs[u.in_buf].channels; a.in2_choff = u.in_choff; a.up_c = u.cin;
        }
        return lv;
    }
    const int lv = e->t.variant >= 0 ? e->t.variant
                   : f16 ? choose_conv_tap16_variant(a)
                   : use_bx3 ? choose_conv_bx3_variant(a)
                         : choose_conv_tap_variant(a.M, a.n16);
    if (use_bx3 && fold_active(m, (int)i) && (o.ksize == 1 || (lv >= 300 && lv < 400 && conv_bx3p_supported(a)))) {
        const pa_op_desc& u = m->ops[m->fold_src[i]];       // the first up_c channels come from the coarse map
        a.in2 = m->bptr[u.in_buf]; a.in2_cs = m->bufs[u.in_buf].channels; a.in2_choff = u.in_choff; a.up_c = u.cin;
    }
    return lv;
}

// op i is the stem, op i + 1 a 3x3 stride-2 conv over exactly the stem's channels, and nothing else reads the stem's output
static bool stem_fusable(const pa_model* m, size_t i) {
    if (i + 1 >= m->ops.size()) return false;
    const pa_op_desc& st = m->ops[i];
    const pa_op_desc& c = m->ops[i + 1];
    if (c.kind != PA_OP_CONV || c.ksize != 3 || c.stride != 2 || c.in_buf != st.out_buf || c.in_choff != st.out_choff || c.cin != st.cout ||
        c.res_buf >= 0)
        return false;
    for (int l = 0; l < 3; ++l) if (m->d.head_buf[l] == st.out_buf) return false;
    for (size_t k = 0; k < m->ops.size(); ++k) {
        if (k == i || k == i + 1) continue;
        const pa_op_desc& o = m->ops[k];
        if (o.kind != PA_OP_STEM && o.in_buf == st.out_buf) return false;
        if (o.kind == PA_OP_CONV && o.res_buf == st.out_buf) return false;
    }
    return true;
}

// replay the op list for `n` images (prof records appended starting at *pi)
static int run_ops(pa_model* m, int n, size_t* pi) {
    pa_engine* e = m->e;
    hipStream_t s = e->stream;
    for (size_t i = 0; i < m->ops.size(); ++i) {
        const pa_op_desc& o = m->ops[i];
        const pa_buf_desc& ob = m->bufs[o.out_buf];
        const int Ho = m->net_h >> ob.level, Wo = m->net_w >> ob.level;
        hipError_t r = hipSuccess;
        ProfRec* pr = nullptr;
        if (o.kind == PA_OP_CONV) {
            ConvArgs a{};
            const int lv = conv_launch_args(m, i, n, a);
            const bool f16 = m->d.dtype == PA_DTYPE_F16, h2 = m->d.dtype == PA_DTYPE_H2;
            const bool use_tap = e->t.impl == 0 || f16;
            const bool use_bx3 = a.w3 != nullptr;
            int bm = 0, bn = 0;
            if (h2 && (lv == 323 || lv == 324)) { bm = 128; bn = 96; }
            else if (h2 && lv == 325) { bm = 128; bn = 64; }
            else if (h2 && lv == 244) { bm = 128; bn = 96; }
            else if (h2 && (lv == 245 || lv == 246)) { bm = 128; bn = 192; }
            else if (h2 && (lv == 247 || lv == 248)) { bm = 64; bn = 192; }
            else if (h2 && lv >= 341 && lv <= 343) { bm = 256; bn = (lv - 340) * 16; }      // wide patch kernel: 16 x 16 pixels                       // quad patch kernel: 8 x 16 pixels x 96 channels
            else if (h2 && (lv == 243 || lv == 239)) conv_variant_shape(lv - 230, &bm, &bn);      // deep-ring tap tiles: the shape of 213 / 209
            else if ((f16 || h2) && lv >= 300) { bm = 128; bn = ((lv - 300) % 10) * 16; }
            else if (f16) conv_tap16_variant_shape(lv, &bm, &bn);
            else if (lv >= 300) { bm = 128; bn = (lv - 300) * 16; }          // patch kernel: 8 x 16 pixels x nf fragments
            else conv_variant_shape(lv >= 200 ? lv - 200 : lv, &bm, &bn);   // profile rows carry BM, BN of the workgroup tile
            pr = prof_begin(m, (*pi)++, o.kind, o.ksize, 2.0 * a.M * (double)o.cout * o.cin * o.ksize * o.ksize);
            if (pr) { pr->M = a.M; pr->cout = o.cout; pr->cin = o.cin; pr->stride = o.stride; pr->mf = bm; pr->nf = bn; pr->res = a.res != nullptr; }
            // tuning only ("timeline"): collect the s_memtime timeline of this launch into timeline_path
            unsigned long long* dbg_dev = nullptr;
            size_t dbg_bytes = 0;
            if (e->t.timeline && h2 && ((lv == 323 && conv_h2q_supported(a)) || (lv == 324 && conv_h2r_supported(a))) && !e->timeline_path.empty()) {
                const size_t patches = (size_t)n * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);          // conv_patch_h2q.hip: one record per workgroup of its 1-D grid
                dbg_bytes = 8 * ((patches + 7) / 8) * (size_t)((a.n16 + 5) / 6) * (8 + 4 * 32 * 5) * 8;          // kQDbgWords of conv_patch_h2q.hip (32-step ring)
            } else if (e->t.timeline && !h2 && use_tap && lv == 7 && o.ksize == 3 && !e->timeline_path.empty()) {
                dbg_bytes = (size_t)((a.M + 63) / 64) * ((o.npad + 95) / 96) * kConvDbgWords * 8;
            }
            if (dbg_bytes) {
                if (hipMalloc(&dbg_dev, dbg_bytes) == hipSuccess) (void)hipMemsetAsync(dbg_dev, 0, dbg_bytes, s);
                else dbg_dev = nullptr;
                a.dbg = dbg_dev;
            }
            if (h2) {
                r = launch_conv_h2(a, lv, s);
            } else if (f16) {
                r = launch_conv_tap16(a, lv, s);
            } else if (use_bx3) {
                r = launch_conv_bx3(a, lv, s);
            } else {
                r = launch_conv_tap(a, lv, s);
            }
            if (dbg_dev) {
                (void)hipStreamSynchronize(s);
                std::vector<unsigned long long> host(dbg_bytes / 8);
                (void)hipMemcpy(host.data(), dbg_dev, dbg_bytes, hipMemcpyDeviceToHost);
                if (FILE* f = fopen(e->timeline_path.c_str(), "wb")) { fwrite(host.data(), 1, dbg_bytes, f); fclose(f); }
                (void)hipFree(dbg_dev);
            }
        } else if (o.kind == PA_OP_STEM) {
            StemArgs a{};
            a.in = m->d_netin; a.w = m->d_w + o.w_off; a.bias = m->d_w + o.b_off;
            a.out = m->bptr[o.out_buf]; a.out_cs = ob.channels; a.out_choff = o.out_choff;
            a.H = m->net_h; a.W = m->net_w; a.Ho = Ho; a.Wo = Wo; a.cout = o.cout; a.B = n;
            a.out_f16 = m->d.dtype == PA_DTYPE_F16 ? 1 : m->d.dtype == PA_DTYPE_H2 ? 2 : 0;
            a.ovf_flag = m->d_ovf;
            pr = prof_begin(m, (*pi)++, o.kind, 3, 2.0 * n * Ho * Wo * (double)o.cout * 27);
            // tuning "fuse_stem": the stem and the stride-2 3x3 behind it (its only reader) as one kernel; the conv's own
            // turn in this loop is skipped (the profile shows both under the stem's record)
            if (e->t.fuse_stem && m->d.dtype == PA_DTYPE_H2 && stem_fusable(m, i)) {
                ConvArgs ca{};
                conv_launch_args(m, i + 1, n, ca);
                if (stem_l1_h2_supported(a, ca)) {
                    r = launch_stem_l1_h2(a, ca, s);
                    prof_end(m, pr);
                    if (r != hipSuccess) PA_FAIL(e, "fused stem + layer 1 launch failed: %s", hipGetErrorString(r));
                    ++i;
                    continue;
                }
            }
            r = launch_stem(a, s);
        } else if (o.kind == PA_OP_SPPF_POOL) {
            pr = prof_begin(m, (*pi)++, o.kind, 5, 0.0);
            r = launch_sppf_pool(m->bptr[o.in_buf], m->bufs[o.in_buf].channels, o.in_choff, o.cin, n, Ho, Wo, s, (int)m->d.dtype, e->t.fuse_sppf);
        } else if (o.kind == PA_OP_UPSAMPLE2X) {
            if (m->fold_dst[i] >= 0) {                   // absorbed by its consumer conv?  (same decision as at that conv)
                ConvArgs ca{};
                conv_launch_args(m, (size_t)m->fold_dst[i], n, ca);
                if (ca.in2) continue;
            }
            pr = prof_begin(m, (*pi)++, o.kind, 0, 0.0);
            r = launch_upsample2x(m->bptr[o.in_buf], m->bufs[o.in_buf].channels, o.in_choff, m->bptr[o.out_buf],
                                  ob.channels, o.out_choff, o.cin, n, Ho / 2, Wo / 2, s, (int)m->d.dtype);
        } else if (o.kind == PA_OP_MAXPOOL2) {
            pr = prof_begin(m, (*pi)++, o.kind, 2, 0.0);
            r = launch_maxpool2(m->bptr[o.in_buf], m->bufs[o.in_buf].channels, o.in_choff, m->bptr[o.out_buf],
                                ob.channels, o.out_choff, o.cin, n, Ho * 2, Wo * 2, s, (int)m->d.dtype);
        }
        prof_end(m, pr);
        if (r != hipSuccess) PA_FAIL(e, "op %zu (kind %d) launch failed: %s", i, o.kind, hipGetErrorString(r));
    }
    return 0;
}

// the operand-order weight copies of conv_patch_h2r.hip: allocated once, (re)built from the blob whenever it changed
static int ensure_operand_copies(pa_model* m) {
    if (m->wr_valid || m->d.dtype != PA_DTYPE_H2) return 0;
    pa_engine* e = m->e;
    // PA_CONV_W_SINGLE is the caller's promise that a conv's packed m plane is all zero; the two-product kernels skip the wm x ah
    // product on the strength of it.  Checked against the blob once per model (and again after a weight broadcast): a conv whose
    // promise does not hold fails the call instead of computing silently wrong results
    {
        bool any = false;
        PA_HIP(e, hipMemsetAsync(m->d_ovf + 32, 0, sizeof(unsigned) * 8, e->stream));        // scratch words of the 256-byte flag block
        for (size_t i = 0; i < m->ops.size(); ++i) {
            const pa_op_desc& o = m->ops[i];
            if (o.kind != PA_OP_CONV || !(o.flags & PA_CONV_W_SINGLE)) continue;
            const long long ksteps = o.ksize == 3 ? (long long)(o.cin / 32) * 9 + ((o.cin & 16) ? 5 : 0) : (long long)(o.cin + 31) / 32 * o.ksize * o.ksize;
            PA_HIP(e, launch_h2_mplane_check(m->d_w + o.w_off, (long long)o.npad * ksteps, m->d_ovf + 32, e->stream));
            any = true;
        }
        if (any) {
            unsigned bad = 0;
            PA_HIP(e, hipMemcpyAsync(&bad, m->d_ovf + 32, sizeof(bad), hipMemcpyDeviceToHost, e->stream));
            PA_HIP(e, hipStreamSynchronize(e->stream));
            if (bad) PA_FAIL(e, "a conv flagged PA_CONV_W_SINGLE has a non-zero m plane in the weight blob (pack it without the flag, or with fp16-number weights)");
        }
    }
    if (m->wr_off.empty()) {
        m->wr_off.assign(m->ops.size(), -1);
        size_t total = 0;
        for (size_t i = 0; i < m->ops.size(); ++i) {
            const pa_op_desc& o = m->ops[i];
            // conv_patch_h2r.hip: stride 1, whole chunks, at least two; conv_patch_h2v.hip: stride 1, 16 / 32 / 48 input channels;
            // stem_l1_h2.hip: the stride-2 layer behind the stem (16 / 32 / 48 input channels)
            // conv_1x1_h2s.hip: 1x1, stride 1, whole chunks, at least two
            if (o.kind != PA_OP_CONV) continue;
            const bool few = o.cin == 16 || o.cin == 32 || o.cin == 48;
            const bool whole = o.stride == 1 && (o.cin & 31) == 0 && o.cin >= 64;
            // conv_1x1_h2s.hip's stride-2 3x3 form: whole chunks, two products
            const bool s2 = o.ksize == 3 && o.stride == 2 && (o.cin & 31) == 0 && o.cin >= 32 && (o.flags & PA_CONV_W_SINGLE);
            if (!((o.ksize == 3 && (whole || few)) || s2 || (o.ksize == 1 && whole && (o.flags & PA_CONV_W_SINGLE)))) continue;
            m->wr_off[i] = (long long)total;
            total += conv_h2r_copy_bytes(o.npad / 16, o.cin, o.ksize);
        }
        if (total) {
            PA_HIP(e, hipMalloc((void**)&m->d_wr, total + 8192));          // the last chunk's look-ahead reads run 4 KB past a fragment
            PA_HIP(e, hipMemsetAsync(m->d_wr + total, 0, 8192, e->stream));
        }
    }
    for (size_t i = 0; i < m->ops.size(); ++i) {
        if (m->wr_off[i] < 0) continue;
        const pa_op_desc& o = m->ops[i];
        PA_HIP(e, launch_h2r_repack(m->d_w + o.w_off, m->d_wr + m->wr_off[i], o.npad / 16, o.cin, o.ksize, e->stream));
    }
    m->wr_valid = true;
    return 0;
}

// run_ops, or (tuning "graph", not while profiling) the replay of its capture for this batch size: the op list of
// an n-scale graph is ~100 launches of 10-40 us each, where per-launch host work shows
static int run_graph(pa_model* m, int n, size_t* pi) {
    pa_engine* e = m->e;
    if (ensure_operand_copies(m)) return 1;
    if (!e->t.graph || e->profiling || e->t.timeline) return run_ops(m, n, pi);
    if (m->graph_epoch != e->tuning_epoch) {            // kernel choice may have changed since the capture
        PA_HIP(e, hipStreamSynchronize(e->stream));
        for (auto& g : m->graphs) hipGraphExecDestroy(g.second);
        m->graphs.clear();
        m->graph_epoch = e->tuning_epoch;
    }
    auto it = m->graphs.find(n);
    if (it == m->graphs.end()) {
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        PA_HIP(e, hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
        size_t dummy = 0;
        const int rc = run_ops(m, n, &dummy);
        const hipError_t r = hipStreamEndCapture(e->stream, &g);
        if (rc) { if (g) hipGraphDestroy(g); return 1; }
        if (r != hipSuccess) PA_FAIL(e, "hipStreamEndCapture: %s", hipGetErrorString(r));
        const hipError_t ri = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (ri != hipSuccess) PA_FAIL(e, "hipGraphInstantiate: %s", hipGetErrorString(ri));
        it = m->graphs.emplace(n, ge).first;
    }
    PA_HIP(e, hipGraphLaunch(it->second, e->stream));
    return 0;
}

static int finish_profile(pa_model* m, size_t n_rec) {
    if (!m->e->profiling) { m->n_prof = 0; return 0; }
    for (size_t i = 0; i < n_rec && i < m->prof.size(); ++i)
        hipEventElapsedTime(&m->prof[i].ms, m->prof[i].e0, m->prof[i].e1);
    m->n_prof = n_rec;
    return 0;
}

enum { PROF_PRE = 100, PROF_DECODE = 101, PROF_NMS = 102 };

// decode + NMS + scale_boxes / scale_coords of the head maps in m->lv[] for nb images of a planned model, results copied to
// the caller's arrays (rows beyond max_det are never written on device).  oh x ow: the size upstream treats as the source
// (the PIL-resized image on the stretch path).
static int run_post(pa_model* m, const pa_yolo_params* p, int nb, int oh, int ow, size_t* ppi, float* out_boxes, float* out_kpts,
                int32_t* out_counts, int ovf_slot) {
    pa_engine* e = m->e;
    hipStream_t s = e->stream;
    size_t& pi = *ppi;
    ProfRec* pr = nullptr;
    hipError_t r = hipSuccess;
    const double gain = std::min((double)m->net_h / oh, (double)m->net_w / ow);
    const double kpx = (m->net_w - ow * gain) / 2, kpy = (m->net_h - oh * gain) / 2;
    // ---- decode + NMS
    DecodeArgs da{};
    for (int l = 0; l < 3; ++l) da.lv[l] = m->lv[l];
    da.cs = m->bufs[m->d.head_buf[0]].channels; da.nc = m->d.nc; da.nk = m->d.nk; da.kdim = m->d.kpt_dim;
    da.A = m->A; da.B = nb; da.conf = p->conf; da.classes = m->d_classes; da.n_classes = p->n_classes;
    da.cand = m->d_cand; da.cand_idx = m->d_cidx; da.cand_cnt = m->d_ccnt;
    pr = prof_begin(m, pi++, PROF_DECODE, 0, 0.0);
    r = launch_decode(da, s);
    prof_end(m, pr);
    if (r != hipSuccess) PA_FAIL(e, "decode launch failed: %s", hipGetErrorString(r));
    NmsArgs na{};
    na.cand = m->d_cand; na.cand_idx = m->d_cidx; na.cand_cnt = m->d_ccnt; na.keys = m->d_keys;
    na.order = m->d_order; na.supp = m->d_supp;
    for (int l = 0; l < 3; ++l) na.lv[l] = m->lv[l];
    na.cs = da.cs; na.nc = m->d.nc; na.nk = m->d.nk; na.kdim = m->d.kpt_dim; na.A = m->A; na.B = nb; na.P2 = m->P2;
    na.iou = p->iou; na.max_det = p->max_det; na.max_nms = 30000;
    na.gain = (float)gain;
    na.pad_x = (float)std::nearbyint(kpx - 0.1); na.pad_y = (float)std::nearbyint(kpy - 0.1);
    na.kpad_x = (float)kpx; na.kpad_y = (float)kpy;
    na.w0 = (float)ow; na.h0 = (float)oh;
    na.out_boxes = m->d_oboxes; na.out_kpts = m->d_okpts; na.out_cnt = m->d_ocnt;
    pr = prof_begin(m, pi++, PROF_NMS, 0, 0.0);
    r = launch_nms(na, s);
    prof_end(m, pr);
    if (r != hipSuccess) PA_FAIL(e, "nms launch failed: %s", hipGetErrorString(r));
    // ---- results back to the caller's arrays (rows beyond max_det are never written on device)
    if (m->d.dtype == PA_DTYPE_H2)      // the overflow flag travels with the results: pa_model_take_overflow needs no device round trip
        PA_HIP(e, hipMemcpyAsync(m->h_pin + ovf_slot, m->d_ovf, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    PA_HIP(e, hipMemcpyAsync(out_counts, m->d_ocnt, nb * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    PA_HIP(e, hipMemcpyAsync(out_boxes, m->d_oboxes,
                             (size_t)nb * p->max_det * 6 * sizeof(float), hipMemcpyDeviceToHost, s));
    if (m->d.nk)
        PA_HIP(e, hipMemcpyAsync(out_kpts, m->d_okpts,
                                 (size_t)nb * p->max_det * m->d.nk * sizeof(float), hipMemcpyDeviceToHost, s));
    return 0;
}

// argument checks + plan of a pa_yolo_infer / pa_yolo_submit call, class filter upload
static int yolo_prepare(pa_model* m, const uint8_t* frames, int n, int h, int w, const pa_yolo_params* p, float* out_boxes,
                        float* out_kpts, int32_t* out_counts, const char* who) {
    pa_engine* e = m->e;
    if (m->d.task != PA_TASK_DETECT && m->d.task != PA_TASK_POSE) PA_FAIL(e, "%s on a non-YOLO model", who);
    if (!frames || n <= 0 || h <= 0 || w <= 0 || !out_boxes || !out_counts) PA_FAIL(e, "%s: bad arguments", who);
    if (m->d.nk && !out_kpts) PA_FAIL(e, "%s: out_kpts is NULL for a pose model", who);
    if (p->max_det < 1 || p->max_det > 300) PA_FAIL(e, "max_det %d outside [1,300]", p->max_det);
    PA_HIP(e, hipSetDevice(e->dev));
    if (!m->planned || m->p_h0 != h || m->p_w0 != w || m->p_imgsz != p->imgsz || m->p_pre != p->pre_mode ||
        m->p_auto != p->letterbox_auto || m->p_batch != m->max_batch) {
        if (m->n_inflight) PA_FAIL(e, "%s: the plan would change (source size / imgsz / batch) with %d ticket(s) in flight", who, m->n_inflight);
        if (plan_yolo(m, h, w, p)) return 1;
    }
    if (p->n_classes > 0) {
        const bool same = (int)m->classes_host.size() == p->n_classes && !memcmp(m->classes_host.data(), p->classes, p->n_classes * sizeof(int32_t));
        if (!same) {
            // (the device list is read by queued decode kernels: replace it only once they are done)
            PA_HIP(e, hipStreamSynchronize(e->stream));
            if (p->n_classes > m->classes_cap) {
                if (m->d_classes) hipFree(m->d_classes);
                PA_HIP(e, hipMalloc((void**)&m->d_classes, p->n_classes * sizeof(int32_t)));
                m->classes_cap = p->n_classes;
            }
            m->classes_host.assign(p->classes, p->classes + p->n_classes);
            PA_HIP(e, hipMemcpy(m->d_classes, m->classes_host.data(), p->n_classes * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    return 0;
}

// everything one batch of nb <= max_batch frames needs, enqueued on the engine's stream: upload (host frames), preprocessing,
// network, decode, NMS, result copies (the overflow flag of h2 models into h_pin[ovf_slot]).  Does not wait.
static int yolo_enqueue(pa_model* m, const uint8_t* src, int nb, int h, int w, const pa_yolo_params* p, float* out_boxes,
                        float* out_kpts, int32_t* out_counts, int ovf_slot, size_t* ppi) {
    pa_engine* e = m->e;
    hipStream_t s = e->stream;
    size_t& pi = *ppi;
    const size_t frame_bytes = (size_t)h * w * 3;
    const int S = p->imgsz;
    // scale_boxes / scale_coords parameters (upstream treats the PIL-resized image as the source)
    const int oh = p->pre_mode == PA_PRE_PIL_STRETCH ? S : h, ow = p->pre_mode == PA_PRE_PIL_STRETCH ? S : w;
    if (!p->frames_on_device) {
        if (m->frames_cap < (size_t)nb * frame_bytes) {
            if (m->d_frames) hipFree(m->d_frames);
            m->frames_cap = (size_t)m->max_batch * frame_bytes;
            PA_HIP(e, hipMalloc((void**)&m->d_frames, m->frames_cap));
        }
        PA_HIP(e, hipMemcpyAsync(m->d_frames, src, (size_t)nb * frame_bytes, hipMemcpyHostToDevice, s));
        src = m->d_frames;
    }
    // ---- preprocessing -> u8 NHWC4 network input
    ProfRec* pr = prof_begin(m, pi++, PROF_PRE, 0, 0.0);
    hipError_t r = hipSuccess;
    if (p->pre_mode == PA_PRE_LETTERBOX || (h == S && w == S)) {
        LetterboxArgs a{};
        a.src = src; a.dst = m->d_netin; a.B = nb; a.h0 = h; a.w0 = w; a.rw = m->rw; a.rh = m->rh;
        a.top = m->top; a.left = m->left; a.nh = m->net_h; a.nw = m->net_w; a.mode = m->lb_mode;
        a.reverse = p->channel_reverse; a.xtab = m->d_xtab; a.ytab = m->d_ytab;
        r = launch_letterbox(a, s);
    } else {
        const uint8_t* cur = src; int ch = h, cw = w, cc = 3;
        if (w != S) {
            ResamplePassArgs a{};
            const bool last = (h == S);
            a.in = cur; a.out = last ? m->d_netin : m->d_tmp; a.B = nb; a.in_h = ch; a.in_w = cw; a.in_c = cc;
            a.out_h = ch; a.out_w = S; a.out_c = last ? 4 : 3; a.vertical = 0; a.bounds = m->d_hb; a.coefs = m->d_hk;
            a.ksize = m->hks; a.reverse = last ? p->channel_reverse : 0;
            r = launch_resample_pass(a, s);
            cur = m->d_tmp; cw = S;
        }
        if (r == hipSuccess && h != S) {
            ResamplePassArgs a{};
            a.in = cur; a.out = m->d_netin; a.B = nb; a.in_h = ch; a.in_w = cw; a.in_c = cc;
            a.out_h = S; a.out_w = cw; a.out_c = 4; a.vertical = 1; a.bounds = m->d_vb; a.coefs = m->d_vk;
            a.ksize = m->vks; a.reverse = p->channel_reverse;
            r = launch_resample_pass(a, s);
        }
    }
    prof_end(m, pr);
    if (r != hipSuccess) PA_FAIL(e, "preprocess launch failed: %s", hipGetErrorString(r));
    // ---- network
    if (run_graph(m, nb, &pi)) return 1;
    // ---- decode + NMS + results back to the caller's arrays
    return run_post(m, p, nb, oh, ow, &pi, out_boxes, m->d.nk ? out_kpts : nullptr, out_counts, ovf_slot);
}

int pa_yolo_infer(pa_model* m, const uint8_t* frames, int n, int h, int w, const pa_yolo_params* p,
                  float* out_boxes, float* out_kpts, int32_t* out_counts) {
    if (!m || !p) return 1;
    pa_engine* e = m->e;
    if (yolo_prepare(m, frames, n, h, w, p, out_boxes, out_kpts, out_counts, "pa_yolo_infer")) return 1;
    hipStream_t s = e->stream;
    const size_t frame_bytes = (size_t)h * w * 3;
    size_t pi = 0;
    for (int c0 = 0; c0 < n; c0 += m->max_batch) {
        const int nb = std::min(m->max_batch, n - c0);
        if (yolo_enqueue(m, frames + (size_t)c0 * frame_bytes, nb, h, w, p, out_boxes + (size_t)c0 * p->max_det * 6,
                         m->d.nk ? out_kpts + (size_t)c0 * p->max_det * m->d.nk : nullptr, out_counts + c0, PA_MAX_INFLIGHT, &pi))
            return 1;
        PA_HIP(e, hipStreamSynchronize(s));          // (also completes every ticket still in flight; their waits return at once)
        m->last_n = nb;
        if (m->d.dtype == PA_DTYPE_H2) { m->h_ovf |= m->h_pin[PA_MAX_INFLIGHT]; m->ovf_cached = true; }
    }
    finish_profile(m, pi);
    return 0;
}

int pa_yolo_submit(pa_model* m, const uint8_t* frames, int n, int h, int w, const pa_yolo_params* p,
                   float* out_boxes, float* out_kpts, int32_t* out_counts, int* ticket) {
    if (!m || !p || !ticket) return 1;
    pa_engine* e = m->e;
    if (!p->frames_on_device) PA_FAIL(e, "pa_yolo_submit: frames must be in HBM (frames_on_device = 1)");
    if (n > m->max_batch) PA_FAIL(e, "pa_yolo_submit: n = %d > max_batch %d", n, m->max_batch);
    if (e->profiling || e->t.timeline) PA_FAIL(e, "pa_yolo_submit: not while profiling (use pa_yolo_infer)");
    const int slot = m->next_ticket % PA_MAX_INFLIGHT;
    if (m->tk_busy[slot]) PA_FAIL(e, "pa_yolo_submit: %d tickets in flight (PA_MAX_INFLIGHT)", PA_MAX_INFLIGHT);
    if (yolo_prepare(m, frames, n, h, w, p, out_boxes, out_kpts, out_counts, "pa_yolo_submit")) return 1;
    size_t pi = 0;
    if (yolo_enqueue(m, frames, n, h, w, p, out_boxes, out_kpts, out_counts, slot, &pi)) {
        // part of the call may be queued already (preprocessing, some layers) and would write into the caller's arrays with
        // no ticket to wait on: drain before reporting the failure
        (void)hipStreamSynchronize(e->stream);
        return 1;
    }
    PA_HIP(e, hipEventRecord(m->tk_ev[slot], e->stream));
    m->tk_busy[slot] = true;
    ++m->n_inflight;
    m->last_n = n;
    m->n_prof = 0;
    m->ovf_cached = false;
    *ticket = m->next_ticket++;
    return 0;
}

int pa_yolo_wait(pa_model* m, int ticket, int* overflow) {
    if (!m) return 1;
    pa_engine* e = m->e;
    const int slot = ticket % PA_MAX_INFLIGHT;
    if (ticket < 0 || ticket >= m->next_ticket || ticket + PA_MAX_INFLIGHT < m->next_ticket || !m->tk_busy[slot])
        PA_FAIL(e, "pa_yolo_wait: ticket %d is not in flight", ticket);
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipEventSynchronize(m->tk_ev[slot]));
    m->tk_busy[slot] = false;
    --m->n_inflight;
    if (overflow) *overflow = (m->d.dtype == PA_DTYPE_H2 && m->h_pin[slot]) ? 1 : 0;
    return 0;
}

int pa_yolo_postprocess(pa_model* m, const float* const* heads, int n, int h, int w, const pa_yolo_params* p,
                        float* out_boxes, float* out_kpts, int32_t* out_counts) {
    if (!m || !p || !heads) return 1;
    pa_engine* e = m->e;
    if (m->d.task != PA_TASK_DETECT && m->d.task != PA_TASK_POSE) PA_FAIL(e, "pa_yolo_postprocess on a non-YOLO model");
    if (n <= 0 || n > m->max_batch || !out_boxes || !out_counts || (m->d.nk && !out_kpts)) PA_FAIL(e, "pa_yolo_postprocess: bad arguments");
    if (p->max_det < 1 || p->max_det > 300) PA_FAIL(e, "max_det %d outside [1,300]", p->max_det);
    PA_HIP(e, hipSetDevice(e->dev));
    if (!m->planned || m->p_h0 != h || m->p_w0 != w || m->p_imgsz != p->imgsz || m->p_pre != p->pre_mode ||
        m->p_auto != p->letterbox_auto || m->p_batch != m->max_batch)
        if (plan_yolo(m, h, w, p)) return 1;
    hipStream_t s = e->stream;
    if (p->n_classes > 0) {
        if (p->n_classes > m->classes_cap) {
            if (m->d_classes) hipFree(m->d_classes);
            PA_HIP(e, hipMalloc((void**)&m->d_classes, p->n_classes * sizeof(int32_t)));
            m->classes_cap = p->n_classes;
        }
        PA_HIP(e, hipMemcpyAsync(m->d_classes, p->classes, p->n_classes * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    const int cs = m->bufs[m->d.head_buf[0]].channels;
    for (int l = 0; l < 3; ++l) {
        if (!heads[l]) PA_FAIL(e, "pa_yolo_postprocess: heads[%d] is NULL", l);
        PA_HIP(e, hipMemcpyAsync(const_cast<float*>(m->lv[l].buf), heads[l], (size_t)n * m->lv[l].H * m->lv[l].W * cs * sizeof(float),
                                 hipMemcpyHostToDevice, s));
    }
    const int S = p->imgsz;
    const int oh = p->pre_mode == PA_PRE_PIL_STRETCH ? S : h, ow = p->pre_mode == PA_PRE_PIL_STRETCH ? S : w;
    size_t pi = 0;
    if (run_post(m, p, n, oh, ow, &pi, out_boxes, out_kpts, out_counts, PA_MAX_INFLIGHT)) return 1;
    PA_HIP(e, hipStreamSynchronize(s));
    m->last_n = n;
    finish_profile(m, pi);
    return 0;
}

int pa_yolo_head_shape(pa_model* m, int level, int* h, int* w, int* c) {
    if (!m->planned || level < 0 || level > 2) PA_FAIL(m->e, "pa_yolo_head_shape: no plan / bad level");
    *h = m->lv[level].H; *w = m->lv[level].W; *c = m->bufs[m->d.head_buf[0]].channels;
    return 0;
}

int pa_yolo_read_head(pa_model* m, int level, int n, float* out) {
    pa_engine* e = m->e;
    if (!m->planned || level < 0 || level > 2 || n > m->last_n) PA_FAIL(e, "pa_yolo_read_head: no plan / bad level / n");
    PA_HIP(e, hipSetDevice(e->dev));
    const size_t bytes = (size_t)n * m->lv[level].H * m->lv[level].W * m->bufs[m->d.head_buf[0]].channels * sizeof(float);
    PA_HIP(e, hipMemcpyAsync(out, m->lv[level].buf, bytes, hipMemcpyDeviceToHost, e->stream));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

int pa_yolo_netin_shape(pa_model* m, int* h, int* w) {
    if (!m->planned || m->d.task == PA_TASK_TRACKNET) PA_FAIL(m->e, "pa_yolo_netin_shape: no YOLO plan");
    *h = m->net_h; *w = m->net_w;
    return 0;
}

int pa_yolo_read_netin(pa_model* m, int n, uint8_t* out) {
    pa_engine* e = m->e;
    if (!m->planned || !m->d_netin || n < 1 || n > m->last_n || !out) PA_FAIL(e, "pa_yolo_read_netin: no plan / bad n");
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipMemcpyAsync(out, m->d_netin, (size_t)n * m->net_h * m->net_w * 4, hipMemcpyDeviceToHost, e->stream));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

int pa_model_plan_bytes(pa_model* m, size_t* arena_bytes, size_t* logical_bytes) {
    if (!m->planned) PA_FAIL(m->e, "pa_model_plan_bytes: no plan yet");
    if (arena_bytes) *arena_bytes = m->arena_bytes;
    if (logical_bytes) *logical_bytes = m->logical_bytes;
    return 0;
}

int pa_model_fill_arena(pa_model* m, int byte_value) {
    if (!m) return 1;
    pa_engine* e = m->e;
    if (!m->planned || !m->arena) PA_FAIL(e, "pa_model_fill_arena: no plan yet");
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipMemsetAsync(m->arena, byte_value & 0xFF, m->arena_bytes, e->stream));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

int pa_tracknet_infer(pa_model* m, const float* x, int n, int h, int w, int x_on_device, float* out, int out_on_device) {
    if (!m) return 1;
    pa_engine* e = m->e;
    if (m->d.task != PA_TASK_TRACKNET) PA_FAIL(e, "pa_tracknet_infer on a non-TrackNet model");
    if (!x || !out || n <= 0) PA_FAIL(e, "pa_tracknet_infer: bad arguments");
    PA_HIP(e, hipSetDevice(e->dev));
    m->ovf_cached = false;               // this call's kernels may raise the flag: the host copy is stale
    if (!m->planned || m->net_h != h || m->net_w != w || m->p_batch != m->max_batch) {
        PA_HIP(e, hipStreamSynchronize(e->stream));
        free_plan(m);
        m->net_h = h; m->net_w = w;
        if (plan_buffers(m, m->max_batch)) return 1;
        m->planned = true;
    }
    hipStream_t s = e->stream;
    const int cin = m->bufs[0].channels;
    const int ob = m->d.head_buf[0];
    const int cout = m->bufs[ob].channels;
    const size_t es_in = m->d.dtype == PA_DTYPE_F16 ? 2 : 4;     // fp16 graphs take their input as halves
    const bool h2 = m->d.dtype == PA_DTYPE_H2;                  // h2 graphs take fp32 and encode it on the device
    if (h2 && (cin & 15)) PA_FAIL(e, "pa_tracknet_infer: h2 input buffer has %d channels", cin);
    size_t pi = 0;
    for (int c0 = 0; c0 < n; c0 += m->max_batch) {
        const int nb = std::min(m->max_batch, n - c0);
        const size_t in_bytes = (size_t)nb * h * w * cin * es_in;
        const char* xs = reinterpret_cast<const char*>(x) + (size_t)c0 * h * w * cin * es_in;
        if (h2) {
            const float* src = reinterpret_cast<const float*>(xs);
            if (!x_on_device) {
                if (m->stage_cap < in_bytes) {
                    if (m->d_stage) hipFree(m->d_stage);
                    m->stage_cap = (size_t)m->max_batch * h * w * cin * 4;
                    PA_HIP(e, hipMalloc((void**)&m->d_stage, m->stage_cap));
                }
                PA_HIP(e, hipMemcpyAsync(m->d_stage, xs, in_bytes, hipMemcpyHostToDevice, s));
                src = m->d_stage;
            }
            const hipError_t er = launch_h2_encode(src, m->bptr[0], (long long)(in_bytes / 4), m->d_ovf, s);
            if (er != hipSuccess) PA_FAIL(e, "h2 encode launch failed: %s", hipGetErrorString(er));
        } else
        PA_HIP(e, hipMemcpyAsync(m->bptr[0], xs, in_bytes,
                                 x_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
        if (run_graph(m, nb, &pi)) return 1;
        const size_t ohw = (size_t)(h >> m->bufs[ob].level) * (w >> m->bufs[ob].level);
        const size_t out_bytes = (size_t)nb * ohw * cout * sizeof(float);
        PA_HIP(e, hipMemcpyAsync(out + (size_t)c0 * ohw * cout, m->bptr[ob], out_bytes,
                                 out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
        PA_HIP(e, hipStreamSynchronize(s));
    }
    finish_profile(m, pi);
    return 0;
}


// ------------------------------------------------------------------------------- ball session
struct pa_ball {
    pa_model* m = nullptr;
    int h = 0, w = 0;            // source frame size
    int B = 0;                   // max frames per feed == model max_batch
    int ring = 0;                // resized-frame ring slots (B + 7)
    long long fed = 0;           // frames fed since the last set_background
    bool have_bg = false;
    uint8_t *d_src = nullptr, *d_tmp = nullptr, *d_small = nullptr, *d_med_src = nullptr, *d_med = nullptr;
    uint8_t* d_mask = nullptr; float* d_heat = nullptr;
    float* d_Y = nullptr;        // [7 + B + 7][288][512][cs] window outputs (7 carry rows first)
    float* d_lut = nullptr;
    int32_t *d_hb = nullptr, *d_hk = nullptr, *d_vb = nullptr, *d_vk = nullptr;
    int hks = 0, vks = 0;
    int32_t *d_row0 = nullptr, *d_mode = nullptr; float* d_div = nullptr;
    int32_t *d_label = nullptr, *d_bbox = nullptr, *d_rect = nullptr;
    int cs = 0;
};

static const int BALL_H = 288, BALL_W = 512;

void pa_ball_destroy(pa_ball* b);

int pa_ball_create(pa_model* m, int src_h, int src_w, pa_ball** out) {
    if (!m || !out) return 1;
    pa_engine* e = m->e;
    if (m->d.task != PA_TASK_TRACKNET || (m->d.dtype != PA_DTYPE_F32 && m->d.dtype != PA_DTYPE_H2))
        PA_FAIL(e, "pa_ball_create: not an fp32 / h2 TrackNet model");
    if (m->bufs[0].channels != 32) PA_FAIL(e, "pa_ball_create: TrackNet input buffer must have 32 channels (27 + pad)");
    PA_HIP(e, hipSetDevice(e->dev));
    if (src_h <= 0 || src_w <= 0) PA_FAIL(e, "pa_ball_create: unsupported source size %dx%d", src_w, src_h);
    if (m->bufs[m->d.head_buf[0]].channels < 8)
        PA_FAIL(e, "pa_ball_create: TrackNet output has %d channels (< 8)", m->bufs[m->d.head_buf[0]].channels);
    pa_ball* b = new pa_ball();
    b->m = m; b->h = src_h; b->w = src_w; b->B = m->max_batch; b->ring = b->B + 7;
    b->cs = m->bufs[m->d.head_buf[0]].channels;
    const size_t HW = (size_t)BALL_H * BALL_W;
    struct Guard { pa_ball* b; bool ok = false; ~Guard() { if (!ok) pa_ball_destroy(b); } } guard{b};
    PA_HIP(e, hipMalloc((void**)&b->d_src, (size_t)b->B * src_h * src_w * 3));
    PA_HIP(e, hipMalloc((void**)&b->d_tmp, (size_t)b->B * src_h * BALL_W * 3));
    PA_HIP(e, hipMalloc((void**)&b->d_small, (size_t)b->ring * HW * 3));
    PA_HIP(e, hipMalloc((void**)&b->d_med_src, (size_t)src_h * src_w * 3));
    PA_HIP(e, hipMalloc((void**)&b->d_med, HW * 3));
    PA_HIP(e, hipMalloc((void**)&b->d_mask, (size_t)(b->B + 7) * HW));
    PA_HIP(e, hipMalloc((void**)&b->d_heat, (size_t)(b->B + 7) * HW * sizeof(float)));
    PA_HIP(e, hipMalloc((void**)&b->d_Y, (size_t)(b->B + 14) * HW * b->cs * sizeof(float)));
    PA_HIP(e, hipMalloc((void**)&b->d_row0, (b->B + 7) * sizeof(int32_t)));
    PA_HIP(e, hipMalloc((void**)&b->d_mode, (b->B + 7) * sizeof(int32_t)));
    PA_HIP(e, hipMalloc((void**)&b->d_div, (b->B + 7) * sizeof(float)));
    PA_HIP(e, hipMalloc((void**)&b->d_label, (size_t)(b->B + 7) * HW * sizeof(int32_t)));
    PA_HIP(e, hipMalloc((void**)&b->d_bbox, (size_t)(b->B + 7) * 4 * HW * sizeof(int32_t)));
    PA_HIP(e, hipMalloc((void**)&b->d_rect, (size_t)(b->B + 7) * 4 * sizeof(int32_t)));
    std::vector<float> lut(256);
    for (int i = 0; i < 256; ++i) lut[i] = (float)((double)i / 255.0);     // float64 division, then .float()
    PA_HIP(e, hipMalloc((void**)&b->d_lut, 256 * sizeof(float)));
    PA_HIP(e, hipMemcpyAsync(b->d_lut, lut.data(), 256 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    std::vector<int32_t> bb, kk;
    if (src_w != BALL_W) { b->hks = pil_coeffs(src_w, BALL_W, bb, kk); PA_HIP(e, upload(e, &b->d_hb, bb)); PA_HIP(e, upload(e, &b->d_hk, kk)); }
    if (src_h != BALL_H) { b->vks = pil_coeffs(src_h, BALL_H, bb, kk); PA_HIP(e, upload(e, &b->d_vb, bb)); PA_HIP(e, upload(e, &b->d_vk, kk)); }
    if (src_h == BALL_H && src_w == BALL_W) {
        // identity "resample": one tap of weight 1.0 (1 << 22) per output row
        b->vks = 1;
        bb.assign((size_t)BALL_H * 2, 0);
        kk.assign((size_t)BALL_H, 1 << 22);
        for (int y = 0; y < BALL_H; ++y) { bb[2 * y] = y; bb[2 * y + 1] = 1; }
        PA_HIP(e, upload(e, &b->d_vb, bb)); PA_HIP(e, upload(e, &b->d_vk, kk));
    }
    guard.ok = true;
    *out = b;
    return 0;
}

void pa_ball_destroy(pa_ball* b) {
    if (!b) return;
    hipSetDevice(b->m->e->dev);
    hipStreamSynchronize(b->m->e->stream);
    void* ptrs[] = {b->d_src, b->d_tmp, b->d_small, b->d_med_src, b->d_med, b->d_mask, b->d_heat, b->d_Y, b->d_lut,
                    b->d_hb, b->d_hk, b->d_vb, b->d_vk, b->d_row0, b->d_mode, b->d_div, b->d_label, b->d_bbox, b->d_rect};
    for (void* p : ptrs) if (p) hipFree(p);
    delete b;
}

// Pillow bicubic resize of n u8 HWC images (h x w x 3) to 288 x 512 x 3, optional channel reversal
static int ball_resize(pa_ball* b, const uint8_t* src, int n, uint8_t* dst, int reverse) {
    pa_engine* e = b->m->e;
    hipStream_t s = e->stream;
    const uint8_t* cur = src;
    int cw = b->w;
    hipError_t r = hipSuccess;
    if (b->w != BALL_W) {
        ResamplePassArgs a{};
        const bool last = (b->h == BALL_H);
        a.in = cur; a.out = last ? dst : b->d_tmp; a.B = n; a.in_h = b->h; a.in_w = b->w; a.in_c = 3;
        a.out_h = b->h; a.out_w = BALL_W; a.out_c = 3; a.vertical = 0; a.bounds = b->d_hb; a.coefs = b->d_hk; a.ksize = b->hks;
        a.reverse = last ? reverse : 0;
        r = launch_resample_pass(a, s);
        cur = b->d_tmp; cw = BALL_W;
    }
    if (r == hipSuccess && b->h != BALL_H) {
        ResamplePassArgs a{};
        a.in = cur; a.out = dst; a.B = n; a.in_h = b->h; a.in_w = cw; a.in_c = 3;
        a.out_h = BALL_H; a.out_w = cw; a.out_c = 3; a.vertical = 1; a.bounds = b->d_vb; a.coefs = b->d_vk; a.ksize = b->vks;
        a.reverse = reverse;
        r = launch_resample_pass(a, s);
    }
    if (r == hipSuccess && b->w == BALL_W && b->h == BALL_H) {
        // source already 512x288: Pillow's resize is the identity, only the channel order may change
        // (letterbox kernel in copy mode writes 4-byte pixels, so use a 1-tap "resample" instead)
        ResamplePassArgs a{};
        a.in = src; a.out = dst; a.B = n; a.in_h = b->h; a.in_w = b->w; a.in_c = 3;
        a.out_h = BALL_H; a.out_w = BALL_W; a.out_c = 3; a.vertical = 1; a.bounds = b->d_vb; a.coefs = b->d_vk; a.ksize = b->vks;
        a.reverse = reverse;
        r = launch_resample_pass(a, s);
    }
    if (r != hipSuccess) PA_FAIL(e, "ball resize launch failed: %s", hipGetErrorString(r));
    return 0;
}

static int ball_finish_background(pa_ball* b);

int pa_ball_set_background(pa_ball* b, const uint8_t* median_rgb) {
    if (!b || !median_rgb) return 1;
    pa_engine* e = b->m->e;
    PA_HIP(e, hipSetDevice(e->dev));
    PA_HIP(e, hipMemcpyAsync(b->d_med_src, median_rgb, (size_t)b->h * b->w * 3, hipMemcpyHostToDevice, e->stream));
    return ball_finish_background(b);
}

static int ball_finish_background(pa_ball* b) {
    pa_engine* e = b->m->e;
    if (ball_resize(b, b->d_med_src, 1, b->d_med, 0)) return 1;
    PA_HIP(e, hipMemsetAsync(b->d_Y, 0, (size_t)(b->B + 14) * BALL_H * BALL_W * b->cs * sizeof(float), e->stream));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    b->fed = 0;
    b->have_bg = true;
    return 0;
}

int pa_ball_background_from_frames(pa_ball* b, const uint8_t* frames_bgr, int n, int on_device, uint8_t* out_median_rgb) {
    if (!b || !frames_bgr) return 1;
    pa_engine* e = b->m->e;
    if (n < 1 || n > 65535) PA_FAIL(e, "pa_ball_background_from_frames: n = %d", n);
    PA_HIP(e, hipSetDevice(e->dev));
    hipStream_t s = e->stream;
    const long long fb = (long long)b->h * b->w * 3;
    const uint8_t* src = frames_bgr;
    uint8_t* tmp = nullptr;
    if (!on_device) {
        PA_HIP(e, hipMalloc((void**)&tmp, (size_t)n * fb));
        hipError_t r = hipMemcpyAsync(tmp, frames_bgr, (size_t)n * fb, hipMemcpyHostToDevice, s);
        if (r != hipSuccess) { hipFree(tmp); PA_FAIL(e, "median upload: %s", hipGetErrorString(r)); }
        src = tmp;
    }
    hipError_t r = launch_median(src, n, fb, b->d_med_src, s);
    if (r == hipSuccess && out_median_rgb) r = hipMemcpyAsync(out_median_rgb, b->d_med_src, (size_t)fb, hipMemcpyDeviceToHost, s);
    if (r == hipSuccess) r = hipStreamSynchronize(s);
    if (tmp) hipFree(tmp);
    if (r != hipSuccess) PA_FAIL(e, "median kernel: %s", hipGetErrorString(r));
    return ball_finish_background(b);
}

int pa_ball_feed(pa_ball* b, const uint8_t* frames, int n, int on_device, int flush, uint8_t* out_masks,
                 float* out_heat, int32_t* out_rects, int* out_count) {
    if (!b || (!out_masks && !out_rects) || !out_count) return 1;
    pa_model* m = b->m;
    pa_engine* e = m->e;
    if (!b->have_bg) PA_FAIL(e, "pa_ball_feed: set the background first");
    if (b->B != m->max_batch)
        PA_FAIL(e, "pa_ball_feed: the model's max_batch changed (%d -> %d) after the session was created; create a new session",
                b->B, m->max_batch);
    if (n < 0 || n > b->B || (n > 0 && !frames)) PA_FAIL(e, "pa_ball_feed: n = %d (max %d)", n, b->B);
    PA_HIP(e, hipSetDevice(e->dev));
    m->ovf_cached = false;               // this call's kernels may raise the flag: the host copy is stale
    hipStream_t s = e->stream;
    const size_t HW = (size_t)BALL_H * BALL_W;
    if (!m->planned || m->net_h != BALL_H || m->net_w != BALL_W || m->p_batch != m->max_batch) {
        PA_HIP(e, hipStreamSynchronize(s));
        free_plan(m);
        m->net_h = BALL_H; m->net_w = BALL_W;
        if (plan_buffers(m, m->max_batch)) return 1;
        m->planned = true;
    }
    int nout = 0;
    std::vector<int32_t> row0, mode;
    std::vector<float> div;
    int nw = 0;
    size_t prof_n = 0;
    if (n > 0) {
        // 1. resize the new frames (BGR -> RGB) into the ring; a feed never wraps more than once
        const uint8_t* src = frames;
        if (!on_device) {
            PA_HIP(e, hipMemcpyAsync(b->d_src, frames, (size_t)n * b->h * b->w * 3, hipMemcpyHostToDevice, s));
            src = b->d_src;
        }
        const int slot0 = (int)(b->fed % b->ring);
        const int first = std::min(n, b->ring - slot0);
        if (ball_resize(b, src, first, b->d_small + (size_t)slot0 * HW * 3, 1)) return 1;
        if (first < n && ball_resize(b, src + (size_t)first * b->h * b->w * 3, n - first, b->d_small, 1)) return 1;
        const long long f_old = b->fed, f_new = b->fed + n;
        // 2. new complete windows g in [g_lo, g_hi]
        const long long g_lo = std::max(0ll, f_old - 7), g_hi = f_new - 8;
        nw = g_hi >= g_lo ? (int)(g_hi - g_lo + 1) : 0;
        if (nw > 0) {
            BallAssembleArgs aa{};
            aa.median = b->d_med; aa.frames = b->d_small; aa.lut = b->d_lut; aa.out = m->bptr[0];
            aa.B = nw; aa.H = BALL_H; aa.W = BALL_W; aa.ring = b->ring; aa.first_slot = (int)(g_lo % b->ring);
            aa.out_h2 = m->d.dtype == PA_DTYPE_H2;
            hipError_t r = launch_ball_assemble(aa, s);
            if (r != hipSuccess) PA_FAIL(e, "ball assemble launch failed: %s", hipGetErrorString(r));
            if (run_graph(m, nw, &prof_n)) return 1;
            PA_HIP(e, hipMemcpyAsync(b->d_Y + (size_t)7 * HW * b->cs, m->bptr[m->d.head_buf[0]],
                                     (size_t)nw * HW * b->cs * sizeof(float), hipMemcpyDeviceToDevice, s));
            for (int i = 0; i < nw; ++i) {             // frame g = g_lo + i: rows i .. i+7 (row r <-> window g_lo - 7 + r)
                const long long g = g_lo + i;
                row0.push_back(i);
                mode.push_back(g < 7 ? 1 : 0);
                div.push_back((float)(g + 1));
            }
        }
        b->fed = f_new;
    }
    if (flush && b->fed >= 8) {
        // tail: rows after the last window are zero (ball_tracker.py:486-509)
        PA_HIP(e, hipMemsetAsync(b->d_Y + (size_t)(7 + nw) * HW * b->cs, 0, (size_t)7 * HW * b->cs * sizeof(float), s));
        for (int fi = 1; fi < 8; ++fi) {
            row0.push_back(nw - 1 + fi);               // window index of the last sample is row (nw - 1) + 7
            mode.push_back(1);
            div.push_back((float)(8 - fi));
        }
    }
    nout = (int)row0.size();
    if (nout > 0) {
        PA_HIP(e, hipMemcpyAsync(b->d_row0, row0.data(), nout * sizeof(int32_t), hipMemcpyHostToDevice, s));
        PA_HIP(e, hipMemcpyAsync(b->d_mode, mode.data(), nout * sizeof(int32_t), hipMemcpyHostToDevice, s));
        PA_HIP(e, hipMemcpyAsync(b->d_div, div.data(), nout * sizeof(float), hipMemcpyHostToDevice, s));
        BallEnsembleArgs ea{};
        ea.Y = b->d_Y; ea.cs = b->cs; ea.H = BALL_H; ea.W = BALL_W; ea.row0 = b->d_row0; ea.mode = b->d_mode; ea.div = b->d_div;
        static const float w8[8] = {1.f, 2.f, 3.f, 4.f, 4.f, 3.f, 2.f, 1.f};
        for (int k = 0; k < 8; ++k) ea.w[k] = w8[k] / 20.0f;
        ea.threshold = 0.5f; ea.heat = out_heat ? b->d_heat : nullptr; ea.mask = b->d_mask;
        hipError_t r = launch_ball_ensemble(ea, nout, s);
        if (r != hipSuccess) PA_FAIL(e, "ball ensemble launch failed: %s", hipGetErrorString(r));
        if (out_rects) {
            BallLocateArgs la{};
            la.mask = b->d_mask; la.label = b->d_label; la.bbox = b->d_bbox; la.rect = b->d_rect; la.H = BALL_H; la.W = BALL_W;
            r = launch_ball_locate(la, nout, s);
            if (r != hipSuccess) PA_FAIL(e, "ball locate launch failed: %s", hipGetErrorString(r));
            PA_HIP(e, hipMemcpyAsync(out_rects, b->d_rect, (size_t)nout * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        }
        if (out_masks) PA_HIP(e, hipMemcpyAsync(out_masks, b->d_mask, (size_t)nout * HW, hipMemcpyDeviceToHost, s));
        if (out_heat) PA_HIP(e, hipMemcpyAsync(out_heat, b->d_heat, (size_t)nout * HW * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    // 3. carry the last 7 window rows to the front for the next feed: rows [nw, nw+7) -> [0, 7).  The ranges
    // overlap when nw < 7; copying row by row in ascending order is safe because dst row < src row.
    if (nw > 0) {
        for (int r7 = 0; r7 < 7; ++r7)
            PA_HIP(e, hipMemcpyAsync(b->d_Y + (size_t)r7 * HW * b->cs, b->d_Y + (size_t)(nw + r7) * HW * b->cs,
                                     HW * b->cs * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    PA_HIP(e, hipStreamSynchronize(s));
    finish_profile(m, prof_n);
    *out_count = nout;
    return 0;
}

int pa_ball_locate(pa_ball* b, const uint8_t* masks, int n, int32_t* out_rects) {
    if (!b || !masks || !out_rects) return 1;
    pa_engine* e = b->m->e;
    if (n < 1 || n > b->B + 7) PA_FAIL(e, "pa_ball_locate: n = %d (max %d)", n, b->B + 7);
    PA_HIP(e, hipSetDevice(e->dev));
    hipStream_t s = e->stream;
    const size_t HW = (size_t)BALL_H * BALL_W;
    PA_HIP(e, hipMemcpyAsync(b->d_mask, masks, (size_t)n * HW, hipMemcpyHostToDevice, s));
    BallLocateArgs la{};
    la.mask = b->d_mask; la.label = b->d_label; la.bbox = b->d_bbox; la.rect = b->d_rect; la.H = BALL_H; la.W = BALL_W;
    hipError_t r = launch_ball_locate(la, n, s);
    if (r != hipSuccess) PA_FAIL(e, "ball locate launch failed: %s", hipGetErrorString(r));
    PA_HIP(e, hipMemcpyAsync(out_rects, b->d_rect, (size_t)n * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    PA_HIP(e, hipStreamSynchronize(s));
    return 0;
}

int pa_model_profile_text(pa_model* m, char* buf, size_t cap) {
    size_t off = 0;
    for (size_t i = 0; i < m->n_prof && i < m->prof.size(); ++i) {
        const ProfRec& r = m->prof[i];
        int n = snprintf(buf + off, off < cap ? cap - off : 0, "%d,%d,%d,%d,%d,%d,%d,%d,%.5f,%.0f,%d\n", r.kind, r.ksize, r.M, r.cout,
                         r.cin, r.stride, r.mf, r.nf, r.ms, r.flops, r.res);
        if (n < 0 || off + n >= cap) break;
        off += n;
    }
    if (off < cap) buf[off] = 0;
    return (int)off;
}

int pa_model_last_profile(pa_model* m, int cap, int32_t* kinds, float* ms, double* flops, int32_t* ksizes) {
    int n = 0;
    for (size_t i = 0; i < m->n_prof && i < m->prof.size() && n < cap; ++i, ++n) {
        kinds[n] = m->prof[i].kind; ms[n] = m->prof[i].ms; flops[n] = m->prof[i].flops; ksizes[n] = m->prof[i].ksize;
    }
    return n;
}

// ------------------------------------------------------------------------------- RCCL (one-time weight broadcast)
// librccl is dlopen'ed on first use: a process that already holds torch's bundled librccl.so.1 gets that one
// (same SONAME), a standalone process the ROCm one; single-GPU users never load it.

struct pa_comm {
    void* lib = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static void* rccl_lib() {
    static void* lib = nullptr;
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    return lib;
}

int pa_comm_unique_id(void* out, size_t cap) {
    if (!out || cap < NCCL_UNIQUE_ID_BYTES) PA_FAIL((pa_engine*)nullptr, "pa_comm_unique_id: need %d bytes", NCCL_UNIQUE_ID_BYTES);
    void* lib = rccl_lib();
    if (!lib) PA_FAIL((pa_engine*)nullptr, "librccl.so.1 not found: %s", dlerror());
    auto get = (ncclResult_t (*)(ncclUniqueId*))dlsym(lib, "ncclGetUniqueId");
    if (!get) PA_FAIL((pa_engine*)nullptr, "ncclGetUniqueId missing");
    ncclUniqueId id;
    const ncclResult_t r = get(&id);
    if (r != ncclSuccess) PA_FAIL((pa_engine*)nullptr, "ncclGetUniqueId failed (%d)", (int)r);
    memcpy(out, &id, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int pa_engine_comm_init(pa_engine* e, const void* unique_id, size_t id_bytes, int nranks, int rank) {
    if (!e || !unique_id || id_bytes < NCCL_UNIQUE_ID_BYTES || nranks < 1 || rank < 0 || rank >= nranks)
        PA_FAIL(e, "pa_engine_comm_init: bad arguments");
    if (e->comm) PA_FAIL(e, "pa_engine_comm_init: communicator already initialised");
    void* lib = rccl_lib();
    if (!lib) PA_FAIL(e, "librccl.so.1 not found: %s", dlerror());
    pa_comm* c = new pa_comm();
    c->lib = lib; c->nranks = nranks; c->rank = rank;
    c->CommInitRank = (decltype(c->CommInitRank))dlsym(lib, "ncclCommInitRank");
    c->CommDestroy = (decltype(c->CommDestroy))dlsym(lib, "ncclCommDestroy");
    c->Broadcast = (decltype(c->Broadcast))dlsym(lib, "ncclBroadcast");
    c->AllReduce = (decltype(c->AllReduce))dlsym(lib, "ncclAllReduce");
    c->GetErrorString = (decltype(c->GetErrorString))dlsym(lib, "ncclGetErrorString");
    c->AllGather = (decltype(c->AllGather))dlsym(lib, "ncclAllGather");
    c->Send = (decltype(c->Send))dlsym(lib, "ncclSend");
    c->Recv = (decltype(c->Recv))dlsym(lib, "ncclRecv");
    c->GroupStart = (decltype(c->GroupStart))dlsym(lib, "ncclGroupStart");
    c->GroupEnd = (decltype(c->GroupEnd))dlsym(lib, "ncclGroupEnd");
    if (!c->CommInitRank || !c->CommDestroy || !c->Broadcast || !c->AllReduce || !c->GetErrorString || !c->AllGather || !c->Send || !c->Recv ||
        !c->GroupStart || !c->GroupEnd) {
        delete c;
        PA_FAIL(e, "librccl: missing symbols");
    }
    PA_HIP(e, hipSetDevice(e->dev));
    ncclUniqueId id;
    memcpy(&id, unique_id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = c->CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        const char* msg = c->GetErrorString(r);
        delete c;
        PA_FAIL(e, "ncclCommInitRank(%d/%d): %s", rank, nranks, msg);
    }
    e->comm = c;
    return 0;
}

void pa_engine_comm_destroy(pa_engine* e) {
    if (!e || !e->comm) return;
    hipSetDevice(e->dev);
    hipStreamSynchronize(e->stream);
    if (e->comm->comm) e->comm->CommDestroy(e->comm->comm);
    delete e->comm;
    e->comm = nullptr;
}

// in-place broadcast of device memory from `root` over the engine's communicator (xGMI inside a node)
int pa_engine_bcast(pa_engine* e, void* dev_ptr, size_t nbytes, int root) {
    if (!e || !dev_ptr) return 1;
    if (!e->comm) PA_FAIL(e, "pa_engine_bcast: call pa_engine_comm_init first");
    PA_HIP(e, hipSetDevice(e->dev));
    const ncclResult_t r = e->comm->Broadcast(dev_ptr, dev_ptr, nbytes, ncclUint8, root, e->comm->comm, e->stream);
    if (r != ncclSuccess) PA_FAIL(e, "ncclBroadcast: %s", e->comm->GetErrorString(r));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

// the one collective of the path: the packed weight blob goes from the rank that loaded the checkpoint to every
// other GPU, HBM to HBM (north_star "one-time RCCL broadcast of weights over xGMI")
int pa_engine_bcast_weights(pa_engine* e, pa_model* m, int root) {
    if (!e || !m || m->e != e) return 1;
    m->wr_valid = false;
    return pa_engine_bcast(e, m->d_w, m->n_w * sizeof(float), root);
}

// The same broadcast with a separate SOURCE on the root: the root rank sends `src`'s blob (the model it loaded), every rank —
// the root included — receives into `dst` (a model created from a NULL blob).  Ranks other than the root pass src = NULL.
// On one GPU (nranks == 1) this is how a test proves that a blob that only ever travelled through RCCL gives bitwise the
// detections of the loaded one (BASELINE configs[3]: weights reach 7 of the 8 shards this way).
int pa_engine_bcast_weights_from(pa_engine* e, pa_model* src, pa_model* dst, int root) {
    if (!e || !dst || dst->e != e || (src && (src->e != e || src->n_w != dst->n_w))) return 1;
    if (!e->comm) PA_FAIL(e, "pa_engine_bcast_weights_from: call pa_engine_comm_init first");
    PA_HIP(e, hipSetDevice(e->dev));
    dst->wr_valid = false;
    const void* send = src ? src->d_w : dst->d_w;
    const ncclResult_t r = e->comm->Broadcast(send, dst->d_w, dst->n_w * sizeof(float), ncclUint8, root, e->comm->comm, e->stream);
    if (r != ncclSuccess) PA_FAIL(e, "ncclBroadcast: %s", e->comm->GetErrorString(r));
    PA_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

// The sharded runner's gather (include/padel_hip.h, ABI v5): variable-length host buffers of every rank to the root, over the
// communicator the library owns.  Lengths by ncclAllGather (pa_engine_gather_sizes), payload by one ncclSend per rank and
// nranks - 1 ncclRecv on the root inside one group (the root's own part is a host copy); device staging buffers live for the call.
int pa_engine_gather_sizes(pa_engine* e, size_t nbytes, uint64_t* sizes) {
    if (!e || !sizes) PA_FAIL(e, "pa_engine_gather_sizes: NULL argument");
    const int nranks = e->comm ? e->comm->nranks : 1;
    if (nranks == 1) { sizes[0] = nbytes; return 0; }
    PA_HIP(e, hipSetDevice(e->dev));
    pa_comm* c = e->comm;
    unsigned long long* d_sizes = nullptr;
    PA_HIP(e, hipMalloc((void**)&d_sizes, (size_t)(nranks + 1) * sizeof(unsigned long long)));
    const unsigned long long mine = nbytes;
    hipError_t h = hipMemcpyAsync(d_sizes + nranks, &mine, sizeof(mine), hipMemcpyHostToDevice, e->stream);
    ncclResult_t r = ncclSuccess;
    if (h == hipSuccess) r = c->AllGather(d_sizes + nranks, d_sizes, 1, ncclUint64, c->comm, e->stream);
    std::vector<unsigned long long> hs((size_t)nranks);
    if (h == hipSuccess && r == ncclSuccess) h = hipMemcpyAsync(hs.data(), d_sizes, (size_t)nranks * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream);
    if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
    hipFree(d_sizes);
    if (r != ncclSuccess) PA_FAIL(e, "ncclAllGather: %s", c->GetErrorString(r));
    if (h != hipSuccess) PA_FAIL(e, "pa_engine_gather_sizes: %s", hipGetErrorString(h));
    for (int k = 0; k < nranks; ++k) sizes[k] = hs[(size_t)k];
    return 0;
}

int pa_engine_gather(pa_engine* e, const void* send, size_t nbytes, void* recv, size_t recv_cap, const uint64_t* sizes, int root) {
    if (!e || !sizes || (nbytes && !send)) PA_FAIL(e, "pa_engine_gather: NULL argument");
    const int nranks = e->comm ? e->comm->nranks : 1, me = e->comm ? e->comm->rank : 0;
    if (root < 0 || root >= nranks) PA_FAIL(e, "pa_engine_gather: root %d of %d", root, nranks);
    if (sizes[me] != nbytes) PA_FAIL(e, "pa_engine_gather: sizes[%d] = %llu, nbytes = %zu", me, (unsigned long long)sizes[me], nbytes);
    size_t total = 0;
    for (int k = 0; k < nranks; ++k) total += (size_t)sizes[k];
    // the capacity check comes BEFORE the exchange and depends only on what every rank knows: a root that bails out alone would
    // leave the others inside their sends
    if (me == root && (recv_cap < total || (total && !recv))) PA_FAIL(e, "pa_engine_gather: recv capacity %zu < %zu", recv_cap, total);
    if (nranks == 1) {
        if (nbytes) memcpy(recv, send, nbytes);
        return 0;
    }
    PA_HIP(e, hipSetDevice(e->dev));
    pa_comm* c = e->comm;
    char* d_send = nullptr;
    char* d_recv = nullptr;
    hipError_t h = hipSuccess;
    ncclResult_t r = ncclSuccess;
    if (me != root && nbytes) {
        h = hipMalloc((void**)&d_send, nbytes);
        if (h == hipSuccess) h = hipMemcpyAsync(d_send, send, nbytes, hipMemcpyHostToDevice, e->stream);
    }
    if (me == root && total) h = hipMalloc((void**)&d_recv, total);
    // (an allocation failure still enters the group with nothing posted: the peers' sends then fail inside RCCL instead of hanging)
    r = c->GroupStart();
    if (h == hipSuccess && r == ncclSuccess) {
        if (me == root) {
            size_t off = 0;
            for (int k = 0; k < nranks && r == ncclSuccess; ++k) {
                if (k != root && sizes[k]) r = c->Recv(d_recv + off, (size_t)sizes[k], ncclUint8, k, c->comm, e->stream);
                off += (size_t)sizes[k];
            }
        } else if (nbytes) {
            r = c->Send(d_send, nbytes, ncclUint8, root, c->comm, e->stream);
        }
    }
    const ncclResult_t r2 = c->GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (h == hipSuccess && r == ncclSuccess && me == root) {
        size_t off = 0;
        for (int k = 0; k < nranks && h == hipSuccess; ++k) {
            const size_t nb = (size_t)sizes[k];
            if (k == root) { if (nb) memcpy((char*)recv + off, send, nb); }
            else if (nb) h = hipMemcpyAsync((char*)recv + off, d_recv + off, nb, hipMemcpyDeviceToHost, e->stream);
            off += nb;
        }
    }
    if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
    if (d_send) hipFree(d_send);
    if (d_recv) hipFree(d_recv);
    if (r != ncclSuccess) PA_FAIL(e, "ncclSend/Recv: %s", c->GetErrorString(r));
    if (h != hipSuccess) PA_FAIL(e, "pa_engine_gather: %s", hipGetErrorString(h));
    return 0;
}

// max over ranks of one double (bench: step time) — keeps the measurement inside the same communicator
int pa_engine_allreduce_max(pa_engine* e, double* value) {
    if (!e || !value) return 1;
    if (!e->comm) PA_FAIL(e, "pa_engine_allreduce_max: call pa_engine_comm_init first");
    PA_HIP(e, hipSetDevice(e->dev));
    double* d = nullptr;
    PA_HIP(e, hipMalloc((void**)&d, sizeof(double)));
    hipError_t h = hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, e->stream);
    ncclResult_t r = ncclSuccess;
    if (h == hipSuccess) r = e->comm->AllReduce(d, d, 1, ncclDouble, ncclMax, e->comm->comm, e->stream);
    if (h == hipSuccess && r == ncclSuccess) h = hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, e->stream);
    if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
    hipFree(d);
    if (r != ncclSuccess) PA_FAIL(e, "ncclAllReduce: %s", e->comm->GetErrorString(r));
    if (h != hipSuccess) PA_FAIL(e, "pa_engine_allreduce_max: %s", hipGetErrorString(h));
    return 0;
}

}  // extern "C"
