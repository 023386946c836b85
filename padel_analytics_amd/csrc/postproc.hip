// K8/K9 — Detect/Pose decode + confidence filter + per-image batched NMS + scale_boxes/scale_coords
// (gfx950).  Restates on device what the reference gets from
//   [upstream] ultralytics Detect/Pose inference branch + ops.non_max_suppression + torchvision.ops.nms
//   + ops.scale_boxes / scale_coords, reached from players_tracker.py:351-359 and
//   players_keypoints_tracker.py:285-292.
//
// Layout: the head of level l is an NHWC fp32 map [B][H_l][W_l][64 + nc + nk]; anchors are numbered
// P3,P4,P5 row-major (y outer) exactly like upstream's make_anchors.
//
// decode_kernel : one thread per (image, anchor).  Reads the nc class logits (contiguous), keeps the
//                 anchor iff max sigmoid > conf and its arg-max class is allowed, only then runs the
//                 DFL softmax-expectation and dist2bbox, and appends to the image's candidate list:
//                 wave ballot + prefix popcount behind ONE atomic per wave (order is restored by the sort
//                 in nms_kernel).
// nms_kernel    : one workgroup per image: bitonic sort of 64-bit keys (~score | anchor | slot) ==
//                 torchvision's stable descending sort; greedy IoU suppression as a bit matrix computed in
//                 parallel (<= 4096 ranked candidates; row blocks in LDS) consumed by a one-wave scan — the
//                 same decisions as the sequential loop, which stays as the path for longer lists; then
//                 the kept boxes / keypoints are rescaled to source-frame pixels and written in rank order.
#include "kernels.h"

namespace padel {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(256) decode_kernel(const DecodeArgs a) {
    const int b = blockIdx.y;
    const int ai = blockIdx.x * 256 + threadIdx.x;
    // no early exits: the compaction below is a wave-wide ballot (one atomic per WAVE, not per survivor)
    bool pass = ai < a.A;
    int l = 0;
    if (pass) { if (ai >= a.lv[2].anchor0) l = 2; else if (ai >= a.lv[1].anchor0) l = 1; }
    const HeadLevel lv = a.lv[l];
    const int pix = pass ? ai - lv.anchor0 : 0;
    const float* h = lv.buf + ((long long)b * lv.H * lv.W + pix) * a.cs;

    // class scores: max over sigmoid == sigmoid of max logit (monotone); arg-max in sigmoid space,
    // first index wins, like torch.max
    const float* cl = h + 64;
    float score = 0.0f;
    int cls = 0;
    if (pass) {
        float mx = cl[0];
        for (int c = 1; c < a.nc; ++c) mx = fmaxf(mx, cl[c]);
        score = sigmoidf_(mx);
        pass = score > a.conf;
    }
    if (pass) {
        for (int c = 0; c < a.nc; ++c) {
            if (sigmoidf_(cl[c]) == score) { cls = c; break; }
        }
        if (a.n_classes > 0) {
            bool ok = false;
            for (int k = 0; k < a.n_classes; ++k) ok |= (a.classes[k] == cls);
            pass = ok;
        }
    }
    float bx[4] = {0.f, 0.f, 0.f, 0.f};
    if (pass) {
        // DFL: softmax over 16 bins, expectation with arange(16)
        float d[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float v[16];
            float m = h[s * 16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = h[s * 16 + i]; m = fmaxf(m, v[i]); }
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = expf(v[i] - m); sum += v[i]; }
            float e = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) e += (v[i] / sum) * (float)i;
            d[s] = e;
        }
        const float ax = (float)(pix % lv.W) + 0.5f, ay = (float)(pix / lv.W) + 0.5f;
        const float st = (float)lv.stride;
        // dist2bbox(xywh=True) * stride, then xywh2xyxy (same op order as upstream)
        const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
        const float cx = ((x1 + x2) / 2.0f) * st, cy = ((y1 + y2) / 2.0f) * st;
        const float w = (x2 - x1) * st, hh = (y2 - y1) * st;
        const float hw = w / 2.0f, hhh = hh / 2.0f;
        bx[0] = cx - hw; bx[1] = cy - hhh; bx[2] = cx + hw; bx[3] = cy + hhh;
    }
    // stream compaction: survivors of a wave take consecutive slots behind ONE atomic (their order inside the list does
    // not matter: nms_kernel's sort key carries the anchor index)
    const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(&a.cand_cnt[b], __builtin_popcountll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (!pass) return;
    const int slot = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    float* o = a.cand + ((long long)b * a.A + slot) * 6;
    o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3]; o[4] = score; o[5] = (float)cls;
    a.cand_idx[(long long)b * a.A + slot] = ai;
}

// (a kernel, not hipMemsetAsync: between two kernels of a stream the runtime's memset left the GPU idle for 0.3 ms —
//  profiles/r4p_copy_trace_c3.txt)
__global__ void zero_i32_kernel(int32_t* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}

hipError_t launch_decode(const DecodeArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(zero_i32_kernel, dim3((a.B + 255) / 256), dim3(256), 0, s, a.cand_cnt, a.B);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    dim3 grid((a.A + 255) / 256, a.B, 1);
    hipLaunchKernelGGL(decode_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- NMS
#define NMS_THREADS 1024
#define NMS_WAVES (NMS_THREADS / 64)
#define NMS_LDS_KEYS 16384         // bitonic sort in LDS up to this many candidates per image (128 KB), in HBM beyond; the same LDS
                                   // holds the ranked boxes of a chunk afterwards
#define NMS_CHUNK 4096             // ranked candidates staged in LDS at a time (boxes 64 KB + areas 16 KB + slot ids 16 KB)

// `IoU(bi, bj) > iou` exactly as torchvision decides it — inter / union > iou in fp32 — with the IEEE division (a dozen
// instructions) only where it can matter: disjoint boxes (inter == 0: 0 / u = 0, or 0 / 0 = NaN, never > iou) skip everything;
// otherwise inter is compared with iou x union widened / narrowed by 1e-6 relative — four orders above the roundings involved
// (2^-24 for the product, 2^-24 for the quotient): beyond that band the quotient's comparison is decided, inside it the division
// decides.  iarea / jarea: (x2 - x1) * (y2 - y1) of the two boxes.
__device__ __forceinline__ bool nms_over(const float4 bi, const float iarea, const float4 bj, const float jarea, const float iou) {
    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
    const float ww = fmaxf(0.0f, xx2 - xx1), hh = fmaxf(0.0f, yy2 - yy1);
    const float inter = ww * hh;
    if (!(inter > 0.0f)) return false;
    const float uni = iarea + jarea - inter;
    const float t = iou * uni;
    if (inter > t * 1.000001f) return true;
    if (inter < t * 0.999999f) return false;
    return inter / uni > iou;
}

__global__ void __launch_bounds__(NMS_THREADS) nms_kernel(const NmsArgs a) {
    // Sort phase: up to 16 384 keys in LDS.  NMS phase: the same 128 KB hold the ranked boxes, areas and slot ids of a chunk.
    __shared__ uint64_t skeys[NMS_LDS_KEYS];
    float4* const sbox = reinterpret_cast<float4*>(skeys);                              // [NMS_CHUNK] class offset applied
    float* const sarea = reinterpret_cast<float*>(skeys + NMS_CHUNK * 2);               // [NMS_CHUNK]
    int* const sord = reinterpret_cast<int*>(skeys + NMS_CHUNK * 2 + NMS_CHUNK / 2);    // [NMS_CHUNK] candidate slot of each rank
    static_assert(NMS_CHUNK * (16 + 4 + 4) <= NMS_LDS_KEYS * 8, "chunk staging fits the sort buffer");
    __shared__ int skeep[300];                 // candidate slots of the kept detections, in rank order (max_det <= 300)
    __shared__ float4 kbox[300];               // their boxes (class offset applied) and areas: what later candidates are tested against
    __shared__ float karea[300];
    __shared__ unsigned long long smask[64];   // word of 64 ranks: bit j of smask[i] = IoU(rank i, rank j) > iou, j > i
    __shared__ unsigned long long s_sup;       // word of 64 ranks: suppressed by a box kept before the word
    __shared__ int s_kept, s_done;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int n = a.cand_cnt[b];
    if (n > a.A) n = a.A;
    const float* cand = a.cand + (long long)b * a.A * 6;
    const int32_t* cidx = a.cand_idx + (long long)b * a.A;
    int32_t* order = a.order + (long long)b * a.A;

    int p2 = 1;
    while (p2 < n) p2 <<= 1;
    uint64_t* keys = (p2 <= NMS_LDS_KEYS) ? skeys : (a.keys + (long long)b * a.P2);

    for (int i = tid; i < p2; i += NMS_THREADS) {
        uint64_t k = ~0ull;
        if (i < n) {
            const uint32_t sb = __float_as_uint(cand[i * 6 + 4]);     // score > 0: bits are order-preserving
            k = ((uint64_t)(~sb) << 32) | ((uint64_t)(uint32_t)cidx[i] << 16) | (uint32_t)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < p2; i += NMS_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = keys[i], y = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    int ne = n < a.max_nms ? n : a.max_nms;
    // ranked slot ids to HBM once (parallel): the LDS that held the keys is the staging area of the chunks below
    for (int i = tid; i < ne; i += NMS_THREADS) order[i] = (int)(keys[i] & 0xffffu);
    if (tid == 0) { s_kept = 0; s_done = 0; s_sup = 0ull; }
    __syncthreads();

    // Greedy suppression in rank order; boxes are offset by cls * 7680 like upstream (agnostic=False).  A candidate survives
    // iff no box KEPT before it overlaps it by more than iou (torchvision's definition), so only pairs (kept box, candidate)
    // matter: at most max_det x n of them, against the n^2 / 2 of the full IoU matrix rounds 2-4 built — with the
    // bench's 1 700 - 5 000 candidates per pose frame the matrix was 1.15 ms on the one CU an image has
    // (profiles/r4j_pmc_nms.txt); this walk is 0.37 ms (profiles/r4k_pmc_nms.txt).  The ranking is consumed a WORD of 64 candidates at a time (one per lane):
    //   1. every wave tests the word against a 16th of the boxes kept so far (leaving as soon as all 64 are suppressed) and
    //      computes four rows of the word's own 64 x 64 triangle by ballot;
    //   2. wave 0 walks the survivors serially (ctz over the not-removed bits), appending to the kept list.
    // The loop ends when max_det are kept (upstream slices the kept list to max_det: the same set).
    for (int c0 = 0; c0 < ne && !s_done; c0 += NMS_CHUNK) {
        const int nc = min(NMS_CHUNK, ne - c0);
        __syncthreads();                                      // the previous chunk's boxes are not read any more
        for (int i = tid; i < nc; i += NMS_THREADS) {
            const int slot = order[c0 + i];
            sord[i] = slot;
            const float* bi = cand + slot * 6;
            const float off = bi[5] * 7680.0f;
            const float4 bx = make_float4(bi[0] + off, bi[1] + off, bi[2] + off, bi[3] + off);
            sbox[i] = bx;
            sarea[i] = (bx.z - bx.x) * (bx.w - bx.y);
        }
        __syncthreads();
        for (int w0 = 0; w0 < nc; w0 += 64) {
            const int j = w0 + lane;
            const bool valid = j < nc;
            const float4 bj = sbox[valid ? j : w0];
            const float ja = sarea[valid ? j : w0];
            const unsigned long long vmask = __ballot(valid);
            // 1a. against the boxes kept before this word
            const int K = s_kept;
            bool sup = false;
            for (int k = wave; k < K; k += NMS_WAVES) {
                sup = sup || nms_over(kbox[k], karea[k], bj, ja, a.iou);
                if ((__ballot(sup) & vmask) == vmask) break;
            }
            const unsigned long long sw = __ballot(sup);
            if (lane == 0 && sw) atomicOr(&s_sup, sw);
            // 1b. the word's own triangle: rows 4 wave .. 4 wave + 3
#pragma unroll
            for (int r = 0; r < 64 / NMS_WAVES; ++r) {
                const int il = wave * (64 / NMS_WAVES) + r, i = w0 + il;
                unsigned long long bits = 0ull;
                if (i < nc) bits = __ballot(valid && lane > il && nms_over(sbox[i], sarea[i], bj, ja, a.iou));
                if (lane == 0) smask[il] = bits;
            }
            __syncthreads();
            // 2. serial walk over the survivors of the word
            if (wave == 0) {
                unsigned long long rem = s_sup | ~vmask;
                int k = K;
                bool done = false;
                unsigned long long cur = ~rem;
                while (cur) {
                    const int il = __builtin_ctzll(cur);
                    if (lane == 0) { skeep[k] = sord[w0 + il]; kbox[k] = sbox[w0 + il]; karea[k] = sarea[w0 + il]; }
                    ++k;
                    if (k >= a.max_det) { done = true; break; }
                    rem |= smask[il];
                    cur = il == 63 ? 0ull : (~rem & (~0ull << (il + 1)));
                }
                if (lane == 0) { s_kept = k; s_done = done ? 1 : 0; s_sup = 0ull; }
            }
            __syncthreads();
            if (s_done) break;
        }
    }
    const int kept = s_kept;
    __syncthreads();
    if (tid == 0) a.out_cnt[b] = kept;

    // rows beyond `kept` read as zeros on the host (the whole [max_det] block is copied back)
    const int nkv = a.nk;
    for (int k = kept * 6 + tid; k < a.max_det * 6; k += NMS_THREADS) a.out_boxes[(long long)b * a.max_det * 6 + k] = 0.0f;
    if (nkv > 0 && a.out_kpts)
        for (int k = kept * nkv + tid; k < a.max_det * nkv; k += NMS_THREADS) a.out_kpts[(long long)b * a.max_det * nkv + k] = 0.0f;
    // write kept detections in rank order, rescaled to the source frame
    for (int k = tid; k < kept; k += NMS_THREADS) {
        const int slot = skeep[k];
        const float* c = cand + slot * 6;
        float* o = a.out_boxes + ((long long)b * a.max_det + k) * 6;
        float x1 = (c[0] - a.pad_x) / a.gain, y1 = (c[1] - a.pad_y) / a.gain;
        float x2 = (c[2] - a.pad_x) / a.gain, y2 = (c[3] - a.pad_y) / a.gain;
        o[0] = fminf(fmaxf(x1, 0.0f), a.w0); o[1] = fminf(fmaxf(y1, 0.0f), a.h0);
        o[2] = fminf(fmaxf(x2, 0.0f), a.w0); o[3] = fminf(fmaxf(y2, 0.0f), a.h0);
        o[4] = c[4]; o[5] = c[5];
        if (nkv > 0 && a.out_kpts) {
            const int ai = cidx[slot];
            int l = 0;
            if (ai >= a.lv[2].anchor0) l = 2; else if (ai >= a.lv[1].anchor0) l = 1;
            const HeadLevel lv = a.lv[l];
            const int pix = ai - lv.anchor0;
            const float* h = lv.buf + ((long long)b * lv.H * lv.W + pix) * a.cs + 64 + a.nc;
            const float ax = (float)(pix % lv.W) + 0.5f, ay = (float)(pix / lv.W) + 0.5f;
            const float st = (float)lv.stride;
            float* ko = a.out_kpts + ((long long)b * a.max_det + k) * nkv;
            const int K = nkv / a.kdim;
            for (int q = 0; q < K; ++q) {
                float kx = (h[q * a.kdim] * 2.0f + (ax - 0.5f)) * st;
                float ky = (h[q * a.kdim + 1] * 2.0f + (ay - 0.5f)) * st;
                kx = (kx - a.kpad_x) / a.gain;
                ky = (ky - a.kpad_y) / a.gain;
                ko[q * a.kdim] = fminf(fmaxf(kx, 0.0f), a.w0);
                ko[q * a.kdim + 1] = fminf(fmaxf(ky, 0.0f), a.h0);
                if (a.kdim == 3) ko[q * a.kdim + 2] = sigmoidf_(h[q * a.kdim + 2]);
            }
        }
    }
}

hipError_t launch_nms(const NmsArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(nms_kernel, dim3(a.B), dim3(NMS_THREADS), 0, s, a);
    return hipGetLastError();
}

}  // namespace padel
