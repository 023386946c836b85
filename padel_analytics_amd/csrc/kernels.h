// Internal kernel-launch interface of libpadel_hip.so (not part of the C-ABI).
// All activation tensors are fp32 NHWC in HBM: element (n, y, x, c) of a buffer with
// pixel stride `cs` lives at  base[((n*H + y)*W + x)*cs + c].  A layer reads / writes a
// channel slice [choff, choff+C) of such a buffer, which is how torch.cat / chunk of the
// YOLOv8 / TrackNet graphs are realised without ever copying.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace padel {

enum Act : int { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_LEAKY = 4 };   // LEAKY: nn.LeakyReLU() (slope 0.01): InpaintNet's Conv1DBlock (models.py:83-93)

struct ConvArgs {
    const float* in;      // input buffer base
    const float* w;       // packed weights [Npad][Ktot], K order = (c32 chunk, tap, c16 half)
    const void* w3;       // bf16x3 kernels: the same weights pre-split, [Npad][k-step][hi|mid|lo][32 bf16] (or nullptr)
    const float* bias;    // [Npad]
    const float* res;     // optional residual buffer base (added AFTER the activation) or nullptr
    const float* zeros;   // >= 64 bytes of zeros in HBM (source of padded taps)
    const float* in2;     // bf16x3 1x1 and patch kernels: channels [0, up_c) of the input are read from this buffer of HALF the
    int in2_cs, in2_choff, up_c;   // spatial size at [y >> 1][x >> 1] (an absorbed nn.Upsample(2)); nullptr: none; up_c % 32 == 0
    float* out;           // output buffer base
    int in_cs, in_choff;  // input pixel stride (channels) and channel offset of the slice read
    int out_cs, out_choff;
    int res_cs, res_choff;
    int H, W;             // input spatial size
    int Ho, Wo;           // output spatial size
    int cin;              // channels read (multiple of 16)
    int cout;             // real output channels (stores are masked to < cout)
    int n16;              // rows of the packed weight matrix / 16 (cout padded up to 16)
    int ksize, stride;    // 1 or 3 ; 1 or 2   (pad = ksize/2)
    int act;
    int M;                // batch * Ho * Wo
    int n_mtiles;         // ceil(M / BM)
    int n_ntiles;         // bx3 / fp16 kernels (1-D grid): channel tiles per pixel tile
    int tune;             // bit 0: s_setprio(1) around MFMA clusters; bit 1: staggered workgroup start
    int tap_pd;           // 1x1 tap kernel: prefetch distance 2 or 3 (pa_engine_set_tuning "tap_pd")
    int out_f32;          // fp16 / h2 kernels only: 1 = `out` is an fp32 buffer (convs that feed the Detect/Pose decode)
    const float* oscale;  // h2 kernels: [Npad] 1 / (power-of-two scale of the weight row), applied before the bias
    unsigned* ovf_flag;   // h2 kernels: set to 1 when an output value does not fit the fp16 range (h2_common.h)
    int w_single;         // h2 kernels: 1 = the m plane of the packed weights is all zero (PA_CONV_W_SINGLE): the wm x ah product, its
                          // weight requests and operand reads are skipped — two MFMAs per operand pair (same results: the product is 0)
    const void* wr;       // h2 stride-1 3x3 layers with whole chunks: the weights once more in MFMA operand order [fragment][k-step][h | m][lane][16 B]
                          // (conv_patch_h2r.hip; nullptr: no such copy)
    unsigned long long* dbg;   // tuning only (PADEL_CONV_DBG): per-workgroup s_memtime timeline, see conv_tap.hip
    // m / (Ho*Wo) and rem / Wo without an integer-division sequence (conv_tap.hip prologue): q = (umulhi(n, magic) + n) >> shift,
    // exact for 0 <= n < 2^31 (fill_fastdiv below; the conv kernels' rows satisfy n < 2^31)
    unsigned howo_magic, howo_shift, wo_magic, wo_shift;
};
inline void fill_fastdiv(unsigned d, unsigned* magic, unsigned* shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *magic = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1ull);
    *shift = s;
}
constexpr int kConvDbgSteps = 64;                       // k-steps kept per wave (ring)
constexpr int kConvDbgWords = 8 + 4 * kConvDbgSteps * 5;   // u64 words per workgroup: header + 4 waves x steps x 5 stamps

// The implicit-GEMM conv on v_mfma_f32_16x16x4_f32 (strict fp32: the cross-check / no-argument leg) is the tap-unrolled LDS-DMA
// ring of conv_tap.hip.  The register-staged LDS kernel of round 1 — its bitwise cross-check until round 5, 75-104 vs 109-124
// TFLOP/s (profiles/r5e_f32_tap_vs_lds.txt) — and three more retired generations live in tools/legacy_conv/ and are not
// built.  Tile variants of every family share one id space (conv_variant_shape).
bool conv_variant_shape(int variant, int* bm, int* bn);                         // false: unknown id
// conv_tap.hip reads up to 128 B past the last chunk of a pixel / weight row, so every buffer a conv reads is
// allocated with kConvReadSlack extra bytes; hipErrorNotSupported when the layer or the tile is not covered
hipError_t launch_conv_tap(const ConvArgs& a, int variant, hipStream_t s);      // ids 6,7,9..15,20
constexpr size_t kConvReadSlack = 512;
int choose_conv_tap_variant(int M, int n16);
// fp32 convolutions on the bf16 matrix pipe (conv_tap_bx3.hip): exact 3-way bf16 split, 6 products, fp32 accumulate
hipError_t launch_conv_bx3(const ConvArgs& a, int variant, hipStream_t s);      // ids 6,7,9,11,12,13,14,20,25 (3-stage ring), 206..225 (2-stage), 303/304/306 (patch kernel)
int choose_conv_bx3_variant(const ConvArgs& a);      // per-layer tile heuristic (ids + 200: 2-stage ring, 30x: patch kernel)
// stride-1 3x3, cin % 32 == 0: 8 x 16-pixel patch kernel (conv_patch_bx3.hip), the input patch is split once per chunk;
// nf = channel fragments per workgroup (3, 4, 6); reached through launch_conv_bx3 ids 303 / 304 / 306
bool conv_bx3p_supported(const ConvArgs& a);
hipError_t launch_conv_bx3p(const ConvArgs& a, int nf, hipStream_t s);
// h2 path (conv_tap_h2.hip, conv_patch_h2.hip; h2_common.h): activations are fp16 PAIRS (x ~ h + m / 2048) in 16-channel
// groups of 64 bytes [h x 16 | m x 16], 4 bytes per channel like fp32 (cs / choff count channels); weights `w` are the
// pre-split planes [Npad][k-step][h | m][32 fp16], `oscale` the inverse row scales; three f16 MFMAs per operand pair
hipError_t launch_conv_h2(const ConvArgs& a, int variant, hipStream_t s);        // tap tiles 207..225, patch tiles 303 / 304 / 306
int choose_conv_h2_variant(const ConvArgs& a);
bool conv_h2p_supported(const ConvArgs& a);
hipError_t launch_conv_h2p(const ConvArgs& a, int nf, hipStream_t s);
bool conv_h2q_supported(const ConvArgs& a);            // conv_patch_h2q.hip: stride-1 3x3, cin % 32 == 0, no absorbed upsample
hipError_t launch_conv_h2q(const ConvArgs& a, hipStream_t s);
bool conv_h2r_supported(const ConvArgs& a);            // conv_patch_h2r.hip (round 6): the quad tile with the weights global -> registers, one barrier per chunk; PA_CONV_W_SINGLE layers only
hipError_t launch_conv_h2r(const ConvArgs& a, int nf, hipStream_t s);      // nf 3: tile 324 (96 channels, two products), nf 2: tile 325 (64 channels, two or three products)
size_t conv_h2r_copy_bytes(int n16, int cin, int ksize);          // bytes of the operand-order copy of one conv's weights (both planes)
hipError_t launch_h2r_repack(const float* w, void* wr, int n16, int cin, int ksize, hipStream_t s);
bool conv_h2s_supported(const ConvArgs& a);            // conv_1x1_h2s.hip (round 6): 1x1, two products, register weights, one barrier per two k-steps
hipError_t launch_conv_h2s(const ConvArgs& a, bool nf12, hipStream_t s, bool m64 = false);
bool conv_h2s3_supported(const ConvArgs& a);           // the same machine for stride-2 3x3 layers (tile 246: 128 x 192, two products, whole chunks)
hipError_t launch_conv_h2s3(const ConvArgs& a, hipStream_t s, bool m64 = false);   // m64: 64 x 192 tiles of four waves (tile 248)      // tile 244: 128 x 96 (4 waves), 245: 128 x 192 (8 waves, one workgroup per CU)
// *flag |= 1 when any m-plane bit of `rows_x_ksteps` packed 128-byte k-step records is set (the PA_CONV_W_SINGLE promise, checked once per model)
hipError_t launch_h2_mplane_check(const float* w, long long rows_x_ksteps, unsigned* flag, hipStream_t s);
hipError_t launch_conv_h2_deep(const ConvArgs& a, int variant, hipStream_t s);     // conv_tap_h2p.hip: tap tiles with a 3-stage activation ring (239, 243); hipErrorNotSupported where they do not apply
bool conv_h2w_supported(const ConvArgs& a);            // conv_patch_h2w.hip: stride-1 3x3 with 16 / 32 / 48 input channels
hipError_t launch_conv_h2w(const ConvArgs& a, int nf, hipStream_t s);
bool conv_h2v_supported(const ConvArgs& a);            // conv_patch_h2v.hip (round 6): the wide patch tile with the weights global -> registers, one barrier per workgroup
hipError_t launch_conv_h2v(const ConvArgs& a, int nf, hipStream_t s);
// fp16 path (conv_tap16.hip): in / w / res / out are _Float16 arrays behind the float pointers of ConvArgs (cs and
// choff count elements); cin % 32 == 0; weights packed [Npad][Ktot] with K order (64-channel chunk, tap, 32-channel half)
hipError_t launch_conv_tap16(const ConvArgs& a, int variant, hipStream_t s);
int choose_conv_tap16_variant(const ConvArgs& a);
bool conv_tap16_variant_shape(int variant, int* bm, int* bn);
// fp16 stride-1 3x3 patch kernel (conv_patch16.hip), ids 303 / 304 / 306 of launch_conv_tap16
bool conv_p16_supported(const ConvArgs& a);
hipError_t launch_conv_p16(const ConvArgs& a, int nf, hipStream_t s);

struct StemArgs {
    const uint8_t* in;    // net input u8 NHWC4 [B][H][W][4]
    const float* w;       // [cout][27] (ky,kx,c) fused BN
    const float* bias;    // [cout]
    float* out;           // NHWC fp32 (or fp16 with out_f16), pixel stride out_cs
    int out_cs, out_choff;
    int H, W, Ho, Wo, cout, B;
    int out_f16;          // 1: `out` is a _Float16 buffer (fp16 models; the stem itself computes in fp32); 2: h2 pairs
    unsigned* ovf_flag;   // out_f16 == 2: raised when a value does not fit the fp16 range
    unsigned howo_magic, howo_shift, wo_magic, wo_shift;   // filled by launch_stem (fill_fastdiv): pixel index -> (n, oy, ox)
};
hipError_t launch_stem(const StemArgs& a, hipStream_t s);
struct ConvArgs;
bool stem_l1_h2_supported(const StemArgs& st, const ConvArgs& cv);        // stem_l1_h2.hip: stem + the 3x3 stride-2 conv behind it, one kernel
hipError_t launch_stem_l1_h2(const StemArgs& st, const ConvArgs& cv, hipStream_t s);

// SPPF: three chained MaxPool2d(5,1,2) of slice [choff, choff+c) written to the next three slices
// (f16 == 1 in these three: the buffers hold _Float16 elements; cs / choff / c count elements, c % 8 == 0;
//  f16 == 2: h2 pairs in 16-channel groups, 4 bytes per channel, cs / choff / c count channels)
hipError_t init_misc_kernels();
hipError_t launch_sppf_pool(float* buf, int cs, int choff, int c, int B, int H, int W, hipStream_t s, int f16 = 0, int fused = 1);
// nearest x2 upsample of a slice into a slice of a buffer with twice the spatial size
hipError_t launch_upsample2x(const float* in, int in_cs, int in_choff, float* out, int out_cs, int out_choff,
                             int c, int B, int H, int W, hipStream_t s, int f16 = 0);
// fp32 NHWC -> h2 pairs (n_floats % 16 == 0: whole 16-channel groups)
hipError_t launch_h2_encode(const float* in, float* out, long long n_floats, unsigned* ovf_flag, hipStream_t s);
// MaxPool2d(2,2)
hipError_t launch_maxpool2(const float* in, int in_cs, int in_choff, float* out, int out_cs, int out_choff,
                           int c, int B, int H, int W, hipStream_t s, int f16 = 0);

// ---- preprocessing -----------------------------------------------------------------
struct LetterboxArgs {
    const uint8_t* src;   // [B][h0][w0][3] u8
    uint8_t* dst;         // [B][nh][nw][4] u8 (4th byte 0)
    int B, h0, w0;        // source size
    int rw, rh;           // resized (unpadded) size
    int top, left;        // letterbox offsets
    int nh, nw;           // network input size
    int mode;             // 0 copy, 1 exact 2x2 area, 2 fixed-point bilinear (cv2 INTER_LINEAR)
    int reverse;          // 1: dst channel c = src channel 2-c
    const int32_t* xtab;  // mode 2: [rw][3] = {sx, a0, a1}
    const int32_t* ytab;  // mode 2: [rh][3] = {sy, b0, b1}
};
hipError_t launch_letterbox(const LetterboxArgs& a, hipStream_t s);

// Pillow-style separable resample pass over u8 images (coefficients precomputed on host):
// out[b][y][x][c] = clip8((sum_k coef[o][k] * in[...lo[o]+k...] + (1<<21)) >> 22)
struct ResamplePassArgs {
    const uint8_t* in; uint8_t* out;
    int B, in_h, in_w, in_c;      // input dims (in_c = 3 or 4 bytes per pixel)
    int out_h, out_w, out_c;      // output dims
    int vertical;                 // 0: horizontal pass (out_h == in_h), 1: vertical (out_w == in_w)
    const int32_t* bounds;        // [out][2] = {lo, n}
    const int32_t* coefs;         // [out][ksize]
    int ksize;
    int reverse;                  // reverse the 3 colour channels while writing
};
hipError_t launch_resample_pass(const ResamplePassArgs& a, hipStream_t s);

// ---- detect / pose post-processing -------------------------------------------------
struct HeadLevel { const float* buf; int H, W, stride, anchor0; };
struct DecodeArgs {
    HeadLevel lv[3];
    int cs;               // head pixel stride = 64 + nc + nk
    int nc, nk, kdim;
    int A, B;
    float conf;
    const int32_t* classes; int n_classes;   // device array or nullptr
    // outputs: per image candidate list
    float* cand;          // [B][A][6]  x1,y1,x2,y2 (net px), score, cls
    int32_t* cand_idx;    // [B][A] anchor index
    int32_t* cand_cnt;    // [B]
};
hipError_t launch_decode(const DecodeArgs& a, hipStream_t s);

struct NmsArgs {
    const float* cand; const int32_t* cand_idx; const int32_t* cand_cnt;
    uint64_t* keys;       // [B][P2] sort scratch (P2 = pow2 >= A)
    int32_t* order;       // [B][A] scratch
    uint8_t* supp;        // [B][A] scratch
    HeadLevel lv[3];
    int cs, nc, nk, kdim, A, B, P2;
    float iou; int max_det; int max_nms;
    // scale_boxes / scale_coords
    float gain; float pad_x, pad_y;        // rounded pads for boxes
    float kpad_x, kpad_y;                  // unrounded pads for keypoints
    float w0, h0;
    float* out_boxes;     // [B][max_det][6]
    float* out_kpts;      // [B][max_det][K*kdim] or nullptr
    int32_t* out_cnt;     // [B]
};
hipError_t launch_nms(const NmsArgs& a, hipStream_t s);

// ---- ball path (TrackNet) ---------------------------------------------------------------
struct BallAssembleArgs {
    const uint8_t* median;   // [H][W][3] RGB, resized background
    const uint8_t* frames;   // ring [ring][H][W][3] RGB, resized frames
    const float* lut;        // [256] u8 -> float(double(u)/255)
    float* out;              // [B][H][W][32] fp32 (or h2 pairs: out_h2)
    int B, H, W, ring, first_slot;
    int out_h2;              // 1: the TrackNet graph is an h2 graph (h2_common.h)
};
hipError_t launch_ball_assemble(const BallAssembleArgs& a, hipStream_t s);

struct BallEnsembleArgs {
    const float* Y;          // [rows][H][W][cs] window outputs (sigmoid heat maps), slot = channel
    int cs, H, W;
    const int32_t* row0;     // [nout] first buffer row of each output frame
    const int32_t* mode;     // [nout] 0 weighted, 1 mean
    const float* div;        // [nout] divisor for mode 1
    float w[8];              // ensemble weights
    float threshold;
    float* heat;             // [nout][H][W] or nullptr
    uint8_t* mask;           // [nout][H][W] 255 / 0
};
hipError_t launch_ball_ensemble(const BallEnsembleArgs& a, int nout, hipStream_t s);

struct BallLocateArgs {
    const uint8_t* mask;     // [nout][H][W] 255 / 0
    int32_t* label;          // scratch [nout][H][W]
    int32_t* bbox;           // scratch [nout][4][H][W]
    int32_t* rect;           // out [nout][4] = x, y, w, h  (w = -1: foreground list overflow)
    int H, W;
};
hipError_t launch_ball_locate(const BallLocateArgs& a, int nout, hipStream_t s);
// K11: per-byte median over N BGR frames -> RGB background (np.median + uint8 truncation)
hipError_t launch_median(const uint8_t* frames, int N, long long frame_bytes, uint8_t* out_rgb, hipStream_t s);

}  // namespace padel
