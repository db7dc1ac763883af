// Shared declarations for the gfx950 kernels and the host engine.  HIP only, wave64, no CUDA shims.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

// The 16-bit MFMA operand type of the whole library.  Default build: IEEE fp16 (11-bit significand: what the inference parity
// bounds are stated for).  -DMVD_BF16 (make bf16 -> libmvd_hip_bf16.so, selected with MVD_DTYPE=bf16): bfloat16 operands and
// storage, fp32 accumulation and master weights -- BASELINE configs[3]'s training dtype (fp32 range: no loss scaling needed;
// same MFMA rate on gfx950).  Every kernel is written against `half_t` and the two MFMA macros only.
#ifdef MVD_BF16
typedef __bf16 half_t;
#define MVD_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define MVD_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MVD_DTYPE_NAME "bf16"
#else
typedef _Float16 half_t;
#define MVD_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define MVD_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define MVD_DTYPE_NAME "f16"
#endif
typedef half_t h8 __attribute__((ext_vector_type(8)));
typedef half_t h4 __attribute__((ext_vector_type(4)));
typedef half_t h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MVD_MAX_TAPS 27

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2 };

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM descriptor:  out[row(m)][n] = epilogue( sum_{tap,c} A[in(m,tap)][c] * W[tap][n][c] )
// A is a channels-last activation tensor [B, PZ, PY, PX, lda] (fp16 or fp32) gathered through a
// tap table; W is fp16 [ntaps][N][Cin] (k-contiguous per output channel).
// ------------------------------------------------------------------------------------------------
// Which operand the workgroups of one XCD (one L2) should share.  Workgroup ids are dealt round-robin to the 8 XCDs and the kernels
// remap them so that an XCD owns a CONTIGUOUS range of tiles; walking that range row-tile-major makes an XCD re-use the activations
// of a few row tiles against every column group (all 8 L2s fill with all the weights), column-major the other way round.  Returns
// true when the column-major walk fills the L2s with clearly fewer bytes (deep levels: weights >> activations).
inline bool xcd_prefers_cols(int tiles_m, int groups_n, double a_bytes, double w_bytes) {
  const int nwg = tiles_m * groups_n, q = (nwg + 7) / 8;
  if (nwg < 16) return false;
  auto span = [](int q_, int inner, int outer) {  // distinct outer indices under q_ consecutive ids (inner index fastest), on average
    const double n = q_ % inner == 0 ? q_ / inner : (q_ - 1) / (double)inner + 1.0;
    return n < outer ? n : (double)outer;
  };
  const double at = a_bytes / tiles_m, wt = w_bytes / groups_n;
  const double row = at * span(q, groups_n, tiles_m) + wt * (q < groups_n ? q : groups_n);
  const double col = wt * span(q, tiles_m, groups_n) + at * (q < tiles_m ? q : tiles_m);
  return col < 0.8 * row;
}

struct IGemm {
  const void* a;        // activation base (channel offset already applied)
  int a_f32;            // 1: fp32 source (converted to fp16 while staging), 0: fp16 source
  int lda;              // elements between consecutive pixels
  int Cin;              // channels per tap (multiple of 8)
  int cin_alg;          // the layer's own channel count (Cin / 3 for extended-precision layers): algorithmic FLOP bookkeeping
  // logical output grid; M = B*Z*Y*X
  int B, Z, Y, X;
  // virtual input extent (after nearest upsample), strides, physical extent = virtual >> ups
  int IZ, IY, IX;
  int sz, sy, sx;
  int ups;
  int PZ, PY, PX;
  int ntaps;
  // per loop tap: (dz+1) | (dy+1)<<2 | (dx+1)<<4 | weight_slab<<8.  The weight slab is the tap's index in
  // the packed weights (identity except for the transposed-conv parity classes).
  int tap[MVD_MAX_TAPS];
  // weights
  const half_t* w;
  int N;                // GEMM N (before GEGLU halving)
  // output
  void* out;
  int out_f32;
  int ldc;
  int out_linear;       // 1: row offset = m*ldc ; 0: use the (OZ..oxo) mapping below
  int OZ, OY, OX, ozm, ozo, oym, oyo, oxm, oxo;
  // epilogue
  const float* bias;    // [N] or null
  const float* rowbias; // [B][rb_ld] or null (per-sample bias, e.g. timestep embedding)
  int rb_ld;
  const float* rowscale; // [B][rs_ld] or null: per-sample, per-column scale applied to the accumulator before the
  int rs_ld;             // biases (a GroupNorm folded into the GEMM that produces its input)
  // statistics-only pass (LDS-DMA kernel): nothing is stored; every 256-row tile writes (sum, sumsq) of the
  // accumulators per GroupNorm group to gn_partial[(m0/256) * (N/gn_cpg) + group][2].  Needs N % gn_cpg == 0,
  // gn_cpg in {8,16,32} and rows-per-sample % 256 == 0.
  float* gn_partial;
  int gn_cpg;
  const void* resid;    // same row mapping / ld as out, or null
  int resid_f32;
  int ldr;
  int out_split;        // fp16 output only, 0 or the logical width C: column n is written three times, as
                        // hi = fp16(v) at n and n + 2C and lo = fp16(v - hi) at n + C (ldc >= 3C): the [hi | lo | hi] operand
                        // of an extended-precision consumer (see ConvW::xp)
  int geglu;            // pairs 32-column blocks (x | gate): out cols = N/2
  float alpha;          // scale on the accumulator before bias
  int act;              // ACT_SILU applied last (non-GEGLU path)
  // split-K
  int nch;              // dense GEMM kernel: column tiles walked by one workgroup
  int xcd_cols;         // set by the launchers (xcd_prefers_cols): an XCD's workgroups share COLUMN tiles (weights) instead of row tiles
  // parity-batched launch (LDS-DMA kernel only): blockIdx.z picks one of npar tap tables / output offsets, so the
  // 8 parity classes of a transposed conv (or the 4 of a folded upsample conv) are ONE launch
  int npar;
  int par_ntaps[8];
  int par_tap[8][8];
  int par_oz[8], par_oy[8], par_ox[8];
  int par_walk;         // LDS-DMA kernel: one workgroup walks the npar classes of its tile (grid.z = 1) instead of one per class
  const half_t* wx;     // the same weights as a conv3x fragment stream (k_conv3x.hip) packed for column tiles of wx_bn, or null
  int wx_bn;
  int bn;               // column-tile width (64 / 96 / 128 / 160); 0 = pick from N
  int bm;               // LDS-DMA GEMM only: row-tile height, 256 (0 = 256: eight waves) or 128 (four waves; plain GEMMs)
  int splitk;
  float* partial;       // [splitk][M][N] fp32 when splitk > 1
};

struct MvdStream {
  hipStream_t s;
};

#define HIP_CHECK_RET(expr)                                        \
  do {                                                             \
    hipError_t _e = (expr);                                        \
    if (_e != hipSuccess) return mvd_fail(hipGetErrorString(_e));  \
  } while (0)

int mvd_fail(const char* msg);  // records thread-local error text, returns -1
const char* mvd_error_text();

#define MVD_MAX_DEVICES 64
static inline int mvd_current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MVD_MAX_DEVICES) d = 0;
  return d;
}
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int igemm_tap(int dz, int dy, int dx, int slab) { return (dz + 1) | ((dy + 1) << 2) | ((dx + 1) << 4) | (slab << 8); }

// ---- kernel launchers (defined in the .hip files) -------------------------------------------------
int launch_igemm(const IGemm& g, hipStream_t s);
int igemm_pick_bn(int N, int geglu);
int launch_splitk_reduce(const IGemm& g, hipStream_t s);
bool conv3_halo_eligible(const IGemm& g);
int conv3_halo_tiles(const IGemm& g, int bn);
int launch_conv3_halo(const IGemm& g, hipStream_t s);
int igemm_pick_splitk(int M, int N, int ksteps, int bn);
bool conv3x_eligible(const IGemm& g, int bn);
size_t conv3x_stream_halfs(int N, int Cin, int bn);
int conv3x_pack(const half_t* w, int N, int Cin, int bn, half_t* stream, hipStream_t s);
int launch_conv3x(const IGemm& g, const half_t* stream, int bn, hipStream_t s);
bool gemm_dma_eligible(const IGemm& g);
bool gemm_dma_is_plain(const IGemm& g);
void gemm_dma_plan(int M, int N, int ksteps, int bn, int geglu, int* nch_out, int* splitk_out);
int launch_target_encoder(const float* x, const float* pre, int n_views, const half_t* const* w, const float* const* bias,
                          const int* cin, const float* const* gamma, const float* const* beta, float* feats, hipStream_t s);
void gemm_dma_plan_us(int M, int N, int ksteps, int geglu, int out_b, int res_b, int* bn_io, int* nch_out, int* splitk_out,
                      int* bm_out = nullptr);
int launch_gemm_dma(const IGemm& g, hipStream_t s);

int launch_gn_stats(const float* x, int ld, int B, int rows_per_sample, int C, int G, const float* preadd, int pld,
                    float* partial, int* nslabs_out, hipStream_t s);
int launch_gn_apply(const float* x, int ld, int B, int rows_per_sample, int C, int G, const float* preadd, int pld,
                    const float* partial, int nslabs, const float* gamma, const float* beta, float eps, int act,
                    half_t* out, int ldo, hipStream_t s, int split = 0);
int gn_max_slabs();
// slab / tile partials [B][nslabs][G][2] -> per-sample scale[c] = rstd*gamma, shift[c] = beta - mean*rstd*gamma
int launch_gn_finalize(const float* partial, int B, int nslabs, int rows_per_sample, int C, int G, const float* gamma,
                       const float* beta, float eps, float* scale, float* shift, int ld, hipStream_t s);
bool gn_group_eligible(int ld, int rows, int C, int G, int pld, int ldo);
// split != 0: out rows hold [hi | lo | hi] (3C halfs, ldo >= 3C), see IGemm::out_split
int launch_gn_group(const float* x, int ld, int B, int rows, int C, int G, const float* preadd, int pld, const float* gamma,
                    const float* beta, float eps, int act, half_t* out, int ldo, hipStream_t s, int split = 0,
                    int nslab = 1, size_t slab_stride = 0, const float* bias2 = nullptr, const float* resid = nullptr, int ldr = 0,
                    float* mat = nullptr, int ldm = 0);
// raw (optional): x itself as fp16, rows ld_raw halfs apart (widths 320 / 640 / 1280 / 2560, 16-byte aligned)
int launch_layernorm(const float* x, int rows, int C, const float* gamma, const float* beta, float eps,
                     half_t* out, hipStream_t s, half_t* raw = nullptr, int ld_raw = 0);
bool layernorm_slabs_takes(int C);
// x = nslab split-K slabs [rows][C] (slab_stride floats apart): their sum + bias + rowbias[row / T] + resid is written to mat (fp32)
// and normalised into out (fp16): the producing GEMM's reduce pass and the LayerNorm in one launch
int launch_layernorm_slabs(const float* slabs, int nslab, size_t slab_stride, int rows, int C, const float* bias, const float* rowbias,
                           int rb_ld, int T, const float* resid, int ldr, float* mat, const float* gamma, const float* beta, float eps,
                           half_t* out, hipStream_t s, half_t* raw = nullptr, int ld_raw = 0);
int launch_layernorm_f32(const float* x, long ldx, int rows, int C, const float* gamma, const float* beta, float eps,
                         float* out, hipStream_t s);
// q | k rows at `qk` (k at column offset heads*d), V row-major at `v` (row strides ldqk / ldv: all three normally live
// in one fused projection buffer).  Tstride (0 = T): rows between consecutive samples when the token axis is padded
// (CLIP: 257 tokens in 264 rows)
int launch_attention(const half_t* qk, int ldqk, const half_t* v, int ldv, half_t* out, int ldo, int B, int T, int heads,
                     int d, hipStream_t s, int Tstride = 0);
int launch_clip_patches(const float* img, int B, int H, int W, int S, int P, int Kp, half_t* out, hipStream_t s);
int launch_clip_tokens(const float* pe, const float* cls, const float* pos, int B, int T, int Tp, int C, float* x,
                       hipStream_t s);
int launch_scale_copy(const float* src, size_t n, float k, float* dst, hipStream_t s);
int launch_softmax_rows(const float* s, long rows, int cols, half_t* p, hipStream_t st);
// ldx (0 = Cc): halfs between consecutive rows of ctxn (the block's column slice of a context fold shared by one level)
int launch_depth_attn(const float* qk, const half_t* ctxn, half_t* z, int n_cond, int HW, int D, int Cc, int heads,
                      hipStream_t s, int split = 0, int nfill = 0, const half_t* fill_row = nullptr, int ldx = 0);
int launch_small_linear(const float* a, int lda, int rows, int K, const half_t* w, const float* bias, int N,
                        int act_in, float* out, int ldo, int accumulate, hipStream_t s);
int launch_timestep_embedding(const int64_t* t, int B, int dim, float* out, hipStream_t s);
int launch_nchw_to_nhwc(const float* in, int B, int C, int HW, float* out, int ldo, int cpad, hipStream_t s);
int launch_nhwc_to_nchw(const float* in, int ld, int B, int C, int HW, float* out, hipStream_t s);
// xp != 0: Cin is the packed width 3 * Cl and each row holds [w_hi | w_hi | w_lo] (w_hi = fp16(w), w_lo = fp16(w - w_hi))
int launch_pack_weight(const float* src, int N, int Cin, int taps, int transposed, int geglu, half_t* dst,
                       hipStream_t s, int cin_src = -1, int xp = 0);
// fp32 rows -> fp16 [hi | lo | hi] rows of 3C halfs
int launch_rows_f32_to_f16_split(const float* in, int lda, long rows, int C, half_t* out, hipStream_t s);
int launch_permute_geglu_bias(const float* src, int N, float* dst, hipStream_t s);
int launch_f32_to_f16(const float* in, half_t* out, size_t n, hipStream_t s);
int launch_rows_f32_to_f16(const float* in, int lda, long rows, int C, half_t* out, hipStream_t s);
int launch_pack_upconv_weight(const float* src, int N, int Cin, half_t* dst, hipStream_t s);
int launch_fill_rows_f16(half_t* out, int ld, int rows, const half_t* vec, int n, hipStream_t s);
int launch_fold_qk(const float* wq, const float* wk, int heads, int hd, int Cc, int I, float scale, half_t* out,
                   hipStream_t s, float* out32 = nullptr);
int launch_fold_ov(const float* wo, const float* wv, int heads, int hd, int Cc, int I, half_t* out, hipStream_t s,
                   float* out32 = nullptr);
// C[m][n] = sum_k A[m][k] B[k][n] in fp64 (row-major fp32 operands, row strides lda / ldb), written as fp16 (out16, row stride ldc)
// or fp32 (out32, row stride ldc): weight folds at finalize time
int launch_fold_mm(const float* a, int lda, const float* b, int ldb, int M, int N, int K, half_t* out16, int ldc, float* out32,
                   hipStream_t s);
int launch_relu_beta_tile(const float* beta, int Cc, int heads, half_t* out, hipStream_t s, int split = 0);
int launch_cfg_ddim(const float* eps_c, const float* eps_u, float scale, const float* x, const float* noise,
                    float sqrt_one_minus_at, float sqrt_at, float sqrt_aprev, float dir_coef, float sigma,
                    float* eps_out, float* x_prev, size_t n, hipStream_t s);
// conditioner
struct ViewCam {
  float P[12];     // 3x4 projection into the size x size feature map (rows of the 4x4 minus the last)
  float Pinv[12];  // 3x4 inverse mapping (perspective: inv(P); orthographic: see engine)
  float Kinv[9];   // orthographic only
  float near_, far_;
};
int launch_vertex_gather(const float* feats, const ViewCam* cams, const int* view_idx, int n_views, const float* verts, int Nv, int V,
                         float vol_len, int fsize, int persp, float* out, hipStream_t s);
int launch_fuse_views(const float* vf, int n_views, int Nv, int total_views, const float* w, const float* b,
                      float* out, int accumulate, hipStream_t s);
// w: [27][Cin][Cout]; wp: the same weights as matrix-core B fragments (launch_sparse_w_frag) or null -> one-site-per-workgroup kernel
int launch_sparse_conv(const float* in, const int* nbr, int n_out, int Cin, int Cout, const float* w, const float* wp,
                       const float* scale, const float* shift, float* out, hipStream_t s, int mask_nonrep = 0);
bool sparse_mfma_takes(int Cin, int Cout);
int launch_sparse_w_frag(const float* w, int Cl_in, int Cl_out, int transposed, int flip, float* dst, hipStream_t s);
// BatchNorm1d in train mode over n rows of C <= 256 channels + ReLU, in place: x = relu((x - mean) / sqrt(var + eps) * g + b),
// biased variance (what F.batch_norm normalises with)
int launch_sparse_w_pack(const float* src, int Cin, int Cout, int layout, float* dst, hipStream_t s);
int launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C, float* scale,
                   float* shift, hipStream_t s);
int launch_bn_rows_relu(const float* x, float* y, int n, int C, const float* gamma, const float* beta, float eps, float* stats_out,
                        hipStream_t s, float* rmean = nullptr, float* rvar = nullptr, float momentum = 0.f);
int launch_mse(const float* a, const float* b, size_t n, float* out, hipStream_t s);
int launch_latent_gather(const float* feats, const int* grid, int gd, int gh, int gw, const float* min_xyz,
                         const int* out_sh, float voxel, int V, float vol_len, float* out, hipStream_t s);
int launch_sparse_densify(const float* feats, const int* grid, long nvox, int C, float* out, hipStream_t s);
int launch_frustum_gather(const float* vol, const ViewCam* cams, const int* view_idx, int TN, int D, int S, int V,
                          float vol_len, int persp, half_t* out, hipStream_t s);
int launch_bits_checksum(const void* p, size_t bytes, unsigned long long* out, hipStream_t s);
int launch_add_rows(float* dst, const float* a, const float* b, size_t n, hipStream_t s);
// out[(b * HW + p) * ldo + c] = in[(b * HW + p) * ldi + c] + img[p * C + c] for b < nb (C, ldi, ldo multiples of 4)
int launch_add_image_rows(const float* in, int ldi, const float* img, int C, int nb, int HW, float* out, int ldo, hipStream_t s);
// nearest x2 upsample + fp16 cast: in fp32 [B][H][W][ldi] -> out fp16 [B][2H][2W][C]
int launch_upsample2_f16(const float* in, int ldi, int B, int H, int W, int C, half_t* out, hipStream_t s);
// 3 x 3, pad 1, stride 1 convolution of a channels-last fp32 image batch with 4 or 8 input channels, in exact fp32 on the vector ALU:
// x [B][H][W][ldx], w [N][cin_src][3][3] (the checkpoint's layout), out [B][H][W][ldo] (+ bias)
int launch_conv_in_f32(const float* x, int ldx, const float* w, int cin_src, const float* bias, int N, int B, int H, int W, float* out,
                       int ldo, hipStream_t s);
// 3 x 3, pad 1, stride 1 convolution C -> N <= 4 of a channels-last fp32 image batch in exact fp32 on the vector ALU (the UNet's
// output head): a [B][H][W][C], w [N][C][3][3] (the checkpoint's layout), out [B][H][W][ldo] (+ bias).  C % 32 == 0, W % 16 == 0.
int launch_out_conv_f32(const float* a, int C, const float* w, const float* bias, int N, int B, int H, int W, float* out, int ldo,
                        hipStream_t s);
int launch_probe_null(hipStream_t s);  // one wave that does nothing: the launch path's own cost (ProbeScope calibration)
