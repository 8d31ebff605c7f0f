// Internal host-side launchers shared between translation units (not part of the C ABI).
#pragma once
#include "common.hpp"

namespace p4c {

// conv_f32.hip
int prep_weights(const float* w, int CO, int CI, int ks, int transpose_flip, int M_pad, int K_pad, float* out,
                 hipStream_t stream);
int conv_fwd_f32(const float* in, int CI, const float* wp, int ks, const float* in_scale, const float* in_shift,
                 int in_relu, const float* bias, float* out, int out_cs, float* stat_partial, int B, int H, int W,
                 int m_blocks, hipStream_t stream);
int conv_wgrad_f32(const float* in, int CI, int ks, const float* in_scale, const float* in_shift, int in_relu,
                   const float* dout, float* partial, int G, int B, int H, int W, int CO, int CIreal, float* grad,
                   hipStream_t stream);
// floats of scratch needed by conv_wgrad_f32 (per-workgroup partials; 32-channel chunks use two slots per workgroup)
static inline int64_t wgrad_partial_floats(int CI_pad, int ks, int G) { return (int64_t)2 * G * ks * ks * CI_pad * 64; }

int wgrad_reduce(const float* partial, int nslots, int ks, int CI_pad, int ci_lo, int ci_hi, int CO, int CI, float* grad,
                 hipStream_t stream);
// Deferred reductions: while a collector is installed on the calling thread (WgradCollect), wgrad_reduce() records its job instead
// of launching, and wgrad_reduce_batch() reduces all of them in ONE launch -- every weight gradient of a network call (the HalfUNet
// backward plan: 14 dependent ~8 us launches per call on the weight-gradient stream otherwise).  Each job must own its partial
// buffer until the batch has run; jobs of one batch must write disjoint gradient elements.
struct WgradReduceJob {
    const float* partial;
    float* grad;
    int nslots, ntaps, CI_pad, ci_lo, ci_hi, CO, CI;
};
constexpr int WGRAD_BATCH_MAX = 24;
struct WgradCollect {
    WgradReduceJob job[WGRAD_BATCH_MAX];
    int n = 0;
};
void wgrad_collect_into(WgradCollect* c);   // nullptr: wgrad_reduce launches again
int wgrad_reduce_batch(const WgradCollect& c, hipStream_t stream);

// conv_bf16.hip (bf16 matrix cores, fp32 storage)
int prep_weights_bf16(const float* w, int CO, int CI, int ks, int transpose_flip, int M_pad, int K_pad, void* out,
                      hipStream_t stream);
// all weight tensors of a network in one launch (job table passed by value in the kernel arguments)
struct PrepJob {
    const float* w;   // canonical torch weight [CO][CI][ks][ks]
    void* out;        // prepared operand stream
    int CO, CI, ntaps, transpose_flip, M_pad, K_pad;
};
struct PrepBatch {
    PrepJob job[32];
    int n;
    int bf16;         // 1: bf16 stream (conv_bf16.hip), 0: fp32 stream (conv_f32.hip)
    unsigned int* zero_words;   // ticket counters of the in-kernel finalizes (below): cleared by the same launch
    int n_zero;
};

// BatchNorm statistics finished INSIDE the producing kernel: every workgroup leaves ONE slot of channel sums, takes a ticket,
// and the workgroup that draws the last ticket reduces the slots in a fixed order (bitwise reproducible, fp64 combine) and
// writes scale / shift / mean / rstd (+ the running statistics) -- what norm_finalize does in a launch of its own (a
// dependent ~7 us launch per convolution on the forward critical path).  slots == nullptr: off.
struct BatchFin {
    float* slots;              // [workgroups][2][64]
    unsigned int* ticket;      // zero between launches (the last workgroup resets it)
    const float* gamma;
    const float* beta;
    float* running_mean;       // may be null
    float* running_var;
    float* scale;              // (B,64) each
    float* shift;
    float* mean;
    float* rstd;
    double count;              // B*H*W
    float eps, momentum;
    int B;
};
// The same for the normalisation BACKWARD (BatchNorm): a kernel that leaves pass-1 slots [B][nblk][2][64] (norm_bwd_reduce,
// enc_out_bwd) lets its last workgroup do what norm_bwd_finalize does in a launch of its own -- d(gamma) += sum g * xhat,
// d(beta) += sum g, k1 / k2 (B,64) -- when there are few enough slots for one workgroup to sum (bwd_fin_max_slots(): the coarse
// levels, where the dependent ~6 us launch is pure latency on the backward chain).  ticket == nullptr: off.
struct BwdFin {
    unsigned int* ticket;      // zero between launches (the last workgroup resets it)
    const float* gamma;
    float* dgamma;
    float* dbeta;
    float* k1;                 // (B,64) each
    float* k2;
    float count;               // B*H*W
    int B, training;
};
int bwd_fin_max_slots();       // B * nblk up to which a producer finishes in place (P4C_BWD_INFIN_MAX; 0: never)
int prep_weights_batch(const PrepBatch& pb, hipStream_t stream);
int conv_bf16_stat_slots(int CI, int storage, int B, int H, int W, int ks = 3);
// `storage` = element type of in/out/dout in HBM (P4C_F32 or P4C_BF16)
// Pass 1 of a normalisation backward taken by the DATA-GRADIENT convolution that produces dA (ring kernel only): with
// y = the raw forward output the gradient belongs to and its normalisation rows, the kernel leaves in `stat_partial`
// [b][slot][2][64] the per-(workgroup, wave) sums of g = dA * [relu(y*scale+shift) alive] and of g * xhat -- what
// norm_bwd_reduce would compute from a read of dA and y (norm_pool.hip); the number of slots per sample goes to *nblk_out.
struct RingBwdStats {
    const void* y;            // (B,H,W,64) bf16
    const float* scale;       // (B,64) each
    const float* shift;
    const float* mean;
    const float* rstd;
};
constexpr int RING_BWD_STATS_MAXB = 21;
// Pass 2 of a normalisation backward applied by the CONSUMER of dY while it loads its operand (round 3: no norm_bwd_apply launch
// and no dY map for the blocks whose consumers can do this -- the data-gradient launches that do not also take pass 1 of the next
// normalisation, i.e. the blocks whose input is a pooled / summed map: conv 10, 2, 4, 6).  The operand map holds dA, and
//   dY = rstd * (gamma * g - k1 - xhat * k2),  g = dA * [y * scale + shift > 0],  xhat = (y - mean) * rstd
//      = alpha * g + beta * y + delta   per (sample, channel)
// with alpha = rstd * gamma, beta = -rstd^2 * k2, delta = rstd^2 * k2 * mean - rstd * k1 (k1, k2 from norm_bwd_finalize).
struct NormBwdCoef {
    const void* y;            // (B,H,W,64) bf16: the raw forward output of the block; nullptr = operand is dY itself
    const float* gamma;       // (64)
    const float* scale;       // (B,64) each: the forward normalisation of the block (ReLU mask)
    const float* shift;
    const float* rstd;
    const float* mean;
    const float* k1;
    const float* k2;
};
int conv_fwd_bf16(const void* in, int storage, int CI, const void* wp, int ks, const float* in_scale,
                  const float* in_shift, int in_relu, void* out, int out_cs, float* stat_partial, int B, int H, int W,
                  int m_blocks, hipStream_t stream, const BatchFin* fin = nullptr, const RingBwdStats* bst = nullptr,
                  int* nblk_out = nullptr, const NormBwdCoef* nb = nullptr);
// true when conv_fwd_bf16 / conv_wgrad_bf16 can take their 64-channel gradient operand through NormBwdCoef (row kernel for the data
// gradient, the role-split weight-gradient kernel)
bool conv_bf16_norm_bwd_fused_ok(int storage, int CI, int B, int H, int W);
// true when conv_fwd_bf16 with these arguments runs one of the two role-split 3x3 64->64 kernels (row-streaming
// conv3x3_bf16_rows, or the older tile ring conv3x3_bf16_ring for narrow maps): the ones that can finish BatchNorm themselves
bool conv_bf16_is_ring(int storage, int CI, int ks, int m_blocks, int out_cs, int B, int H, int W);
// true when that kernel can also take pass 1 of a normalisation backward (RingBwdStats) within NORM_BWD_MAX_BLOCKS slots
bool conv_bf16_bwd_stats_ok(int storage, int B, int H, int W);

// conv_rows.hip: the row-streaming 3x3 64->64 bf16 kernel (forward and data gradient)
bool conv_bf16_is_rows(int storage, int CI, int ks, int m_blocks, int out_cs, int B, int H, int W);
void conv_rows_geometry(int B, int H, int W, int* nstrips, int* nseg);
int conv_rows_stat_slots(int B, int H, int W);   // statistics slots per sample ([2][64] floats each)
int launch_conv3x3_bf16_rows(const void* in, const void* wp, int ks, const float* in_scale, const float* in_shift, int in_relu,
                             void* out, int out_cs, float* stat_partial, int B, int H, int W, hipStream_t stream, const BatchFin* fin,
                             const RingBwdStats* bst, int* nblk_out, const NormBwdCoef* nb = nullptr);
int conv_wgrad_bf16(const void* in, int storage, int CI, int ks, const float* in_scale, const float* in_shift, int in_relu,
                    const void* dout, float* partial, int G, int B, int H, int W, int CO, int CIreal, float* grad,
                    hipStream_t stream, const NormBwdCoef* nb = nullptr);
// conv_wgrad_rows.hip: the row-streaming weight gradient of the 3x3 convolution to 64 output channels, one chunk of <= 64 input channels
// per launch (maps at least 64 pixels wide, W % 64 == 0); partial: [*nslots_out <= G][9][part_cip][64] floats for wgrad_reduce
bool conv_wgrad_rows_ok(int storage, int in_cs, int ci_off, int dout_cs, int ks, int G, int B, int H, int W);
int launch_conv3x3_wgrad_bf16_rows(const void* x, const float* x_scale, const float* x_shift, int x_relu, const void* dout, float* partial,
                                   int G, int B, int H, int W, hipStream_t stream, const NormBwdCoef* nb, int* nslots_out, int in_cs = 64,
                                   int ci_off = 0, int ci_real = 64, int part_cip = 64);
// out_conv_bwd.hip: the whole backward of a 1x1 convolution from 64 channels in one pass -- data gradient dA (B,N,64), pass 1 of the
// normalisation backward of the block before it (statistics slots [B][slots][2][64], as RingBwdStats leaves them) and per-workgroup
// weight-gradient partials [B * slots][64 ci][64 co] for wgrad_reduce (ks = 1, CI_pad = 64)
bool out_conv_bwd_ok(int storage, int B, int64_t N);
int out_conv_bwd_slots(int B, int64_t N);   // workgroups per sample
int launch_out_conv_bwd(const void* dy, const void* wp_dgrad, const void* y, const float* scale, const float* shift, const float* mean,
                        const float* rstd, void* dA, float* stat_partial, float* wpartial, int B, int64_t N, hipStream_t stream,
                        int* nblk_out);
// plain convolution / its weight gradient on feature maps with fewer than 64 channels, in place (no padded copies): row kernel only
bool conv_wgrad_bf16_takes_nb(int storage, int CI, int ks, int B);
bool conv_rows_compact_ok(int storage, int in_cs, int out_cs, int ks, int B, int H, int W);
int launch_conv_bf16_rows_compact(const void* in, int in_cs, const void* wp, int ks, void* out, int out_cs, int B, int H, int W,
                                  hipStream_t stream);
int conv_wgrad_bf16_compact(const void* in, int in_cs, int ks, const void* dout, int dout_cs, float* partial, int G, int B, int H, int W,
                            int CO, int CIreal, float* grad, hipStream_t stream);

// conv_thin.hip: the first convolution at 65..72 input channels (the benchmark's 69) as a 64-channel row launch on channels 0..63 of
// the wide pixels + a tail pass that adds the product of the channels beyond 64 to y in place and takes the statistics of the result
bool first_conv_split_ok(int compute, int storage, int cin, int cin_pad, int B, int H, int W);
int first_conv_tail_slots(int B, int H, int W);     // statistics slots ([2][64] floats) per sample the tail leaves
int launch_first_conv_tail(const void* x, int x_cs, int cin, const float* w, void* y, float* stat_partial, int B, int H, int W, hipStream_t stream);
// conv_rows.hip: plain 3x3 convolution 64 -> 64 whose input pixels are `pixel_channels` (> 64, multiple of 8) channels apart (the
// first 64 channels of each are read)
int launch_conv_bf16_rows_wide_pixels(const void* in, int pixel_channels, const void* wp, void* out, int B, int H, int W, hipStream_t stream);

// norm_pool.hip
int norm_finalize(const float* partial, int tiles_per_sample, int B, int64_t hw, int mode, int groups, const float* gamma,
                  const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale,
                  float* shift, float* mean, float* rstd, hipStream_t stream);
int norm_eval(int B, const float* gamma, const float* beta, float eps, const float* running_mean,
              const float* running_var, float* scale, float* shift, float* mean, float* rstd, hipStream_t stream);
int norm_bwd_blocks(int64_t hw);
// `storage` = element type of activations / activation gradients in HBM (P4C_F32 or P4C_BF16)
int norm_bwd(int storage, const void* dA, const void* y, const float* scale, const float* shift, const float* mean,
             const float* rstd, const float* gamma, int relu, int B, int64_t hw, int mode, int groups, int training,
             float* partial, float* k1, float* k2, float* dgamma, float* dbeta, void* dY, hipStream_t stream, int pre_nblk = 0,
             unsigned int* fin_ticket = nullptr, bool pre_finalized = false);
// partial slots per sample the normalisation backward's buffers are sized for (standalone pass 1: <= 512; producers that take
// pass 1 themselves -- enc_out_bwd, the ring data-gradient convolution -- may leave up to this many)
constexpr int NORM_BWD_MAX_BLOCKS = 1024;
int pool_fwd(int storage, const void* y, const float* scale, const float* shift, int B, int H, int W, void* P,
             hipStream_t stream);
int upsum_fwd(int storage, const void* const* y, const float* const* scale, const float* const* shift, int B, int H, int W,
              void* S, hipStream_t stream);
// tx[k-1]: (B,H,W>>k,64) for k = 1..4
int up_bwd_x4(int storage, const void* dS, int B, int H, int W, void* const* tx, hipStream_t stream);
// upbwd_mfma.hip: the same pass for bf16 rows with W % 64 == 0 as one small GEMM per row strip (interpolation matrix x dS)
bool up_bwd_x4_mfma_ok(int B, int H, int W);
int launch_up_bwd_x4_mfma(const void* dS, int64_t rows, int W, void* const* tx, hipStream_t stream);
// mean / rstd / partial / nblk_out given (bf16 storage): the kernel also takes pass 1 of the normalisation backward of the level's
// second convolution on the dA it forms; *nblk_out = slots per sample left in `partial` (0: not fused, run the pass as usual)
int enc_out_bwd(int storage, const void* Tx, int Hfull, int s, const void* dS, const void* dP, const void* y,
                const float* scale, const float* shift, int B, int Hk, int Wk, void* dA, hipStream_t stream,
                const float* mean = nullptr, const float* rstd = nullptr, float* partial = nullptr, int* nblk_out = nullptr,
                const BwdFin* fin = nullptr, bool* finalized_out = nullptr);

// nodeproj.hip: queue of parameter-gradient reductions (mesh GNNs).  A job sums `slots` partials of n floats in a fixed order and ADDS the
// result into the gradient buffers p[] (+=): GRAD_JOB_MLP = the layout of mlp.hip's partials (p = dw1, dw2, db1, db2, dgamma, dbeta; ld[0] =
// row stride of dw1), GRAD_JOB_PROJ = up to three 64 x 64 blocks (p[i] with row stride ld[i]).  Launched at once, or -- while
// p4c_grad_reduce_defer(1) is in force -- queued until p4c_grad_reduce_flush reduces all queued jobs GRAD_BATCH per launch.
constexpr int GRAD_JOB_MLP = 0, GRAD_JOB_PROJ = 1;
struct GradReduceJob {
    const float* partial;
    int slots, n, kind, K;
    float* p[6];
    int ld[3];
    int k_real, o_real;
};
int grad_reduce_submit(const GradReduceJob& job, hipStream_t stream);
bool grad_reduce_deferring();      // p4c_grad_reduce_defer(1) is in force
// gemm.hip: the same deferral for the accumulating reductions of p4c_gemm_tn (weight gradients added into .grad buffers: the split-K
// slabs of the calls of one backward pass are reduced TN_BATCH per launch when the pass ends); driven by the entry points above
int tn_reduce_flush(hipStream_t stream);
int tn_reduce_pending();
void tn_reduce_drop();

// tiles of the conv kernels (for sizing the statistics partial buffers)
constexpr int CONV_TH = 4;
constexpr int CONV_TW = 32;
static inline int conv_tiles_per_sample(int H, int W) { return ((H + CONV_TH - 1) / CONV_TH) * ((W + CONV_TW - 1) / CONV_TW); }

}  // namespace p4c
