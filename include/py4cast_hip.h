/*
 * py4cast_hip.h -- C ABI of libpy4cast_hip.so: the MI355X (gfx950) implementation of
 * the py4cast autoregressive training hot path.
 *
 * The reference (meteofrance/py4cast) is 100 % Python and has no FFI; each entry point
 * below replaces a span of reference Python (cited as file:line, relative to the
 * reference repository) that today runs as a chain of PyTorch kernels.  The library is
 * bound with ctypes (see INTEGRATION.md for the stub a py4cast maintainer would add).
 *
 * Conventions
 *  - plain C, no C++/torch types.  Device buffers are raw pointers owned by the caller
 *    (the PyTorch caching allocator in practice), work-spaces included: the library
 *    allocates no device memory and frees nothing.
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); it never
 *    synchronises the device, so calls are hipGraph-capturable.
 *  - host-side state, all of it: (1) a lock-guarded table of "dynamic LDS size set" marks per
 *    (kernel, device) and the CU count per device; (2) p4c_halfunet_backward orders its weight
 *    gradients on a SIDE stream: by default one non-blocking stream + a handful of timing-disabled
 *    events per calling thread, created on first use and kept for the life of the thread -- pass
 *    your own with p4c_set_side_stream() and the library creates none; (3) the opt-in profiler
 *    below (process-wide, lock-guarded; events created by p4c_prof_enable only).  Everything else
 *    is reentrant and safe from several host threads / devices at once.
 *  - returns 0 on success, a negative P4C_ERR_* otherwise; p4c_last_error() returns a
 *    thread-local message.  Nothing throws or aborts across the boundary.
 *  - tensors are dense row-major; "N" is the number of grid points (H*W for grid models,
 *    ngrid for graph models: the rollout arithmetic is identical for both, the reference
 *    flattens the spatial dims of graph batches at lightning.py:526-535).
 *  - strides are in ELEMENTS.
 */
#ifndef PY4CAST_HIP_H
#define PY4CAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P4C_VERSION 100 /* 0.1.0 */

typedef void* p4c_stream_t; /* hipStream_t */

enum p4c_error {
    P4C_OK = 0,
    P4C_ERR_INVALID = -1, /* bad argument (shape, dtype, alignment, null pointer) */
    P4C_ERR_LAUNCH = -2,  /* kernel launch failed */
    P4C_ERR_RUNTIME = -3, /* HIP runtime call failed */
    P4C_ERR_UNSUPPORTED = -4
};

enum p4c_dtype { P4C_F32 = 0, P4C_BF16 = 1 };

/* loss kind = torch.nn class named in the yaml `losses[].params.loss` (losses.py:25-31) */
enum p4c_loss_kind { P4C_LOSS_MSE = 0, P4C_LOSS_L1 = 1 };

/* how the per-element mask of losses.py:144/195 is provided */
enum p4c_mask_mode {
    P4C_MASK_NONE = 0,    /* mask == 1 everywhere (lightning.py:797, mask_on_nan=False) */
    P4C_MASK_FROM_NAN = 1,/* mask = !isnan(target), target = nan_to_num(target) (lightning.py:792-796) fused */
    P4C_MASK_F32 = 2,     /* explicit float mask tensor, same shape as target */
    P4C_MASK_U8 = 3       /* explicit bool/uint8 mask tensor */
};

int p4c_version(void);
const char* p4c_last_error(void);
/* number of compute units of the current device (persistent-grid sizing, bench reporting) */
int p4c_num_cus(void);

/* Per-launch kernel timing for the bench harness (HIP events recorded on the launch stream, around
 * every launch of the tagged kernels, including those issued inside p4c_halfunet_forward/backward).
 * Disabled by default; p4c_prof_enable(mask, n) creates its 2n events up front (none in the timed region), (0,0) disables.  Not graph-capturable while on.
 * `units` of a record = output pixels (B*H*W) of that launch. */
enum p4c_prof_tag {
    P4C_PROF_CONV3X3_C64 = 1,     /* conv 3x3, 64 -> 64 channels: launches of the forward plan (nothing else runs beside them) */
    P4C_PROF_WGRAD3X3_C64 = 2,    /* weight gradient of the same (side stream of the backward plan: overlaps other kernels) */
    P4C_PROF_CONV3X3_C64_BWD = 4  /* the same conv kernel evaluating data gradients in the backward plan (overlapped likewise) */
};
int p4c_prof_enable(int tag_mask, int max_records);
/* only launches covering at least min_units units (output pixels) are recorded (reset to 0 by p4c_prof_enable) */
int p4c_prof_filter(int64_t min_units);
/* sums over the finished records of `tag` with units >= min_units; synchronises on their events */
int p4c_prof_collect(int tag, int64_t min_units, double* total_ms, int* count, double* total_units);

/* ------------------------------------------------------------------------------------
 * K1  build_x  -- replaces AutoRegressiveLightning._next_x (lightning.py:711-767) and the
 * layout handling around the model call (lightning.py:586-596).
 *
 *   x[b,n,:] = [ prev[b,0,n,:F], ..., prev[b,T_in-1,n,:F], statics[b,n,:Fs], forcing[b,n,:Ff],
 *                (mask_on_nan ? !any_nan(inputs,forcing)[b,n] : -) , 0-padding up to c_pad ]
 *   with NaN -> 0 in inputs/forcing when mask_on_nan (lightning.py:732-757, bit-exact).
 *   downscaling_only drops the prev block (lightning.py:759-762).
 * prev: (B,T_in,N,F) with element strides prev_bs/prev_ts (inner (N,F) dense);
 * statics: (B,N,Fs) with batch stride statics_bs (0 = broadcast one (N,Fs) map);
 * forcing: time-selected (B,N,Ff) with batch stride forcing_bs.
 * x: (B,N,c_pad) dense, dtype x_dtype; channels >= C_in are written as zeros.
 * Three kernels, the same values bit for bit: flat streams through an LDS tile of 64 rows (round 4: any feature counts <= 64 each,
 * 16-byte aligned sources with N * F / N * Fs / N * Ff multiples of 4, rows of x whole 16-byte slots, no NaN mask), a lane per
 * output quad (c_pad % 4 == 0), a lane per element (everything else, the NaN mask, the masked-auto-encoder block mask).
 */
int p4c_build_x(const float* prev, int64_t prev_bs, int64_t prev_ts, const float* statics, int64_t statics_bs,
                const float* forcing, int64_t forcing_bs, void* x, int x_dtype, int c_pad, int B, int T_in,
                int64_t N, int F, int Fs, int Ff, int mask_on_nan, int downscaling_only, p4c_stream_t stream);

/* Backward of K1 wrt prev: dprev[b,t,n,f] = dx[b,n,t*F+f]  (dprev dense (B,T_in,N,F)). */
int p4c_build_x_bwd(const void* dx, int dx_dtype, int c_pad, float* dprev, int B, int T_in, int64_t N, int F,
                    p4c_stream_t stream);

/* K1 with the masked-auto-encoder block mask of mask_tensor (lightning.py:769-785, applied at :580-581) fused in:
 * x[b, (yy,xx), c] *= 0 for every grid point whose block index (yy / block_h) * W + (xx / block_w) is set in
 * block_selected (H*W bytes, 1 = drawn by the caller's randperm; the draw itself stays with torch's CPU generator,
 * as in the reference).  The product with 0.0 is literal (x * False in torch: signed zeros, NaN stays NaN).
 * Grid layout only (N == H*W).  The backward is the same mask applied to dx. */
int p4c_build_x_masked(const float* prev, int64_t prev_bs, int64_t prev_ts, const float* statics, int64_t statics_bs,
                       const float* forcing, int64_t forcing_bs, void* x, int x_dtype, int c_pad, int B, int T_in,
                       int64_t N, int F, int Fs, int Ff, int mask_on_nan, int downscaling_only,
                       const uint8_t* block_selected, int H, int W, int block_h, int block_w, p4c_stream_t stream);
int p4c_build_x_bwd_masked(const void* dx, int dx_dtype, int c_pad, float* dprev, int B, int T_in, int64_t N, int F,
                           const uint8_t* block_selected, int H, int W, int block_h, int block_w, p4c_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K2  ar_update -- replaces lightning.py:599-633 (clone, scaled / plain residual update,
 * border forcing) in ONE pass, evaluated in the reference's operation order with FMA
 * contraction disabled so that fp32 results are bit-identical to the torch op chain:
 *
 *   p   = nan_to_num?(prev) * keep_prev + y * std + mean        (scaled_ar, std != NULL)
 *   p   = nan_to_num?(prev) * keep_prev + y                     (diff_ar / downscaling)
 *   new = border_mask * nan_to_num?(border_state) + interior_mask * p   (border_mask != NULL)
 *
 * prev/border_state/new_state: (B,N,F) with batch strides; y: (B,N,y_cs) dense where only the
 * first F channels of each row are used (y_cs >= F lets the model emit padded rows);
 * std/mean: (F); masks: (N) floats.  prev may be NULL when keep_prev == 0.
 */
int p4c_ar_update_fwd(const float* prev, int64_t prev_bs, const void* y, int y_dtype, int y_cs,
                      const float* border_state, int64_t border_bs, const float* std, const float* mean,
                      const float* border_mask, const float* interior_mask, float* new_state, int64_t new_bs,
                      int B, int64_t N, int F, float keep_prev, int nan_to_num, p4c_stream_t stream);

/* Backward of K2: dpred = dnew * interior_mask (or dnew when no border forcing);
 *   dy = dpred * std (or dpred); dprev = dpred * keep_prev.  dy: (B,N,y_cs), channels >= F zeroed.
 * dprev may be NULL. */
int p4c_ar_update_bwd(const float* dnew, int64_t dnew_bs, const float* std, const float* interior_mask, void* dy,
                      int dy_dtype, int y_cs, float* dprev, int64_t dprev_bs, int B, int64_t N, int F,
                      float keep_prev, p4c_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K3  losses -- replaces WeightedLoss.forward (losses.py:130-169) and
 * ScaledLoss.forward (losses.py:186-210).
 *
 * prediction/target: (B,T,N,F) with strides (bs, ts), inner (N,F) dense.
 * mask per p4c_mask_mode.  weights: (F).  interior_mask: (N) floats.
 * denominators: losses.py:156,167 -- num_interior - #(grid points masked for every b,t,f).
 */

/* count of grid points n with mask[b,t,n,f]==0 for all (b,t,f)  -> *count (device int32).
 * For P4C_MASK_FROM_NAN `mask_or_target` is the raw target. */
int p4c_mask_all_zero_count(const void* mask_or_target, int mask_mode, int64_t bs, int64_t ts, int B, int T,
                            int64_t N, int F, int32_t* count, p4c_stream_t stream);

/* workspace size (bytes) needed by the reducing loss kernels below */
size_t p4c_loss_workspace_bytes(int B, int T, int64_t N, int F);

/* WeightedLoss, reduce_spatial_dim=True -> out (B,T):
 *   out[b,t] = sum_n interior[n] * sum_f w[f]*l(pred*m, tgt*m) / (num_interior - *masked_count)
 * masked_count may be NULL (=0). */
int p4c_weighted_loss_fwd(const float* pred, int64_t pred_bs, int64_t pred_ts, const float* target, int64_t tgt_bs,
                          int64_t tgt_ts, const void* mask, int mask_mode, const float* weights,
                          const float* interior_mask, float num_interior, const int32_t* masked_count, int kind,
                          float* out, void* workspace, int B, int T, int64_t N, int F, p4c_stream_t stream);

/* WeightedLoss, reduce_spatial_dim=False -> out_map (B,T,N) (losses.py:150-154; used by plots.py:606). */
int p4c_weighted_loss_map(const float* pred, int64_t pred_bs, int64_t pred_ts, const float* target, int64_t tgt_bs,
                          int64_t tgt_ts, const void* mask, int mask_mode, const float* weights, int kind,
                          float* out_map, int B, int T, int64_t N, int F, p4c_stream_t stream);

/* Backward of p4c_weighted_loss_fwd wrt prediction: gout (B,T) -> dpred (B,T,N,F) with strides. */
int p4c_weighted_loss_bwd(const float* gout, const float* pred, int64_t pred_bs, int64_t pred_ts, const float* target,
                          int64_t tgt_bs, int64_t tgt_ts, const void* mask, int mask_mode, const float* weights,
                          const float* interior_mask, float num_interior, const int32_t* masked_count, int kind,
                          float* dpred, int64_t dpred_bs, int64_t dpred_ts, int B, int T, int64_t N, int F,
                          p4c_stream_t stream);

/* ScaledLoss -> out (B,T,F): mean over interior points per feature, sqrt if MSE, times std[f]. */
int p4c_scaled_loss_fwd(const float* pred, int64_t pred_bs, int64_t pred_ts, const float* target, int64_t tgt_bs,
                        int64_t tgt_ts, const void* mask, int mask_mode, const float* std,
                        const float* interior_mask, float num_interior, const int32_t* masked_count, int kind,
                        float* out, void* workspace, int B, int T, int64_t N, int F, p4c_stream_t stream);

/* Anomaly-correlation sums of MetricACC.update (metrics.py:387-433), the validation metric next to the losses:
 * out (3,B,T,F) = spatial means of (p-c)(t-c)m, ((p-c)m)^2, ((t-c)m)^2 with c = climate_means[f].
 * workspace: 3 * p4c_loss_workspace_bytes(B,T,N,F). */
int p4c_acc_sums(const float* pred, int64_t pred_bs, int64_t pred_ts, const float* target, int64_t tgt_bs, int64_t tgt_ts,
                 const void* mask, int mask_mode, const float* climate_means, float* out, void* workspace, int B, int T,
                 int64_t N, int F, p4c_stream_t stream);

/* NaN-aware moments for the dataset statistics (compute_dataset_stats.py:11-127): out (5,B,F) = per (sample, feature)
 * sum, sum of squares, count of non-NaN values, min, max over `rows` rows of F features (sample b starts at
 * x + b*batch_stride).  With x_next != NULL the value is x_next[i] - x[i] (time-step differences: pass the views
 * in_out[:, 1:] and in_out[:, :-1]).  workspace: 5 * p4c_loss_workspace_bytes(B, 1, rows, F). */
int p4c_nan_moments(const float* x, const float* x_next, int64_t batch_stride, float* out, void* workspace, int B,
                    int64_t rows, int F, p4c_stream_t stream);

/* One AdamW step (torch.optim.AdamW semantics, decoupled weight decay, no amsgrad) over flat fp32 buffers of n elements;
 * `step` is the 1-based step count used for the bias corrections (configure_optimizers, lightning.py:442-467). */
int p4c_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                   double beta2, double eps, double weight_decay, int64_t step, p4c_stream_t stream);

/* Rows next to the path.
 * p4c_unnormalize: out[r,f] = x[r,f]*std[f] + mean[f] as two rounded steps (predict path, lightning.py:1162-1169);
 *   out may alias x.
 * p4c_pack_standardize: planes raw[f*plane_stride + r] -> rows out[r*F + f] = (raw - mean[f]) / std[f]
 *   (Sample.load standardisation datasets/base.py:448-452 + NamedTensor.concat + collate_fn :173-195 in one pass). */
int p4c_unnormalize(const float* x, const float* std, const float* mean, float* out, int64_t rows, int F,
                    p4c_stream_t stream);

/* The same un-normalisation for the output writers (py4cast/io/outputs.py:116-241 take one (lat, lon) plane per time step and
 * feature: `raw_data.tensor[:, :, idx].cpu().numpy()`, a strided gather + a synchronous copy per plane in the reference):
 * x (images, N, F) features-last -> out (images, F, N), out[i, f, n] = x[i, n, f] * std[f] + mean[f] (two rounded steps, bit-exact
 * with lightning.py:1162-1169).  images = B*T.  `out` may be device memory or PINNED HOST memory mapped into the device (the
 * kernel then streams the planes straight over PCIe; see py4cast_amd/outputs.py). */
int p4c_unnormalize_planes(const float* x, const float* std, const float* mean, float* out, int64_t images, int64_t N, int F,
                           p4c_stream_t stream);
int p4c_pack_standardize(const float* raw, int64_t plane_stride, const float* mean, const float* std, float* out,
                         int64_t rows, int F, p4c_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K2+K3 fused (training path): one AR step's state update AND its contribution to the
 * training loss in a single pass over (B,N,F) -- the target of step i is also the border
 * state of step i (lightning.py:567 / 816), so it is read once.
 *   new_state as in p4c_ar_update_fwd (border_state == target);
 *   loss_out[b] = sum_n interior[n] * sum_f w[f]*l(new*m, tgt*m) / denom     (one (b,t) column)
 * loss_out: (B) with element stride loss_stride (so that it can alias out[:, t] of a (B,T) tensor).
 */
int p4c_ar_update_loss_fwd(const float* prev, int64_t prev_bs, const void* y, int y_dtype, int y_cs,
                           const float* target, int64_t tgt_bs, const float* std, const float* mean,
                           const float* border_mask, const float* interior_mask, float* new_state, int64_t new_bs,
                           const float* weights, float num_interior, const int32_t* masked_count, int kind,
                           int mask_mode, float* loss_out, int64_t loss_stride, void* workspace, int B, int64_t N,
                           int F, float keep_prev, p4c_stream_t stream);

/* The same step, additionally emitting the NEXT AR step's network input x_next (B,N,c_pad) in the dtype / padded
 * layout of p4c_build_x with T_in = 1: [new state (F) | statics (Fs) | forcing of the next step (Ff) | zeros], so the
 * state just computed is not read back by a separate p4c_build_x ("feed next step", lightning.py:636-656 + 711-767).
 * x_next has the dtype of y.  Needs mask_mode == P4C_MASK_NONE and one of two paths: the 16-byte path (F % 4 == 0, F <= 64,
 * Fs % 4 == 0, aligned rows: a lane owns 4 features of one grid point) or, for ANY feature count F <= 64 -- the shipped Titan
 * configuration has 21 --, the flat path (round 4: N * F % 4 == 0, rows of y / x_next whole 16-byte slots, c_pad <= 256, 16-byte
 * aligned fp32 rows; the (N, F) arrays are streamed flat, the row tensors pass through LDS tiles of 64 grid points; the same values
 * bit for bit); returns P4C_ERR_UNSUPPORTED otherwise (call p4c_build_x then).  P4C_NO_FLAT_STEP=1 turns the flat path off. */
int p4c_ar_update_loss_fwd_next(const float* prev, int64_t prev_bs, const void* y, int y_dtype, int y_cs,
                                const float* target, int64_t tgt_bs, const float* std, const float* mean,
                                const float* border_mask, const float* interior_mask, float* new_state, int64_t new_bs,
                                const float* weights, float num_interior, const int32_t* masked_count, int kind,
                                int mask_mode, float* loss_out, int64_t loss_stride, void* workspace, int B, int64_t N,
                                int F, float keep_prev, void* x_next, int c_pad, const float* statics, int64_t statics_bs,
                                int Fs, const float* forcing_next, int64_t forcing_bs, int Ff, p4c_stream_t stream);

/* Backward of the fused step.  g_next: gradient wrt new_state arriving from the later AR step
 * (may be NULL for the last step; g_next2 is an optional second addend with channel stride g2_cs,
 * e.g. the first F channels of the model's dx); gloss: (B) with stride, the upstream gradient of
 * loss_out.  Writes dy (B,N,y_cs) and dprev (B,N,F) (dprev may be NULL). */
int p4c_ar_update_loss_bwd(const float* g_next, int64_t g_next_bs, const void* g_next2, int g2_dtype, int g2_cs,
                           const float* gloss, int64_t gloss_stride, const float* new_state, int64_t new_bs,
                           const float* target, int64_t tgt_bs, const float* std, const float* interior_mask,
                           int force_border, const float* weights, float num_interior, const int32_t* masked_count,
                           int kind, int mask_mode, void* dy, int dy_dtype, int y_cs, float* dprev, int64_t dprev_bs,
                           int B, int64_t N, int F, float keep_prev, p4c_stream_t stream);

/* The network's 1x1 output convolution AND the fused AR step in one pass (HalfUNet, bf16 maps): y = relu(a * a_scale + a_shift) wout^T
 * is formed on the matrix cores per 32 grid points, rounded to bf16 exactly where the two-kernel route stores it, and consumed in
 * registers by the state update / border forcing / weighted loss / next-input emission / saved loss gradient of
 * p4c_ar_update_loss_fwd_next_saved (same arguments, same arithmetic, same new state; the loss is summed in another order).  y is
 * never written.  a: (B,N,64) bf16; a_scale / a_shift: (B,64); wout: (cout,64) fp32, F <= cout <= 64; x_next and lgrad may be NULL;
 * no NaN masks.  Two forms, same bits: the FLAT one (any F <= 64 with N * F a multiple of 4, rows of x_next whole 16-byte slots,
 * c_pad <= 256, 16-byte aligned fp32 rows: the (N, F) arrays are streamed flat, 16 bytes per lane, the convolution heads every
 * 64-point tile) wherever those conditions hold -- the shipped Titan configuration's F = 21 included --, otherwise the 16-byte form
 * (F, Fs, c_pad multiples of 4, at most pow2_ge(F / 4) tail quads).  P4C_NO_FLAT_STEP=1 / P4C_TAIL_V4=1 prefer the 16-byte form. */
int p4c_out_conv_update_loss_fwd(const void* a, const float* a_scale, const float* a_shift, const float* wout, int cout,
                                 const float* prev, int64_t prev_bs, const float* target, int64_t tgt_bs, const float* std,
                                 const float* mean, const float* border_mask, const float* interior_mask, float* new_state,
                                 int64_t new_bs, const float* weights, float num_interior, const int32_t* masked_count, int kind,
                                 float* loss_out, int64_t loss_stride, void* workspace, int B, int64_t N, int F, float keep_prev,
                                 void* x_next, int c_pad, const float* statics, int64_t statics_bs, int Fs, const float* forcing_next,
                                 int64_t forcing_bs, int Ff, void* lgrad, int64_t lgrad_bs, p4c_stream_t stream);
/* The fused step that ALSO saves d loss_elem / d pred of every element (2 (pred - target) mask for MSE, sign for L1) as bf16 rows --
 * lgrad: (N, F) per sample, batch stride lgrad_bs elements, 16-byte aligned -- and the backward that reads them instead of the new
 * state and the target: at F = 60 the backward reads 120 instead of 480 bytes per grid point (the forward writes 120 more).  The
 * saved values are rounded to bf16 (2^-9 relative), the precision the network gradient dy is stored in anyway: used by the bf16
 * flavour of the native rollout; the fp32 flavour keeps p4c_ar_update_loss_bwd.  x_next may be NULL (last AR step).
 * 16-byte or flat path (as p4c_ar_update_loss_fwd_next); P4C_ERR_UNSUPPORTED otherwise. */
int p4c_ar_update_loss_fwd_next_saved(const float* prev, int64_t prev_bs, const void* y, int y_dtype, int y_cs,
                                      const float* target, int64_t tgt_bs, const float* std, const float* mean,
                                      const float* border_mask, const float* interior_mask, float* new_state, int64_t new_bs,
                                      const float* weights, float num_interior, const int32_t* masked_count, int kind,
                                      int mask_mode, float* loss_out, int64_t loss_stride, void* workspace, int B, int64_t N,
                                      int F, float keep_prev, void* x_next, int c_pad, const float* statics, int64_t statics_bs,
                                      int Fs, const float* forcing_next, int64_t forcing_bs, int Ff, void* lgrad, int64_t lgrad_bs,
                                      p4c_stream_t stream);
int p4c_ar_update_loss_bwd_saved(const float* g_next, int64_t g_next_bs, const void* g_next2, int g2_dtype, int g2_cs,
                                 const float* gloss, int64_t gloss_stride, const void* lgrad, int64_t lgrad_bs, const float* std,
                                 const float* interior_mask, int force_border, const float* weights, float num_interior,
                                 const int32_t* masked_count, int kind, int mask_mode, void* dy, int dy_dtype, int y_cs,
                                 float* dprev, int64_t dprev_bs, int B, int64_t N, int F, float keep_prev, p4c_stream_t stream);

/* ====================================================================================
 * Model kernels -- the network arithmetic the reference obtains from mfai v5.0.1
 * (py4cast/models.py:10-20; model forward at py4cast/lightning.py:591-596) and, underneath,
 * from cuDNN/cuBLAS.  Activations are (B,H,W,C) with C contiguous ("features last"), C a
 * multiple of 32, fp32 unless a dtype argument says otherwise.
 * ==================================================================================== */

/* Re-order a canonical torch conv weight w[CO][CI][ks][ks] into the MFMA operand stream used by
 * p4c_conv_fwd: out[M_pad/64][ks*ks][K_pad/8][2][64][4] (zero padded).
 * transpose_flip=0: forward (M = CO, K = CI).  transpose_flip=1: data gradient (M = CI, K = CO, taps
 * flipped), so the same conv kernel evaluates dL/dinput.  out needs M_pad*K_pad*ks*ks elements (fp32 for
 * compute = P4C_F32; bf16 in the [..][K_pad/16][2][64][8] order for compute = P4C_BF16). */
int p4c_prep_weights(const float* w, int CO, int CI, int ks, int transpose_flip, int M_pad, int K_pad, void* out,
                     int compute, p4c_stream_t stream);

/* "same" convolution (ks = 1 or 3, stride 1, zero padding) on the fp32 matrix cores
 * (compute = P4C_F32: v_mfma_f32_32x32x2_f32, exact; P4C_BF16: v_mfma_f32_32x32x16_bf16 on operands rounded
 * to bf16 while staged).  `storage` is the element type of in/out (and dout) in HBM: P4C_F32, or P4C_BF16 with
 * compute = P4C_BF16.  out[b,y,x,m] = sum_{tap,k} act(in)[b,y+dy,x+dx,k] * W[m][k][tap] (+ bias[m])
 * with act(v) = relu?(v*in_scale[b,k] + in_shift[b,k]) applied while the input tile is staged
 * (in_scale/in_shift: (B,CI) or NULL).  stat_partial (or NULL): per-tile channel sums,
 * [B*tiles][2][64] floats with tiles = p4c_conv_stat_tiles(compute, storage, CI, B, H, W) -- the BatchNorm/GroupNorm statistics of
 * the output, produced in the epilogue.  in: (B,H,W,CI), CI in {32,64,96}; out: (B,H,W,out_cs),
 * m_blocks*64 channels written. */
int p4c_conv_fwd(const void* in, int compute, int storage, int CI, const void* wprep, int ks, const float* in_scale,
                 const float* in_shift, int in_relu, const float* bias, void* out, int out_cs, float* stat_partial,
                 int B, int H, int W, int m_blocks, p4c_stream_t stream);

/* Weight gradient of the same convolution: grad[CO][CI][ks][ks] += sum_px act(in)[px+tap][ci] * dout[px][co].
 * dout: (B,H,W,64).  workspace: p4c_conv_wgrad_workspace_bytes(CI_pad, ks) bytes (per-workgroup partials,
 * reduced deterministically). */
size_t p4c_conv_wgrad_workspace_bytes(int CI_pad, int ks);
/* rows per sample of p4c_conv_fwd's stat_partial output (pixel tiles, or (workgroup, wave) slots of the
 * persistent bf16 kernel) */
int p4c_conv_stat_tiles(int compute, int storage, int CI, int B, int H, int W);
/* the same for a given kernel size (1x1 and 3x3 convolutions of one shape may run different kernels);
 * p4c_conv_stat_tiles is the ks = 3 value */
int p4c_conv_stat_tiles_ks(int compute, int storage, int CI, int ks, int B, int H, int W);
/* which kernel p4c_conv_fwd launches for a 64-filter convolution of this shape: 2 = row-streaming kernel (csrc/conv_rows.hip),
 * 1 = tile-ring kernel, 0 = generic tiled kernel.  (bench.py names the roofline kernel with it.) */
int p4c_conv_kernel_kind(int compute, int storage, int CI, int ks, int B, int H, int W);
int p4c_conv_wgrad(const void* in, int compute, int storage, int CI_pad, int ks, const float* in_scale, const float* in_shift,
                   int in_relu, const void* dout, int CO, int CI, float* grad, void* workspace, int B, int H, int W,
                   p4c_stream_t stream);
/* The weight gradient of a 3x3 convolution 64 -> 64 on bf16 maps inside a [conv -> norm -> ReLU] block, taken from dA -- the gradient
 * with respect to the block's post-ReLU activation -- instead of dY: pass 2 of the normalisation backward is applied while the operand
 * is staged,  dY = rstd * (gamma * g - k1 - xhat * k2),  g = dA * [y * nscale + nshift > 0],  xhat = (y - mean) * rstd,  with y the
 * block's raw convolution output (B,H,W,64) bf16, gamma (64), nscale / nshift / rstd / mean / k1 / k2 (B,64) fp32 (k1, k2: the sums of
 * g and g * xhat over the statistics' support, divided by its size -- what the normalisation backward's pass 1 leaves).  What the
 * HalfUNet backward plan does for every block (no dY map in memory); exported for tests and reuse.  Other arguments as p4c_conv_wgrad. */
int p4c_conv_wgrad_nb(const void* in, const float* in_scale, const float* in_shift, int in_relu, const void* dA, const void* y,
                      const float* gamma, const float* nscale, const float* nshift, const float* rstd, const float* mean,
                      const float* k1, const float* k2, int CO, int CI, float* grad, void* workspace, int B, int H, int W,
                      p4c_stream_t stream);
/* x pass of the adjoint of HalfUNet's decoder merge (mfai's HalfUNet sums `F.interpolate(level_k, scale_factor=2^k, mode="bilinear")`,
 * k = 1..4, at full resolution; its backward runs under py4cast/lightning.py:591-596 / training_step :806-831), all four levels from
 * ONE read of the (B,H,W,64) gradient dS:  tx_k[b,y,X,:] = sum_x wx_k(x,X) dS[b,y,x,:],  tx_k is (B,H,W/2^k,64), align_corners=False
 * weights (border columns take the clamped taps).  bf16 storage with W % 64 == 0 runs as a banded GEMM on the matrix cores
 * (csrc/upbwd_mfma.hip), otherwise on the vector ALU; W % 16 == 0 required.  The y pass is part of the backward plan. */
int p4c_upsample_sum_bwd_x(int storage, const void* dS, int B, int H, int W, void* tx1, void* tx2, void* tx3, void* tx4,
                           p4c_stream_t stream);

/* The whole backward of a 1x1 convolution from 64 channels behind a [conv -> norm -> ReLU] block (the HalfUNet's output convolution:
 * mfai's `outconv`) in ONE pass over its operands, bf16 maps (csrc/out_conv_bwd.hip):
 *   dA (B,N,64) = the data gradient (wprep_dgrad: p4c_prep_weights(w, CO, 64, ks 1, transpose_flip 1, 64, 64, compute P4C_BF16));
 *   stat_partial (B, p4c_out_conv_bwd_slots, 2, 64) = per-workgroup sums of g and g * xhat, g = dA * [y * scale + shift > 0],
 *     xhat = (y - mean) * rstd -- pass 1 of the block's normalisation backward (y: the block's raw convolution output, (B,N,64) bf16;
 *     scale / shift / mean / rstd (B,64) fp32);
 *   grad_w (CO,64) += sum over pixels of dy[px][co] * relu(y[px][ci] * scale + shift)  (fixed-order reduction of per-workgroup
 *     partials in `workspace`, p4c_out_conv_bwd_workspace_bytes).
 * dy (B,N,64) bf16 with channels >= CO zero.  What the HalfUNet backward plan runs for its last layer; exported for tests and reuse. */
int p4c_out_conv_bwd_slots(int B, int64_t N);
size_t p4c_out_conv_bwd_workspace_bytes(int B, int64_t N);
int p4c_out_conv_bwd(const void* dy, const void* wprep_dgrad, const void* y, const float* scale, const float* shift, const float* mean,
                     const float* rstd, void* dA, float* stat_partial, int CO, float* grad_w, void* workspace, int B, int64_t N,
                     p4c_stream_t stream);
/* which kernel the 3x3 64 -> 64 weight gradient of this shape runs on: 1 = row-streaming kernel (csrc/conv_wgrad_rows.hip), 0 = tile kernel */
int p4c_conv_wgrad_kernel_kind(int storage, int B, int H, int W);
/* The same convolution on bf16 feature maps with FEWER than 64 channels, in place: in (B,H,W,in_c), out (B,H,W,out_c), in_c and
 * out_c multiples of 8 up to 64 (absent channels are staged as zeros / not stored), weights prepared for 64 x 64 (p4c_prep_weights
 * with the real CO / CI).  Plain launches of the row kernel only (no input transform, no statistics): what a features-last
 * `Conv2d(bias=False)` of mfai's SwinUNETR decoder needs forward, for its data gradient (transposed weights) and -- p4c_conv_wgrad_compact,
 * workspace of p4c_conv_wgrad_workspace_bytes(64, ks) -- for its weight gradient.  p4c_conv_compact_supported: 1 if the shape is served. */
int p4c_conv_compact_supported(int in_c, int out_c, int ks, int B, int H, int W);
int p4c_conv_fwd_compact(const void* in, int in_c, const void* wprep, int ks, void* out, int out_c, int B, int H, int W,
                         p4c_stream_t stream);
int p4c_conv_wgrad_compact(const void* in, int in_c, int ks, const void* dout, int dout_c, int CO, int CI, float* grad, void* workspace,
                           int B, int H, int W, p4c_stream_t stream);

/* HalfUNet (the network behind `model_name: HalfUNet`, config/CLI/model/halfunet.yaml): 5 encoder
 * blocks [conv3x3 -> norm -> ReLU] x2 at 64 filters with 2x2 max-pool between, all levels bilinearly
 * up-sampled and summed, one decoder block, 1x1 output conv.  One call enqueues the whole
 * forward (or backward) on the stream. */
typedef struct p4c_halfunet_desc {
    int32_t B, H, W;      /* H, W multiples of 16 */
    int32_t cin;          /* real input channels */
    int32_t cin_pad;      /* x is (B,H,W,cin_pad), cin_pad in {32,64,96}; channels >= cin must be zero */
    int32_t cout;         /* real output channels (<= 64); y is (B,H,W,64), channels >= cout are zero */
    int32_t dx_channels;  /* leading input channels whose gradient is returned (<= 64, 0 = none) */
    int32_t dtype;        /* storage of x, y, dy, dx and of the saved activations: P4C_F32, or P4C_BF16 (needs compute = P4C_BF16) */
    int32_t norm;         /* 0 = BatchNorm2d, 1 = GroupNorm */
    int32_t groups;       /* GroupNorm groups (divides 64) */
    int32_t has_bias;     /* conv bias (settings.bias); only 0 is implemented */
    float eps;
    float momentum;
    int32_t compute;      /* matrix-core input type of the convolutions: P4C_F32 (exact fp32 MFMA) or P4C_BF16
                             (operands rounded to bf16, fp32 accumulate) */
    int32_t weights_prepared; /* 1: p4c_halfunet_prepare_weights() ran on this scratch workspace since the parameters
                                 last changed (e.g. once per rollout); forward/backward then skip their own
                                 re-layout of the weights.  0: each call prepares what it needs (one extra launch). */
    int32_t skip_out_conv;    /* forward only: 1 = stop before the 1x1 output convolution (y may be NULL): the caller runs it fused with
                                 the AR step, p4c_out_conv_update_loss_fwd on the tensors p4c_halfunet_tail names.  Backward is the same. */
} p4c_halfunet_desc;

/* The first convolution of the plan at 65..72 input channels (the benchmark's 69 = 60 state + 5 forcing + 4 static features,
 * py4cast/lightning.py:256-261) runs, in the bf16 flavour, as a 64-channel row launch on channels 0..63 + this tail pass (round 6,
 * csrc/conv_thin.hip): y (B,H,W,64) bf16, in place, += conv3x3 of the channels 64 .. cin-1 of x (B,H,W,x_cs) bf16 with the fp32 master
 * weight w [64][cin][3][3]; stat_partial (or NULL): p4c_first_conv_tail_slots(B,H,W) slots of [2][64] channel sums / sums of squares of
 * the stored y per sample.  W a multiple of 32, x_cs >= 72 a multiple of 8. */
int p4c_first_conv_tail_slots(int B, int H, int W);
int p4c_first_conv_tail(const void* x, int x_cs, int cin, const float* w, void* y, float* stat_partial, int B, int H, int W, p4c_stream_t stream);

/* Side stream of the calling thread for p4c_halfunet_backward (weight gradients run beside the backward chain, ordered by
 * events and joined before the call returns control of `stream`): `side` a hipStream_t, `events` n_events >= 8 hipEvent_t
 * handles created with hipEventDisableTiming, all owned by the caller and alive until replaced.  side = NULL restores the
 * default (library-created on first use).  p4c_side_stream_enable(0) runs everything on `stream`. */
int p4c_set_side_stream(p4c_stream_t side, void* const* events, int n_events);
/* Deferred join (calling thread, library-owned side stream only): with on = 1, p4c_halfunet_backward returns WITHOUT making
 * `stream` wait for the weight gradients it put on the side stream -- the gradient buffer is complete only after
 * p4c_side_stream_join(stream).  For the reverse sweep of a rollout (py4cast/lightning.py:565-662 differentiated): the
 * full-resolution weight gradients left at the end of one AR step's backward run beside the next step's chain instead of alone.
 * The caller keeps x / saved / dy of every deferred call alive (and un-reused) until the join; dy may be overwritten by the next
 * call's producer (the call orders that itself). */
int p4c_side_stream_defer(int on);
/* on = 0: the weight gradients run on the caller's stream like everything else (a host that cannot keep two streams fed; the
 * single-stream HIP-graph capture); on = 1: beside the backward chain again.  Between steps only (nothing in flight on the side
 * stream).  PROCESS-wide (round 6): the setting reaches p4c_halfunet_backward on whichever host thread runs it (autograd's device
 * thread, not the caller's).  p4c_side_stream_launch_count: weight-gradient launches this process has issued to a side stream. */
int p4c_side_stream_enable(int on);
long long p4c_side_stream_launch_count(void);
int p4c_side_stream_join(p4c_stream_t stream);
/* Rewrite a captured, not yet instantiated HIP graph (hipGraph_t): every 1-D memset node becomes a kernel node filling the same bytes
 * with the same dependencies.  On this stack memset nodes replay a wrong byte value from the second launch on, which breaks library
 * kernels that zero their scratch with a memset inside the captured region (torch's multi-block reductions).  *replaced / *left: memset
 * nodes rewritten / left alone (2-D ones). */
int p4c_graph_replace_memsets(void* graph, int* replaced, int* left);

/* number of floats of the flat parameter vector, laid out in this order (canonical torch layouts):
 *   for block in enc1..enc5, decoder: conv1.weight (64,cin_b,3,3), norm1.weight (64), norm1.bias (64),
 *                                     conv2.weight (64,64,3,3),   norm2.weight (64), norm2.bias (64)
 *   outconv.weight (cout,64,1,1)
 * (cin_b = cin for enc1, 64 otherwise).  running: [12 norms][2 (mean,var)][64] floats. */
int64_t p4c_halfunet_param_count(const p4c_halfunet_desc* d);
/* saved_bytes: activations kept from forward for backward (one per forward call still awaiting its
 * backward); scratch_bytes: transient buffers shareable between calls on one stream. */
int p4c_halfunet_workspace_bytes(const p4c_halfunet_desc* d, size_t* saved_bytes, size_t* scratch_bytes);
/* re-lays (and for P4C_BF16 rounds) every convolution weight, forward and data-gradient orientation, into the
 * scratch workspace: one launch.  Valid until the parameters change or the scratch workspace is reused elsewhere. */
int p4c_halfunet_prepare_weights(const p4c_halfunet_desc* d, const float* params, void* scratch, p4c_stream_t stream);
int p4c_halfunet_forward(const p4c_halfunet_desc* d, const void* x, const float* params, float* running, void* y,
                         void* saved, void* scratch, int training, p4c_stream_t stream);
/* The operands of the network's last layer inside the workspaces of a forward call: *a = raw output of the decoder's second block
 * (B,H,W,64) in `saved`, *a_scale / *a_shift = its normalisation (B,64) fp32 in `saved` (ReLU follows), *wout = the 1x1 output
 * convolution's weight (cout,64) fp32 inside `params`. */
int p4c_halfunet_tail(const p4c_halfunet_desc* d, const float* params, void* saved, const void** a, const float** a_scale,
                      const float** a_shift, const float** wout);
/* dy: (B,H,W,64), channels >= cout must be ZERO (the fused backward of the output convolution feeds all 64 to the matrix cores
 * against zero weight rows: NaN / Inf garbage there would reach every upstream gradient; the in-tree rollout zero-fills them); dx: (B,H,W,64) or NULL (first dx_channels channels valid);
 * grads: flat, same layout as params, ACCUMULATED into (+=). */
int p4c_halfunet_backward(const p4c_halfunet_desc* d, const void* x, const float* params, const void* dy, void* dx,
                          float* grads, void* saved, void* scratch, int training, p4c_stream_t stream);


/* ------------------------------------------------------------------------------------
 * Mesh-GNN path (model_name GraphLam / HiLam / HiLamParallel: config/CLI/model/graphlam.yaml:19-26, hilam.yaml,
 * hilamparallel.yaml; graph batches are (B, ngrid, features), py4cast/lightning.py:526-535).  An InteractionNet layer
 * is  m_e = MLP_e([e, x_s[src(e)], x_r[dst(e)]]),  agg_n = sum_{e: dst(e)=n} m_e  (mesh_aggr: sum),
 * x_r += MLP_n([x_r, agg]).  These entry points are its edge gather / scatter passes; rows are C contiguous
 * features, C * sizeof(dtype) a multiple of 16 bytes, indices int32.
 * ------------------------------------------------------------------------------------ */
enum p4c_activation { P4C_ACT_NONE = 0, P4C_ACT_RELU = 1, P4C_ACT_SILU = 2 };

/* out[e] = act(base[e] + a[ia[e]] + b[ib[e]])  for e < E: the first Linear of MLP_e distributed over the concat
 * (base = e W_e + bias, a = x_s W_s, b = x_r W_r are projected once per edge / node), replacing
 * index_select + cat + activation.  base, (a, ia), (b, ib) are each optional (NULL).  All rows have C features. */
int p4c_edge_gather_add_fwd(const void* base, const void* a, const int32_t* ia, const void* b, const int32_t* ib,
                            void* out, int64_t E, int C, int dtype, int act, p4c_stream_t stream);
/* dpre[e] = dh[e] * act'(base[e] + a[ia[e]] + b[ib[e]]) (the pre-activation is recomputed, never stored).  dpre is the
 * gradient of base; the gradients of a and b are p4c_segment_sum(dpre) over the CSR of ia and ib. */
int p4c_edge_gather_add_bwd(const void* dh, const void* base, const void* a, const int32_t* ia, const void* b,
                            const int32_t* ib, void* dpre, int64_t E, int C, int dtype, int act, p4c_stream_t stream);
/* out[n] = (init ? init[n] : 0) + sum_{j in [offsets[n], offsets[n+1])} msg[perm ? perm[j] : j]   for n < N:
 * receiver-sorted CSR (offsets: N+1 entries; perm: E edge ids grouped by receiver, NULL if msg is already grouped),
 * replacing index_add_ -- no atomics, fixed summation order (bitwise reproducible).  Also the adjoint of a row
 * gather.  msg: (E,C) of `dtype`; init/out: (N,C) of `out_dtype` (= dtype, or P4C_F32 for bf16 messages). */
int p4c_segment_sum(const void* msg, const int32_t* offsets, const int32_t* perm, const void* init, void* out,
                    int64_t N, int64_t E, int C, int dtype, int out_dtype, p4c_stream_t stream);
/* Two segment sums over the same E rows in ONE launch -- out_a over (offsets_a, perm_a), out_b over (offsets_b, perm_b); both
 * (N*,C) of `dtype`, no init: the two adjoints of an edge MLP's gathers (gradient rows summed by sender and by receiver, as
 * autograd's two index_add_ calls do).  Each side is computed exactly as its own p4c_segment_sum launch would (same order). */
int p4c_segment_sum_pair(const void* msg, const int32_t* offsets_a, const int32_t* perm_a, void* out_a, int64_t Na,
                         const int32_t* offsets_b, const int32_t* perm_b, void* out_b, int64_t Nb, int64_t E, int C, int dtype,
                         p4c_stream_t stream);


/* ------------------------------------------------------------------------------------
 * Swin path (model_name SwinUNetR: config/CLI/model/swinunetr.yaml:19-30 -- depths [2,2,2,2], num_heads [3,6,12,24],
 * feature_size 24 => head_dim 8; grid batches (B, lat, lon, features), py4cast/lightning.py:591-596).
 * Windowed multi-head self-attention of one Swin block on the token grid, shift / window partition / head split and
 * their inverses folded into the addressing:
 *   out[b,y,x,head,:] = sum_k softmax_k(q.k * scale + bias[head][q][k] + shift_mask[q][k]) v_k
 * over the ws x ws window of the grid rolled by (-shift, -shift) that contains (y, x); shift_mask = -100 between tokens
 * of different wrap-around regions (Swin's attn_mask), 0 otherwise.
 *   qkv   : (B, Hp, Wp, 3, heads, head_dim)   -- the qkv Linear's output, `dtype` storage; Hp, Wp multiples of ws
 *   bias_t: (heads, N, N) fp32 indexed [head][key][query], N = ws*ws (the relative-position bias, TRANSPOSED); NULL = none
 *   out   : (B, Hp, Wp, heads*head_dim)
 * head_dim in {8,16,32}, ws in 3..8.  dtype = P4C_BF16: products on the bf16 matrix cores (operands and P rounded to bf16), softmax
 * in fp32.  dtype = P4C_F32: the fp32-exact flavour -- every product an fp32 FMA chain (<= 2e-6 of a float64 evaluation).
 * ------------------------------------------------------------------------------------ */
int p4c_window_attn_fwd(const void* qkv, const float* bias_t, void* out, int B, int Hp, int Wp, int heads, int head_dim,
                        int ws, int shift, float scale, int dtype, p4c_stream_t stream);
/* dqkv: gradient of qkv (same layout; fully written).  dbias_t (or NULL): gradient of bias_t, (heads,N,N) [key][query],
 * summed over batch and windows in a fixed order; needs `workspace` of p4c_window_attn_bwd_workspace_bytes() bytes.
 * The attention matrix is recomputed from qkv, nothing is saved by the forward. */
size_t p4c_window_attn_bwd_workspace_bytes(int B, int Hp, int Wp, int heads, int ws);
int p4c_window_attn_bwd(const void* qkv, const float* bias_t, const void* dout, void* dqkv, float* dbias_t, void* workspace,
                        int B, int Hp, int Wp, int heads, int head_dim, int ws, int shift, float scale, int dtype,
                        p4c_stream_t stream);


/* ------------------------------------------------------------------------------------
 * Row-wise passes of the GNN's MLPs (every MLP of GraphLam / HiLam is Linear - SiLU - Linear - LayerNorm on rows of
 * hidden_dims = 64 features, config/CLI/model/graphlam.yaml:21-22): streams of R rows (0.5 M grid nodes, 1-2 M edges per
 * sample) of C contiguous features, C * sizeof(dtype) a multiple of 16 bytes and at most 1 KiB.
 * ------------------------------------------------------------------------------------ */
/* out[r] = LayerNorm(x[r]) * gamma + beta (+ res[r])   -- torch.nn.LayerNorm(C) semantics (biased variance, eps inside the
 * square root), statistics in fp32; res may be NULL. */
int p4c_row_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta, float eps, void* out,
                          int64_t R, int C, int dtype, p4c_stream_t stream);
/* dx, dgamma (C), dbeta (C; must be dgamma + C: one (2,C) buffer) from dy and the forward INPUT x (row statistics are
 * recomputed, nothing is saved by the forward); the gradient of res is dy itself.  Parameter gradients are reduced in a
 * fixed order (no atomics); workspace: p4c_row_layernorm_bwd_workspace_bytes(R, C, dtype) bytes. */
size_t p4c_row_layernorm_bwd_workspace_bytes(int64_t R, int C, int dtype);
int p4c_row_layernorm_bwd(const void* dy, const void* x, const float* gamma, float eps, void* dx, float* dgamma, float* dbeta,
                          void* workspace, int64_t R, int C, int dtype, p4c_stream_t stream);
/* The same two passes on rows that are the tokens of (B, Hp, Wp) maps of which only [0,H) x [0,W) are real -- Swin pads every map to
 * a multiple of its window (MONAI SwinTransformerBlock.forward: `F.pad(norm1(x), ...)` ... `x[:, :h, :w, :]`; the class reaches py4cast
 * through mfai, py4cast/models.py:10-20, and runs under py4cast/lightning.py:591-596).  A stage kept in the PADDED layout needs
 * `F.pad(norm1(x))` only: out rows of padding tokens are ZERO whatever x holds there; their dy is ignored (dx = 0, no contribution to
 * dgamma / dbeta).  R must be a multiple of Hp * Wp; Hp = Wp = 0 (or H = Hp, W = Wp): no mask. */
int p4c_row_layernorm_fwd_masked(const void* x, const void* res, const float* gamma, const float* beta, float eps, void* out, int64_t R,
                                 int C, int dtype, int Hp, int Wp, int H, int W, p4c_stream_t stream);
int p4c_row_layernorm_bwd_masked(const void* dy, const void* x, const float* gamma, float eps, void* dx, float* dgamma, float* dbeta,
                                 void* workspace, int64_t R, int C, int dtype, int Hp, int Wp, int H, int W, p4c_stream_t stream);
/* (x + add) -> LayerNorm on rows of up to 2 KiB (UNETR++'s token rows with their positional embedding, 128 ... 1024 features):
 * t[r] = x[r] + add[r % add_rows] (add may be NULL: t = x), sum_out[r] = t[r] (optional), out[r] = LayerNorm(t[r]) * gamma + beta.
 * Backward: dt = LN_backward(dy; t) + extra (extra, optional: the gradient reaching t from its other consumers), dgamma / dbeta as
 * p4c_row_layernorm_bwd; workspace p4c_row_add_layernorm_bwd_workspace_bytes(R, C). */
int p4c_row_add_layernorm_fwd(const void* x, const void* add, int64_t add_rows, const float* gamma, const float* beta, float eps,
                              void* sum_out, void* out, int64_t R, int C, int dtype, p4c_stream_t stream);
size_t p4c_row_add_layernorm_bwd_workspace_bytes(int64_t R, int C);
int p4c_row_add_layernorm_bwd(const void* dy, const void* t, const void* extra, const float* gamma, float eps, void* dt, float* dgamma,
                              float* dbeta, void* workspace, int64_t R, int C, int dtype, p4c_stream_t stream);
/* out (n, fp32) (+)= sum over the nb slices of x (nb x n, fp32 or bf16), n a multiple of 4: the gradient of the `add` table of
 * p4c_row_add_layernorm_fwd (broadcast over the samples), written -- with accumulate -- straight into the parameter's gradient. */
int p4c_sum_leading(const void* x, int dtype, int nb, int64_t n, float* out, int accumulate, p4c_stream_t stream);
/* Weight and bias gradient of y = x W^T + b over R >> O rows:  dw_db[0 .. O*K) = dW[o][k] = sum_r dy[r][o] x[r][k],
 * dw_db[O*K .. O*K+O) = db[o] = sum_r dy[r][o]  (fp32, overwritten).  O = 64, K a multiple of 16 up to 128, bf16 rows
 * (bf16 matrix cores, fp32 accumulation, fixed reduction order).  workspace: p4c_row_linear_wgrad_workspace_bytes(R, K). */
size_t p4c_row_linear_wgrad_workspace_bytes(int64_t R, int K);
int p4c_row_linear_wgrad(const void* dy, const void* x, float* dw_db, void* workspace, int64_t R, int O, int K, int dtype,
                         p4c_stream_t stream);


/* Linear layers on token rows (SwinUNetR's qkv / proj / MLP projections, patch merging; py4cast/models.py:10-20 -> mfai SwinUNETR):
 *   y[r][n] = sum_k x[r][k] M[n][k] (+ bias[n]),  M[n][k] = transposed ? w[k * ldw + n] : w[n * ldw + k]
 * x (R, K) and y (R, N) bf16 rows with row strides ldx / ldy (elements; views of wider tensors are fine), w the fp32 master weight
 * (laid out as the bf16 matrix-core operand inside the kernel: no cast launch), bias fp32 or NULL.  Forward: M = W.  Data gradient:
 * x = dy, w = W, transposed = 1, K and N swapped.  K a multiple of 8 up to 512, N a multiple of 4, operand image within LDS:
 * p4c_row_gemm_supported(K, N) says whether a shape is served (callers use the library GEMM otherwise). */
int p4c_row_gemm_supported(int K, int N);
/* Fused epilogue (round 5), in this order: act = 1: the bf16-rounded pre-activation goes to aux (row stride ldaux) and GELU (erf form) of
 * it on; act = 2: times GELU'(aux) (the data gradient through a GELU); then + res[r][n] (bf16 rows, stride ldres; res may be y itself:
 * accumulate).  act = 0, res = NULL: the plain product. */
int p4c_row_gemm(const void* x, int64_t ldx, const float* w, int ldw, int transposed, const float* bias, void* y, int64_t ldy,
                 int64_t R, int K, int N, const void* res, int64_t ldres, int act, void* aux, int64_t ldaux, p4c_stream_t stream);
/* Weight and bias gradient of the same layer over R >> N rows:  out is (64 * ceil(N / 64)) x KP floats (overwritten), KP = 32 *
 * ceil((K + with_bias) / 32):  out[n][k] = dW[n][k] = sum_r dy[r][n] x[r][k] for k < K, and with_bias: out[n][K] = db[n] = sum_r
 * dy[r][n].  bf16 rows, fp32 accumulation, fixed reduction order (bit-identical reruns).  N, K multiples of 8, K + with_bias <= 224,
 * N <= 512: p4c_row_gemm_wgrad_supported.  workspace: p4c_row_gemm_wgrad_workspace_bytes(R, N, K, with_bias) bytes. */
int p4c_row_gemm_wgrad_supported(int N, int K, int with_bias);
size_t p4c_row_gemm_wgrad_workspace_bytes(int64_t R, int N, int K, int with_bias);
int p4c_row_gemm_wgrad(const void* dy, int64_t ldy, const void* x, int64_t ldx, float* out, void* workspace, int64_t R, int N, int K,
                       int with_bias, p4c_stream_t stream);


/* Fused row MLP of the GNN models (make_mlp: Linear - SiLU - Linear - LayerNorm, hidden = out = 64 features):
 *   pre  = x W1^T + b1 (+ gather_a[index_a[r]] + gather_b[index_b[r]])      -- the gathered addends are the sender / receiver
 *                                                                              parts of a distributed edge-MLP first layer
 *   y    = LayerNorm(SiLU(pre) W2^T + b2) * gamma + beta                     (LayerNorm skipped when gamma is NULL)
 *   out[r] = y,  out_res[r] = y + res[r]                                     (each optional)
 * one pass over the rows each way: a row is read once and written once; nothing but x is kept for the backward (the forward is
 * recomputed).  bf16 rows, fp32 parameters; bf16 matrix cores with fp32 accumulation, SiLU / LayerNorm in fp32. */
typedef struct p4c_row_mlp_desc {
    int64_t rows;
    const void* x;           /* (rows, k) bf16, k a multiple of 16 up to 80 (features beyond k_real zero-padded) */
    int32_t k, k_real;
    const float* w1;         /* [64][k_real] with row stride ldw1 (a column slice of a wider weight is fine) */
    int32_t ldw1;
    const float* b1;         /* [64] or NULL */
    const float* w2;         /* [o_real][64] contiguous */
    const float* b2;         /* [o_real] or NULL */
    int32_t o_real;          /* <= 64; output features beyond are written as zeros (no LayerNorm in that case) */
    const float* gamma;      /* [64] or NULL */
    const float* beta;
    float eps;
    const void* gather_a;    /* (n_a, 64) bf16 or NULL */
    const int32_t* index_a;  /* (rows), or NULL with gather_a set: row-aligned addend gather_a[r] */
    const void* gather_b;
    const int32_t* index_b;
    const void* res;         /* (rows, 64) bf16 or NULL */
    void* out;               /* (rows, 64) bf16 or NULL */
    void* out_res;           /* (rows, 64) bf16 or NULL (needs res) */
    const void* prepared;    /* parameters re-laid by p4c_row_mlp_prepare (valid while they are unchanged), or NULL: every launch
                                then re-lays them itself */
    /* backward only */
    const void* dy;          /* gradient of out, or NULL */
    const void* dy_res;      /* gradient of out_res, or NULL (the gradient of res is dy_res itself) */
    void* dx;                /* (rows, k) bf16 or NULL */
    void* dpre;              /* (rows, 64) bf16 or NULL: gradient of the pre-activation = gradient of the gathered rows before
                                their p4c_segment_sum over index_a / index_b */
    int32_t dx_plus_dy_res;  /* backward, k = 64, res IS x (an edge update e <- e + MLP(e, ..)): dx receives dx + dy_res, the whole
                                gradient of that one tensor, summed in fp32 before the rounding (round 6) */
} p4c_row_mlp_desc;
/* Re-lays w1, b1, w2, b2, gamma, beta (as described by d) into the operand images both kernels use: `prepared` needs
 * p4c_row_mlp_prepared_bytes(k) bytes and stays valid until a parameter changes. */
size_t p4c_row_mlp_prepared_bytes(int k);
int p4c_row_mlp_prepare(const p4c_row_mlp_desc* d, void* prepared, p4c_stream_t stream);
int p4c_row_mlp_fwd(const p4c_row_mlp_desc* d, p4c_stream_t stream);
/* grads (fp32, overwritten): dW1 [64][k] | dW2 [64][64] | db1 [64] | db2 [64] | dgamma [64] | dbeta [64], reduced in a fixed
 * order.  workspace: p4c_row_mlp_bwd_workspace_bytes(rows, k) bytes. */
size_t p4c_row_mlp_bwd_workspace_bytes(int64_t rows, int k);
int p4c_row_mlp_bwd(const p4c_row_mlp_desc* d, float* grads, void* workspace, p4c_stream_t stream);
/* The same backward with the parameter gradients ADDED (+=) straight into the parameters' gradient buffers instead of returned:
 * replaces six `param.grad += g` launches per call (an MLP is applied once per AR step: torch's AccumulateGrad adds them one by
 * one).  dw1: [64][k_real] with row stride ld_dw1 (a column slice of a wider gradient is fine); dw2: [o_real][64]; NULL = dropped. */
typedef struct p4c_row_mlp_grad_sinks {
    float* dw1;
    int32_t ld_dw1;
    float* db1;
    float* dw2;
    float* db2;
    float* dgamma;
    float* dbeta;
} p4c_row_mlp_grad_sinks;
int p4c_row_mlp_bwd_accumulate(const p4c_row_mlp_desc* d, const p4c_row_mlp_grad_sinks* sinks, void* workspace, p4c_stream_t stream);

/* Mesh-GNN launch grouping (round 6; csrc/nodeproj.hip).  An InteractionNet of GraphLAM / HiLAM (config/CLI/model/graphlam.yaml:19-26,
 * hilam.yaml, hilamparallel.yaml; classes taken at py4cast/models.py:66-89) multiplies a node tensor x (R, 64) bf16 with up to three
 * 64 x 64 blocks of wider fp32 Linear weights (the sender / receiver parts of the distributed edge-MLP first layer and the receiver part
 * of the node-update MLP's): W_i[o][k] = w[i][o * ldw[i] + k].  One launch per direction for all n <= 3 blocks:
 *   fwd:   y[i] = x W_i^T                                  (R, 64) bf16 each
 *   dgrad: dx   = sum_i dy[i] W_i (+ acc)                  acc (R, 64) bf16 or NULL; may be dx itself
 *   wgrad: dw[i][o * ld_dw[i] + k] += sum_r dy[i][r][o] x[r][k]   fp32, through the reduction queue below (dw[i] NULL: dropped);
 *          workspace: p4c_node_proj_wgrad_workspace_bytes(R, n) bytes, owned by the call until its reduction has been enqueued
 * bf16 matrix cores with fp32 accumulation, fixed summation order (bit-identical reruns). */
int p4c_node_proj_fwd(const void* x, int64_t R, int n, const float* const* w, const int32_t* ldw, void* const* y, p4c_stream_t stream);
int p4c_node_proj_dgrad(const void* const* dy, int64_t R, int n, const float* const* w, const int32_t* ldw, void* dx, const void* acc,
                        p4c_stream_t stream);
size_t p4c_node_proj_wgrad_workspace_bytes(int64_t R, int n);
int p4c_node_proj_wgrad(const void* const* dy, const void* x, int64_t R, int n, float* const* dw, const int32_t* ld_dw, void* workspace,
                        p4c_stream_t stream);
/* Reduction queue of the parameter-gradient partials that p4c_row_mlp_bwd_accumulate and p4c_node_proj_wgrad leave.  By default each
 * call enqueues its own reduction launch.  After p4c_grad_reduce_defer(1) the reductions are queued instead (process-wide; the
 * workspaces must stay alive) and p4c_grad_reduce_flush(stream) -- the stream of the backward -- reduces all queued jobs, 32 per launch,
 * in submission order: ~550 dependent 5 us launches of a HiLAM step become ~20, and the accumulated gradients are bit-identical.
 * p4c_grad_reduce_defer returns the previous setting (on = -1: drop the queued jobs -- a backward pass that died before its flush -- and
 * reduce at once again); p4c_grad_reduce_pending the number of queued jobs. */
int p4c_grad_reduce_defer(int on);
int p4c_grad_reduce_pending(void);
int p4c_grad_reduce_flush(p4c_stream_t stream);

/* ====================================================================================
 * Tall-skinny products of the efficient paired attention (EPA) of UNETR++ (config/CLI/model/unetrpp.yaml:19-35; the class
 * comes from mfai v5.0.1, py4cast/models.py:10-20).  Per group g = (sample b, head h) a token matrix X[g] is N x d with
 * element (n, i) at x + b*x_bs + h*x_hs + n*x_rs + i (strides in elements): operands are addressed in place inside the
 * qkvv projection's (B, N, 4, heads, d) output or a (B, N, C) token tensor, outputs likewise.  d, e <= 64.
 * ==================================================================================== */
/* gram: partial[g][split] (d x e, fp32) = sum over the split's tokens of X[g][n,:]^T Y[g][n,:]; the caller sums the
 * p4c_ts_gram_splits(N) partials (fixed order).  A shared operand (the E / F projection weights) has bs = hs = 0.
 * d, e multiples of 4. */
int p4c_ts_gram_splits(int64_t N);
/* bf16 x bf16 with d, e multiples of 8 and 16-byte aligned rows runs on the matrix cores (both operands transposed LDS reads of
 * [token][column] tiles, 64 x 64 result blocks, any width); p4c_ts_gram_wide_ok says whether a (dtypes, d, e) is served that way. */
int p4c_ts_gram_wide_ok(int x_dtype, int y_dtype, int d, int e);
int p4c_ts_gram(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const void* y, int y_dtype, int64_t y_bs,
                int64_t y_hs, int64_t y_rs, float* partial, int B, int heads, int64_t N, int d, int e, p4c_stream_t stream);
/* apply: out[g] (N x e) = X[g] (N x d) M[g] (d x e, fp32, group stride m_gs elements; 0 = one matrix for all groups)
 * (+ out when accumulate).  The adjoint of gram: the backward of either is the other. */
/* bf16 token matrices with d, e multiples of 8, d <= 256 (any e) and strides / bases that are multiples of 8 elements / 16 bytes run on
 * the matrix cores (O^T = M^T X^T: a lane's operand is ONE 16-byte load of its token's row, M^T of the heads in LDS) and are not
 * limited to 64 columns; p4c_ts_apply_wide_ok says whether a (dtype, d, e) is served that way (P4C_TS_NO_MFMA=1 turns it off). */
int p4c_ts_apply_wide_ok(int x_dtype, int out_dtype, int d, int e);
int p4c_ts_apply(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m, int64_t m_gs, void* out,
                 int out_dtype, int64_t o_bs, int64_t o_hs, int64_t o_rs, int B, int heads, int64_t N, int d, int e, int accumulate,
                 p4c_stream_t stream);
/* The same product with M given TRANSPOSED in memory (m_t: per group an (e x d) row-major matrix, M[k][c] = m_t[c * d + k]) -- the
 * adjoint applies of an EPA block multiply with At^T, Mq^T, dG^T, VP^T; matrix-core kernel only: p4c_ts_apply_mt_ok says whether the
 * operands meet its alignment / width conditions (otherwise transpose and call p4c_ts_apply). */
int p4c_ts_apply_mt_ok(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m, int64_t m_gs, const void* out,
                       int out_dtype, int64_t o_bs, int64_t o_hs, int64_t o_rs, int d, int e);
int p4c_ts_apply_mt(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m_t, int64_t m_gs, void* out,
                    int out_dtype, int64_t o_bs, int64_t o_hs, int64_t o_rs, int B, int heads, int64_t N, int d, int e, int accumulate,
                    p4c_stream_t stream);
/* apply with a fused epilogue over each token's output row (bf16 in / out, matrix-core form: d, e multiples of 8, d <= 256, e <= 64):
 *   epi = 1: out = softmax_row(X M)                                        -- S = softmax(q Mq) without the logits reaching memory
 *   epi = 2: out = S * (X M - rowsum(X M * S)), S (B, heads, N, e) bf16     -- the softmax backward of the products X M = dS
 * (the spatial branch of EPA, py4cast_amd/ops_ts.py::epa_spatial: no N x p softmax passes either way). */
int p4c_ts_apply_softmax(const void* x, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m, int64_t m_gs, void* out, int64_t o_bs,
                         int64_t o_hs, int64_t o_rs, int B, int heads, int64_t N, int d, int e, int epi, const void* s, int64_t s_bs,
                         int64_t s_hs, int64_t s_rs, p4c_stream_t stream);
/* Sums of the gram partials without the tensor library (round 6; replaces `part.sum(dim=1)` + the strided copies / cast / bias addition
 * that followed it in py4cast_amd/ops_ts.py): part is (A, S, R, E) fp32 dense; out = sum over s of part[a, s, r, :] (+ bias[column %
 * bias_len] when bias is given) (+ out when accumulate), the E columns cut into nseg <= 3 consecutive segments of seg_len[i] columns,
 * segment i stored dense as (A, R, seg_len[i]) at outs[i] (G | nq2 | nk2 of a p4c_ts_gram_norms row).  Lengths multiples of 4. */
int p4c_ts_reduce_splits(const float* part, int A, int S, int R, int E, int nseg, const int* seg_len, float* const* outs, const float* bias,
                         int bias_len, int accumulate, p4c_stream_t stream);
/* out (E x R, dense) (+)= sum over s of part[s] (R x E)^T, E <= 64 and a multiple of 4: the weight gradient of EPA's token-axis Linear (mfai's `E` / `F`,
 * weight (p, N)) from the per-(k | v_sa, sample) token-major products of p4c_ts_apply. */
int p4c_ts_reduce_transpose(const float* part, int S, int64_t R, int E, float* out, int accumulate, p4c_stream_t stream);
/* The x_SA merge of the published EPA code, `x_SA.permute(0, 3, 1, 2).reshape(B, N, C)` on a (B, heads, N, d) tensor (mfai v5.0.1
 * UNetRPP, EPA.forward), for bf16: x = the token-major (B, N, heads, d) memory the apply kernels write, out = (B, d, heads, N) memory
 * (read as (B, N, C) by the output projection); inverse = 1: the adjoint, out = token-major gradient from a (B, d, heads, N) one.
 * A tiled transpose through LDS.  N and C = heads * d multiples of 8, buffers dense and 16-byte aligned. */
int p4c_ts_merge_published(const void* x, void* out, int B, int64_t N, int heads, int d, int inverse, p4c_stream_t stream);
/* Column sums of njobs <= 4 small dense fp32 matrices (rows[i] x cols[i], cols <= 64) in one launch: out[i][j] = sum over the rows.
 * The tails of an EPA backward (bias gradient of the token-axis Linear, the two temperature gradients). */
int p4c_ts_colsums(int njobs, const float* const* in, const int64_t* rows, const int* cols, float* const* out, p4c_stream_t stream);
/* The small matrices of one EPA block in one launch each way (all fp32, contiguous): G = q^T k, Gq = q^T q, Gk = k^T k (B, heads, d, d)
 * from p4c_ts_gram, KP (B, heads, d, p), temperatures t1 / t2 (heads):
 *   nq_i = max(sqrt(max(Gq_ii, 0)), 1e-12), nk_j likewise;  A = softmax_j(t1 G_ij / (nq_i nk_j));  Mq_ic = t2 KP_ic / nq_i.
 * Forward writes At = A^T, Mq, nq, nk (B, heads, d).  Backward: dG, dGq, dGk (zero off the diagonal), dKP and per-(b, head) partials of
 * the temperature gradients, (B, heads) each, which the caller sums over b.  d <= 64; forward: p <= 64. */
int p4c_epa_small_fwd(const float* G, const float* Gq, const float* Gk, const float* KP, const float* t1, const float* t2, float* At, float* Mq,
                      float* nq, float* nk, int B, int heads, int d, int p, int diag_only, p4c_stream_t stream);
int p4c_epa_small_bwd(const float* G, const float* Gq, const float* Gk, const float* KP, const float* t1, const float* t2, const float* At,
                      const float* nq, const float* nk, const float* dAt, const float* dMq, float* dG, float* dGq, float* dGk, float* dKP,
                      float* dt1_part, float* dt2_part, int B, int heads, int d, int p, int diag_only, p4c_stream_t stream);
/* diag_only = 1: Gq / Gk (and dGq / dGk) are the DIAGONALS (B, heads, d) -- the column sums of squares of q and k that
 * p4c_ts_gram_norms produces next to q^T k: partial (B, splits, heads, d*e + d + e) = [X^T Y | sum_n X_ni^2 | sum_n Y_nj^2] (bf16 token
 * matrices; the caller sums the splits, fixed order). */
int p4c_ts_gram_norms(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const void* y, int y_dtype, int64_t y_bs,
                      int64_t y_hs, int64_t y_rs, float* partial, int B, int heads, int64_t N, int d, int e, p4c_stream_t stream);

/* ====================================================================================
 * Ghost module's cheap operation (HalfUNet with use_ghost, config/CLI/model/halfunet.yaml:22): depthwise 3x3 convolution
 * (zero padding, no bias) of the 32 primary channels, concatenated behind them.  Tensors are (B,H,W,64) features-last:
 * channels 0..31 primary, 32..63 depthwise.  w: (32, 9) = torch's (32,1,3,3) depthwise weight.
 * ==================================================================================== */
/* out[..., :32] = in[..., :32]; out[..., 32 + c] = sum_tap w[c][tap] * in[p + tap, c]  (in's upper half is ignored) */
int p4c_ghost_dw_fwd(const void* in, const float* w, void* out, int dtype, int B, int H, int W, p4c_stream_t stream);
/* din[..., c] = dout[..., c] + sum_tap w[c][tap] * dout[p - tap, 32 + c]; din[..., 32:] = 0 */
int p4c_ghost_dw_bwd_data(const void* dout, const float* w, void* din, int dtype, int B, int H, int W, p4c_stream_t stream);
/* partial[blk][c][tap] = partial sums of in[p + tap, c] * dout[p, 32 + c]; blk < p4c_ghost_dw_wgrad_blocks(B,H,W); the caller sums */
int p4c_ghost_dw_wgrad_blocks(int B, int H, int W);
int p4c_ghost_dw_wgrad(const void* in, const void* dout, float* partial, int dtype, int B, int H, int W, p4c_stream_t stream);

/* ====================================================================================
 * InstanceNorm2d(affine) + LeakyReLU (+ residual) on features-last tensors (B, N = H*W, C) -- MONAI's UnetResBlock norm
 * (config/CLI/model/swinunetr.yaml:23, unetrpp.yaml:24 `norm_name: instance`).  C % 4 == 0, C <= 1024.
 * ==================================================================================== */
int p4c_inorm_blocks(int64_t N, int C);
/* partial[b][blk][2][C], blk < p4c_inorm_blocks(N, C).  dy == NULL: sums of x and x^2 (forward statistics).
 * dy != NULL: sums of dz = dy * (y > 0 ? 1 : slope) and of dz * xhat, xhat = (x - mean[b,c]) * rstd[b,c] (backward). */
int p4c_inorm_reduce(const void* x, const void* dy, const void* y, const float* mean, const float* rstd, float slope, float* partial,
                     int dtype, int B, int64_t N, int C, p4c_stream_t stream);
/* dy == NULL: out = lrelu(x * scale[b,c] + shift[b,c] (+ res)).
 * dy != NULL: out = dx = scale * (dz - m1[b,c] - xhat * m2[b,c]); dres (optional) = dz.
 * dy != NULL and rstd == NULL: out = dx = scale[b,c] * dz - m1[b,c] - (x - mean[b,c]) * m2[b,c]  (the caller's coefficients as they
 * are: GroupNorm, whose statistics are shared by the channels of a group -- ops_inorm.group_norm). */
int p4c_inorm_apply(const void* x, const void* res, const void* dy, const void* y, const float* scale, const float* shift,
                    const float* mean, const float* rstd, const float* m1, const float* m2, float slope, void* out, void* dres, int dtype,
                    int B, int64_t N, int C, p4c_stream_t stream);
/* (dy != NULL, res != NULL: res is a gradient that reached the residual operand by another path; dres = dz + res.)
 * The same two passes with a MULTIPLIER behind the activation (round 6): out = lrelu((x * scale + shift (+ res)) * m), m =
 * mul[(row / mul_rows)][c] * mul_factor >= 0 (fp32 table, C per row group): the channel dropout in front of UNETR++'s conv8 (mfai's
 * `nn.Sequential(nn.Dropout2d(0.1), nn.Conv2d(...))`: mul = the Bernoulli draw per (sample, channel), mul_rows = H * W, mul_factor =
 * 1 / (1 - p)) applied by the batch norm + LeakyReLU pass that produces the convolution's input; backward: dz = dy * lrelu' * m in the
 * sums (p4c_inorm_reduce_mul, dy != NULL) and in the apply pass.  mul_rows must divide N. */
int p4c_inorm_reduce_mul(const void* x, const void* dy, const void* y, const float* mean, const float* rstd, float slope, float* partial,
                         int dtype, int B, int64_t N, int C, const float* mul, int64_t mul_rows, float mul_factor, p4c_stream_t stream);
int p4c_inorm_apply_mul(const void* x, const void* res, const void* dy, const void* y, const float* scale, const float* shift,
                        const float* mean, const float* rstd, const float* m1, const float* m2, float slope, void* out, void* dres, int dtype,
                        int B, int64_t N, int C, const float* mul, int64_t mul_rows, float mul_factor, p4c_stream_t stream);

/* p4c_inorm_reduce with the finalize INSIDE the launch (round 6): the workgroup that draws the last ticket turns the partial sums of all
 * samples into the statistics, exactly as p4c_inorm_finalize_fwd / _bwd would (same summation order: bit-identical) -- one dependent
 * 5-8 us launch less per normalisation each way.  Instance form only (one channel per statistics group).  ticket: one zeroed uint32 in
 * device memory per launch in flight (the last workgroup resets it).
 *   fwd: mean, rstd, scale = rstd gamma, shift = beta - mean scale     (B, C) each
 *   bwd: c1 = sum dz / N, c2 = sum dz xhat / N (B, C); dgamma, dbeta (C), written (sums over the samples in order) */
int p4c_inorm_reduce_finalize_fwd(const void* x, float* partial, unsigned int* ticket, const float* gamma, const float* beta, float eps,
                                  float* mean, float* rstd, float* scale, float* shift, int dtype, int B, int64_t N, int C, p4c_stream_t stream);
int p4c_inorm_reduce_finalize_bwd(const void* x, const void* dy, const void* y, const float* mean, const float* rstd, float slope,
                                  float* partial, unsigned int* ticket, float* c1, float* c2, float* dgamma, float* dbeta, int dtype, int B,
                                  int64_t N, int C, p4c_stream_t stream);
/* The statistics of one normalisation from its partial sums, one launch each way (instead of a dozen tiny torch launches):
 *   forward : mean, rstd, scale = rstd * gamma, shift = beta - mean * scale  -- all (B, C) fp32; statistics per channel (groups = 0:
 *             instance norm) or per group of C / groups channels (GroupNorm; at most 64 channels per group);
 *   backward: groups = 0: c1 = S0 / N, c2 = S1 / N (the m1 / m2 of p4c_inorm_apply);  groups > 0: the GroupNorm coefficients
 *             c1 = rstd * mean_g(gamma S0), c2 = rstd^2 * mean_g(gamma S1) of p4c_inorm_apply's rstd == NULL form;
 *             dgamma[c] = sum_b S1[b,c], dbeta[c] = sum_b S0[b,c] (written), S0 / S1 = the two backward sums of p4c_inorm_reduce. */
int p4c_inorm_finalize_fwd(const float* partial, int nblk, int B, int64_t N, int C, int groups, const float* gamma, const float* beta,
                           float eps, float* mean, float* rstd, float* scale, float* shift, p4c_stream_t stream);
int p4c_inorm_finalize_bwd(const float* partial, int nblk, int B, int64_t N, int C, int groups, const float* gamma, const float* rstd,
                           float* c1, float* c2, float* dgamma, float* dbeta, p4c_stream_t stream);

/* ====================================================================================
 * Wide-channel GEMMs / implicit-GEMM convolutions on the bf16 matrix cores (csrc/gemm.hip, round 5): the 3x3 / 1x1 convolutions of
 * 128 ... 1024 channels with their batch norms and the qkvv / out_proj / fc1 / fc2 / reduction Linears of the UNETR++ and SwinUNETR
 * configurations (config/CLI/model/unetrpp.yaml:19-35, swinunetr.yaml:19-30; the reference takes both classes from mfai,
 * py4cast/models.py:10-20, where these are torch.nn.Conv2d / BatchNorm2d / Linear calls).
 * ==================================================================================== */
/* bf16 operand images of an fp32 master weight in the torch layout (CO, CI, taps) -- taps = 9: (CO, CI, 3, 3); taps = 1: a Linear's
 * (out, in) or a 1x1 convolution: fwd[co][tap][ci] = w[co][ci][tap] (forward), dgrad[ci][tap'][co] = w[co][ci][taps-1-tap'] (data
 * gradient: the same kernel on the transposed, tap-flipped image).  Either output may be NULL. */
int p4c_gemm_prep_weight(const float* w, int CO, int CI, int taps, void* fwd, void* dgrad, p4c_stream_t stream);
/* The images of n weights in ONE launch per 24 (round 6: a model prepares all its stale images at the first miss of a step --
 * py4cast_amd/ops_gemm.py::_prepare_logged): parallel arrays of the arguments of p4c_gemm_prep_weight_scaled; rowscale / bias / bias_out
 * (the arrays, or single entries) may be NULL. */
int p4c_gemm_prep_weight_batch(int n, const float* const* w, const float* const* rowscale, const float* const* bias, float* const* bias_out,
                               const int* CO, const int* CI, const int* taps, void* const* fwd, void* const* dgrad, p4c_stream_t stream);
/* The same with a per-output-channel scale folded in: images of rowscale[co] * w[co] (fp32 product, one bf16 rounding) and, with a bias,
 * bias_out[co] = rowscale[co] * bias[co] (fp32).  UNETR++'s transformer block `t + gamma * epa(norm(t))` (layer scale gamma, the class comes
 * through mfai: py4cast/models.py:10-20, config/CLI/model/unetrpp.yaml:19-35): gamma rides in the output projections' weights, so the
 * projection's epilogue writes t + gamma * (...) directly.  rowscale NULL = p4c_gemm_prep_weight; bias / bias_out NULL together. */
int p4c_gemm_prep_weight_scaled(const float* w, const float* rowscale, const float* bias, float* bias_out, int CO, int CI, int taps,
                                void* fwd, void* dgrad, p4c_stream_t stream);
/* Backward bookkeeping of such a layer (z = x W^T + b, y = gamma (.) z) from the RAW gradients of p4c_gemm_tn on the unscaled dy
 * (dw_raw (CO,K) = dy^T x, db_raw (CO) = column sums of dy):  dw = gamma (.) dw_raw,  db = gamma (.) db_raw,
 * dgamma[c] = <dw_raw[c,:], w[c,:]> + b[c] db_raw[c]  (= sum_r dy[r,c] z[r,c], without z).  accumulate != 0: ADD into dw / db / dgamma
 * (the parameters' gradient buffers).  b, db_raw, db NULL together.  Fixed-order sums. */
int p4c_gemm_scale_fold_bwd(const float* dw_raw, const float* db_raw, const float* w, const float* b, const float* gamma, int CO, int K,
                            float* dw, float* db, float* dgamma, int accumulate, p4c_stream_t stream);
/* C (M, N) bf16 = epilogue(A x Bimg^T), Bimg (N, K) bf16 with K contiguous (p4c_gemm_prep_weight), fp32 accumulation.
 * taps = 1: A = (M, K) bf16 rows, row stride lda.  taps = 9: A = an NHWC map (batch, H, W, Cin), M = batch * H * W pixels, pixel
 * stride lda >= Cin, K = 9 * Cin: the 3x3 "same" convolution with zero padding (no im2col buffer).
 * Epilogue, in this order: + bias[n] (fp32, optional); act = 1: the bf16-rounded pre-activation goes to aux_out (row stride ldaux) and
 * GELU (erf form) of it on; act = 2: times GELU'(aux_in) (the data gradient through a GELU); + res[m][n] (bf16, row stride ldr,
 * optional); rounded to bf16 into C (row stride ldc).  stats (optional): [p4c_gemm_nt_stat_blocks][2][N] fp32 column sums and sums of
 * squares of the ROUNDED output (batch-norm statistics; p4c_bnorm_finalize).  N, K, every row stride: multiples of 8.
 * Deep reductions run split-K: workspace of p4c_gemm_nt_workspace_bytes(M, N, K) bytes (0 = not needed), slabs summed in a fixed
 * order (reruns are bit-identical). */
size_t p4c_gemm_nt_workspace_bytes(int M, int N, int K);
int p4c_gemm_nt_stat_blocks(int M, int N, int K);
int p4c_gemm_nt(const void* A, int64_t lda, const void* Bimg, int M, int N, int K, int H, int W, int Cin, int taps, const float* bias,
                const void* res, int64_t ldr, int act, const void* aux_in, void* aux_out, int64_t ldaux, void* C, int64_t ldc,
                float* stats, void* workspace, p4c_stream_t stream);
/* Weight (+ bias) gradient: dw (Mo, Cin, taps) fp32 in the torch layout = sum over the R rows of dy[r][:Mo]^T (x) x[r] -- taps = 9:
 * x's 3x3 neighbourhood of pixel r (x = the NHWC map, R = batch * H * W) --, db (Mo) = column sums of dy (or NULL).  dy / x bf16 rows
 * with strides ldp / ldq (multiples of 8).  Split over the rows, fp32 slabs summed in a fixed order.  accumulate = 1: dw / db are ADDED
 * to (a parameter's .grad buffer: no AccumulateGrad launch per AR step).  workspace:
 * p4c_gemm_tn_workspace_bytes(R, Mo, taps * Cin) bytes. */
size_t p4c_gemm_tn_workspace_bytes(int R, int Mo, int No);
int p4c_gemm_tn(const void* dy, int64_t ldp, const void* x, int64_t ldq, int R, int Mo, int H, int W, int Cin, int taps, float* dw,
                float* db, int accumulate, void* workspace, p4c_stream_t stream);
/* BatchNorm2d (training mode) statistics from column partial sums [nblk][2][C] over `count` values per channel: mean, rstd,
 * scale = gamma rstd, shift = beta - mean scale (C each, fp32); running_mean / running_var (optional) get torch's momentum update
 * with the unbiased variance, num_batches_tracked (optional, the module's int64 counter) is incremented by one. */
int p4c_bnorm_finalize(const float* partial, int nblk, double count, int C, const float* gamma, const float* beta, float eps,
                       float momentum, float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift,
                       int64_t* num_batches_tracked, p4c_stream_t stream);

/* Bilinear up-sampling of a features-last bf16 map (B, H, W, C) by an integer factor (torch interpolate, align_corners = False),
 * + skip (B, H*scale, W*scale, C) when given -- mfai's UnetrUpBlock with `linear_upsampling: true` (config/CLI/model/unetrpp.yaml:29).
 * Backward in gather form (fixed order, no atomics): dx from dout; the skip's gradient is dout.  C a multiple of 8. */
int p4c_upsample_bilinear_fwd(const void* x, const void* skip, void* out, int B, int H, int W, int C, int scale, p4c_stream_t stream);
int p4c_upsample_bilinear_bwd(const void* dout, void* dx, int B, int H, int W, int C, int scale, p4c_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PY4CAST_HIP_H */
