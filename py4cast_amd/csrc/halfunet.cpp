// HalfUNet forward / backward plan: enqueues every kernel of one network call on the caller's
// stream (no host synchronisation, no allocation: all buffers are carved from the two
// caller-provided workspaces).  Architecture: see include/py4cast_hip.h and oracle/halfunet.py.
//
// Two knobs of the descriptor select the kernel flavours:
//   compute = P4C_F32  : exact fp32 matrix cores (conv_f32.hip); activations fp32
//   compute = P4C_BF16 : bf16 matrix cores (conv_bf16.hip); activations / activation gradients stored
//                        fp32 (dtype = P4C_F32) or bf16 (dtype = P4C_BF16).  Parameters, statistics,
//                        normalisation coefficients and weight gradients are always fp32.
#include <stdlib.h>

#include <atomic>
#include <functional>
#include <vector>

#include "kernels.hpp"

namespace p4c {
namespace {

constexpr int NF = 64;       // num_filters
constexpr int NLEV = 5;      // encoder levels
constexpr int NCONV = 12;    // 3x3 convs: enc k conv j -> 2(k-1)+j-1 ; decoder -> 10, 11
// prepared-weight cache in the scratch workspace: slot i = conv i forward, NCONV + i = conv i data gradient,
// 2*NCONV / 2*NCONV + 1 = 1x1 output conv forward / data gradient
constexpr int NWSLOT = 2 * NCONV + 3;   // forward + data-gradient images of the 3x3 blocks, the two of the 1x1 output conv, the first conv's 64-channel row image
constexpr int64_t WSLOT_FLOATS = 9 * 96 * 64;

struct Layout {
    int esz;                  // bytes per activation element
    int64_t n[NLEV];          // B*Hk*Wk per level
    int Hk[NLEV], Wk[NLEV];
    // parameter offsets (floats)
    int64_t w[NCONV], gamma[NCONV], beta[NCONV], wout, nparams;
    // saved workspace: activations (element offsets) then normalisation arrays (float offsets from norm_base bytes)
    int64_t Y[NCONV], P[NLEV], S, act_elems, norm_base, norm[NCONV], saved_bytes;
    // scratch: fp32 region (float offsets) then gradient buffers (element offsets from g_base bytes)
    int64_t wprep, statp, wgradp[NCONV + 1], ocbp[2], nbwdp, k1i[2][NCONV], k2i[2][NCONV], tickets, f_floats, g_base, G0, G1, G2, TB, DY[2][NCONV], scratch_bytes;
    int G;  // workgroups of the weight-gradient kernels
};

inline int conv_level(int i) { return i < 10 ? i / 2 : 0; }
inline int conv_cin(const p4c_halfunet_desc& d, int i) { return i == 0 ? d.cin : NF; }
inline int conv_cin_pad(const p4c_halfunet_desc& d, int i) { return i == 0 ? d.cin_pad : NF; }
inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

inline int stat_tiles(int compute, int storage, int CI, int B, int H, int W, int ks = 3) {
    if (compute == P4C_BF16) return conv_bf16_stat_slots(CI, storage, B, H, W, ks);
    return ((H + CONV_TH - 1) / CONV_TH) * ((W + CONV_TW - 1) / CONV_TW);
}

// dtype-dispatching wrappers: one call site per use, all kernel flavours
inline int prep_w(int compute, const float* w, int CO, int CI, int ks, int tf, int M_pad, int K_pad, void* out, hipStream_t st) {
    return compute == P4C_BF16 ? prep_weights_bf16(w, CO, CI, ks, tf, M_pad, K_pad, out, st)
                               : prep_weights(w, CO, CI, ks, tf, M_pad, K_pad, (float*)out, st);
}
inline int conv_fwd(int compute, int storage, const void* in, int CI, const void* wp, int ks, const float* sc, const float* sh,
                    int relu, void* out, int out_cs, float* statp, int B, int H, int W, int mblocks, hipStream_t st,
                    const BatchFin* fin = nullptr, const RingBwdStats* bst = nullptr, int* nblk_out = nullptr,
                    const NormBwdCoef* nb = nullptr) {
    return compute == P4C_BF16
               ? conv_fwd_bf16(in, storage, CI, wp, ks, sc, sh, relu, out, out_cs, statp, B, H, W, mblocks, st, fin, bst, nblk_out, nb)
               : conv_fwd_f32((const float*)in, CI, (const float*)wp, ks, sc, sh, relu, nullptr, (float*)out, out_cs, statp, B, H, W,
                              mblocks, st);
}
inline int conv_wgrad(int compute, int storage, const void* in, int CI, int ks, const float* sc, const float* sh, int relu,
                      const void* dout, float* partial, int G, int B, int H, int W, int CO, int CIreal, float* grad,
                      hipStream_t st, const NormBwdCoef* nb = nullptr) {
    return compute == P4C_BF16
               ? conv_wgrad_bf16(in, storage, CI, ks, sc, sh, relu, dout, partial, G, B, H, W, CO, CIreal, grad, st, nb)
               : conv_wgrad_f32((const float*)in, CI, ks, sc, sh, relu, (const float*)dout, partial, G, B, H, W, CO, CIreal, grad, st);
}

int check_desc(const p4c_halfunet_desc* d) {
    P4C_CHECK_ARG(d, "halfunet: null descriptor");
    P4C_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0, "halfunet: bad shape");
    P4C_CHECK_ARG(d->H % 16 == 0 && d->W % 16 == 0, "halfunet: H and W must be multiples of 16 (got %dx%d)", d->H, d->W);
    P4C_CHECK_ARG(d->cin_pad == 32 || d->cin_pad == 64 || d->cin_pad == 96,
                  "halfunet: cin_pad must be 32, 64 or 96 (got %d)", d->cin_pad);
    P4C_CHECK_ARG(d->cin > 0 && d->cin <= d->cin_pad, "halfunet: cin out of range");
    P4C_CHECK_ARG(d->cout > 0 && d->cout <= 64, "halfunet: cout must be in 1..64 (got %d)", d->cout);
    P4C_CHECK_ARG(d->dx_channels >= 0 && d->dx_channels <= 64 && d->dx_channels <= d->cin, "halfunet: bad dx_channels");
    P4C_CHECK_ARG(d->norm == 0 || (d->norm == 1 && d->groups > 0 && 64 % d->groups == 0), "halfunet: bad norm/groups");
    P4C_CHECK_ARG(d->compute == P4C_F32 || d->compute == P4C_BF16, "halfunet: compute must be P4C_F32 or P4C_BF16");
    P4C_CHECK_ARG(d->dtype == P4C_F32 || d->dtype == P4C_BF16, "halfunet: dtype must be P4C_F32 or P4C_BF16");
    if (d->dtype == P4C_BF16 && d->compute != P4C_BF16)
        return fail(P4C_ERR_UNSUPPORTED, "halfunet: bf16 activation storage needs compute = P4C_BF16");
    if (d->has_bias) return fail(P4C_ERR_UNSUPPORTED, "halfunet: conv bias (settings.bias=True) is not implemented");
    return P4C_OK;
}

void make_layout(const p4c_halfunet_desc& d, Layout& L) {
    L.esz = d.dtype == P4C_BF16 ? 2 : 4;
    for (int k = 0; k < NLEV; ++k) {
        L.Hk[k] = d.H >> k;
        L.Wk[k] = d.W >> k;
        L.n[k] = (int64_t)d.B * L.Hk[k] * L.Wk[k];
    }
    int64_t off = 0;
    for (int i = 0; i < NCONV; ++i) {
        L.w[i] = off; off += (int64_t)NF * conv_cin(d, i) * 9;
        L.gamma[i] = off; off += NF;
        L.beta[i] = off; off += NF;
    }
    L.wout = off; off += (int64_t)d.cout * NF;
    L.nparams = off;

    off = 0;
    for (int i = 0; i < NCONV; ++i) { L.Y[i] = off; off += L.n[conv_level(i)] * NF; }
    L.P[0] = -1;
    for (int k = 1; k < NLEV; ++k) { L.P[k] = off; off += L.n[k] * NF; }
    L.S = off; off += L.n[0] * NF;
    L.act_elems = off;
    L.norm_base = align256(off * L.esz);
    off = 0;
    for (int i = 0; i < NCONV; ++i) { L.norm[i] = off; off += 4 * (int64_t)d.B * NF; }
    L.saved_bytes = L.norm_base + off * (int64_t)sizeof(float);

    // The weight-gradient kernels run beside the main stream's kernels, and every one of their workgroups holds a whole CU (LDS,
    // registers) for the ~100 us of its segment: HALF of the CUs measured best (round 4, row-streaming kernel: 64 / 96 / 112 / 128 /
    // 144 / 160 / 192 / 256 workgroups -> 5.41 / 4.83 / 4.79 / 4.74 / 4.81 / 4.83 / 4.92 / 5.24 ms per step on one box,
    // profiles/r04_step_ab_runs.txt) -- the other half stays free for the backward chain's own kernels, and there are fewer
    // per-workgroup partials to write and reduce.  P4C_WGRAD_G overrides.  Measured and dropped in round 4: the last two weight
    // gradients of a rollout's backward on the whole chip (nothing is left to overlap with there: no difference), and partials that
    // ACCUMULATE over the AR steps of a rollout with one reduction at the join (the read-modify-write epilogue exposes the partials'
    // memory latency at the end of every launch: weight-gradient launches 104 -> 141 us, the step 4.96 -> 6.33 ms).
    L.G = d.compute == P4C_BF16 ? num_cus() / 2 : num_cus();   // (fp32 matrix cores: the kernel is MFMA-bound, all CUs)
    if (const char* e = diag_env("P4C_WGRAD_G")) { const int g = atoi(e); if (g > 0 && g <= num_cus()) L.G = g; }
    off = 0;
    L.wprep = off; off += (int64_t)NWSLOT * WSLOT_FLOATS;
    const int64_t tps = conv_tiles_per_sample(d.H, d.W);
    {
        int64_t slots = tps > 4 * (int64_t)num_cus() ? tps : 4 * (int64_t)num_cus();
        for (int k = 0; k < NLEV; ++k) {   // (the row-streaming kernel's slot count depends on the level's shape)
            const int64_t s = d.compute == P4C_BF16 ? conv_bf16_stat_slots(NF, d.dtype, d.B, L.Hk[k], L.Wk[k], 3) : 0;
            if (s > slots) slots = s;
        }
        L.statp = off; off += (int64_t)d.B * slots * 128;
    }
    // per-workgroup weight-gradient partials, one region per convolution (NCONV: the 1x1 output convolution): the reductions of a
    // backward call run as ONE launch at its end (wgrad_reduce_batch), so every launch's partials live until then
    for (int i = 0; i <= NCONV; ++i) {
        const int lev = i < NCONV ? conv_level(i) : 0;
        int64_t tiles = (int64_t)d.B * conv_tiles_per_sample(L.Hk[lev], L.Wk[lev]);
        const int g = tiles < L.G ? (int)tiles : L.G;
        L.wgradp[i] = off; off += wgrad_partial_floats(i < NCONV ? conv_cin_pad(d, i) : NF, i < NCONV ? 3 : 1, g);
    }
    // the fused backward of the output convolution (out_conv_bwd.hip) runs on the MAIN stream and leaves one partial per workgroup for
    // the call's batched reduction, which runs on the weight-gradient stream -- with the join deferred over the AR steps, possibly after
    // the next call's launch: two regions, alternating like the dY sets
    for (int s = 0; s < 2; ++s) {
        L.ocbp[s] = off;
        if (d.dtype == P4C_BF16) off += (int64_t)d.B * out_conv_bwd_slots(d.B, (int64_t)d.H * d.W) * 4096;
    }
    L.nbwdp = off; off += (int64_t)d.B * NORM_BWD_MAX_BLOCKS * 128;
    // k1 / k2 of every block's normalisation backward, per gradient-buffer set: the weight-gradient kernels of the side stream read
    // them (NormBwdCoef) after the main stream has gone on to other blocks -- and, with the deferred join, to the next call
    for (int s = 0; s < 2; ++s)
        for (int i = 0; i < NCONV; ++i) {
            L.k1i[s][i] = off; off += (int64_t)d.B * NF;
            L.k2i[s][i] = off; off += (int64_t)d.B * NF;
        }
    L.tickets = off; off += 64;   // ticket counters of the in-kernel finalizes (uint32 words, cleared by the weight preparation)
    L.f_floats = off;
    L.g_base = align256(off * (int64_t)sizeof(float));
    off = 0;
    L.G0 = off; off += L.n[0] * NF;
    L.G1 = off; off += L.n[0] * NF;
    L.G2 = off; off += L.n[0] * NF;
    L.TB = off; off += L.n[0] * NF;  // x-pass outputs of the four up-sampling adjoints: n0*NF*(1/2+1/4+1/8+1/16)
    // dY (gradient wrt the raw conv output) of every conv block in its own buffer: the weight-gradient kernels read them
    // from a side stream while the main stream already overwrites the rotating buffers G0..G2
    // (two sets, alternating between consecutive backward calls: with the side-stream join deferred across the AR steps of a
    // rollout -- p4c_side_stream_defer -- the weight gradients of one call may still be reading their dY while the next call's
    // chain writes its own)
    for (int s = 0; s < 2; ++s)
        for (int i = 0; i < NCONV; ++i) { L.DY[s][i] = off; off += L.n[conv_level(i)] * NF; }
    L.scratch_bytes = L.g_base + off * L.esz;
}

// workspace views
struct WS {
    const Layout& L;
    char* saved;
    char* scratch;
    void* act(int64_t elem_off) const { return saved + elem_off * L.esz; }          // saved activation
    float* nrm(int64_t float_off) const { return reinterpret_cast<float*>(saved + L.norm_base) + float_off; }
    float* f(int64_t float_off) const { return reinterpret_cast<float*>(scratch) + float_off; }  // fp32 scratch
    void* g(int64_t elem_off) const { return scratch + L.g_base + elem_off * L.esz; }             // gradient buffers
};

struct Norm { float *scale, *shift, *mean, *rstd; };
inline Norm norm_at(const WS& ws, int i, int B) {
    float* p = ws.nrm(ws.L.norm[i]);
    return {p, p + (int64_t)B * NF, p + 2 * (int64_t)B * NF, p + 3 * (int64_t)B * NF};
}

inline void* wslot(const WS& ws, int slot) { return ws.f(ws.L.wprep + (int64_t)slot * WSLOT_FLOATS); }

// (re)prepares the weight operand streams: which = 1 forward orientation, 2 data-gradient orientation, 3 both
int prepare_weights(const p4c_halfunet_desc& d, const WS& ws, const float* params, int which, hipStream_t st) {
    const Layout& L = ws.L;
    PrepBatch pb;
    pb.n = 0;
    pb.bf16 = d.compute == P4C_BF16;
    for (int i = 0; i < NCONV; ++i) {
        if (which & 1) pb.job[pb.n++] = {params + L.w[i], wslot(ws, i), NF, conv_cin(d, i), 9, 0, 64, conv_cin_pad(d, i)};
        // data gradient: M = input channel (the first 64 at most: dx_channels <= 64), K = output channel, taps flipped
        if (which & 2) pb.job[pb.n++] = {params + L.w[i], wslot(ws, NCONV + i), NF, conv_cin(d, i), 9, 1, 64, NF};
    }
    if (which & 1) pb.job[pb.n++] = {params + L.wout, wslot(ws, 2 * NCONV), d.cout, NF, 1, 0, 64, NF};
    // the first convolution as a 64-channel row launch + tail (conv_thin.hip): the row kernel's image of input channels 0..63
    if ((which & 1) && first_conv_split_ok(d.compute, d.dtype, d.cin, d.cin_pad, d.B, d.H, d.W))
        pb.job[pb.n++] = {params + L.w[0], wslot(ws, 2 * NCONV + 2), NF, d.cin, 9, 0, 64, NF};
    if (which & 2) pb.job[pb.n++] = {params + L.wout, wslot(ws, 2 * NCONV + 1), d.cout, NF, 1, 1, 64, NF};
    pb.zero_words = reinterpret_cast<unsigned int*>(ws.f(L.tickets));
    pb.n_zero = 64;
    return prep_weights_batch(pb, st);
}

// conv3x3 forward + statistics + normalisation parameters of its output
int conv_block_fwd(const p4c_halfunet_desc& d, const WS& ws, int i, const void* in, const Norm* in_norm, const float* params,
                   float* running, int training, hipStream_t st) {
    const Layout& L = ws.L;
    const int lev = conv_level(i), H = L.Hk[lev], W = L.Wk[lev];
    void* wp = wslot(ws, i);
    const bool batch_stats = (d.norm == 1) || training;
    float* statp = batch_stats ? ws.f(L.statp) : nullptr;
    Norm nm = norm_at(ws, i, d.B);
    float* rm = running ? running + (int64_t)i * 128 : nullptr;
    float* rv = running ? rm + 64 : nullptr;
    // BatchNorm statistics of a ring-kernel convolution are finished by the kernel itself (its last workgroup): no norm_finalize launch
    const char* ie = diag_env("P4C_NO_INKERNEL_FINALIZE");
    const bool no_infin = ie && ie[0] == '1';
    // (below the full resolution only: there the in-kernel tail -- slot store, ticket, the last workgroup's 128 KB read -- costs
    // ~5 us against ~8 for the launch it replaces and stretches the roofline kernel's launches by that much; on the coarse levels
    // every launch is latency and one fewer is a clean gain.  P4C_INKERNEL_FINALIZE_ALL=1 turns it on everywhere.)
    const char* ia = diag_env("P4C_INKERNEL_FINALIZE_ALL");
    const bool all_levels = ia && ia[0] == '1';
    const bool infin = batch_stats && d.norm == 0 && d.compute == P4C_BF16 && !no_infin && (all_levels || lev > 0) &&
                       conv_bf16_is_ring(d.dtype, conv_cin_pad(d, i), 3, 1, NF, d.B, H, W);
    BatchFin fin{};
    if (infin) {
        fin = BatchFin{statp, reinterpret_cast<unsigned int*>(ws.f(L.tickets)), params + L.gamma[i], params + L.beta[i], rm, rv,
                       nm.scale, nm.shift, nm.mean, nm.rstd, (double)d.B * H * W, d.eps, d.momentum, d.B};
    }
    int ntiles = stat_tiles(d.compute, d.dtype, conv_cin_pad(d, i), d.B, H, W);
    if (i == 0 && !in_norm && first_conv_split_ok(d.compute, d.dtype, d.cin, d.cin_pad, d.B, H, W)) {
        // 69 -> 64 at the benchmark's channel count: a 64-channel row launch on channels 0..63 of the 96-channel pixels, then the tail
        // adds the product of the channels beyond 64 and takes the statistics of the result (conv_thin.hip)
        P4C_TRY(launch_conv_bf16_rows_wide_pixels(in, d.cin_pad, wslot(ws, 2 * NCONV + 2), ws.act(L.Y[i]), d.B, H, W, st));
        P4C_TRY(launch_first_conv_tail(in, d.cin_pad, d.cin, params + L.w[0], ws.act(L.Y[i]), statp, d.B, H, W, st));
        ntiles = first_conv_tail_slots(d.B, H, W);
    } else {
        P4C_TRY(conv_fwd(d.compute, d.dtype, in, conv_cin_pad(d, i), wp, 3, in_norm ? in_norm->scale : nullptr,
                         in_norm ? in_norm->shift : nullptr, in_norm ? 1 : 0, ws.act(L.Y[i]), NF, statp, d.B, H, W, 1, st,
                         infin ? &fin : nullptr));
    }
    if (infin) return P4C_OK;
    if (batch_stats) {
        P4C_TRY(norm_finalize(statp, ntiles, d.B, (int64_t)H * W, d.norm,
                              d.groups, params + L.gamma[i], params + L.beta[i], d.eps, d.momentum,
                              d.norm == 0 ? rm : nullptr, d.norm == 0 ? rv : nullptr, nm.scale, nm.shift, nm.mean, nm.rstd, st));
    } else {
        P4C_CHECK_ARG(running, "halfunet: eval-mode BatchNorm needs the running statistics");
        P4C_TRY(norm_eval(d.B, params + L.gamma[i], params + L.beta[i], d.eps, rm, rv, nm.scale, nm.shift, nm.mean,
                          nm.rstd, st));
    }
    return P4C_OK;
}

// Side stream for the weight gradients.  Nothing on the backward chain waits for a weight gradient, and half of that
// chain is latency-bound work on the coarse levels that occupies a few CUs: the weight-gradient kernels (+ their
// reductions) are enqueued on a second stream, ordered after the dY they read by an event, and joined back into the
// caller's stream at the end of the call.
// Ordering events between two streams of ONE device: no timing, and no system-scope release on record -- the consumer is a kernel on
// the same GPU, for which the agent-scope release at the end of every kernel is enough (tools/diagnostics/event_cost.hip: a record
// costs the main stream 7.2 us with the fence, 4.6 us without).
constexpr unsigned kOrderFlags = hipEventDisableTiming | hipEventDisableSystemFence;
void count_side_launches(long long n);
struct SideStream {
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> events;
    size_t next = 0;
    bool enabled = true;
    bool external = false;   // stream and events were handed in by the caller (p4c_set_side_stream): create nothing
    // weight-gradient launches waiting for their ordering event: an event record costs the MAIN stream a bubble (rocprofv3: the
    // kernel after one starts 8 us later in the median, 20 us on average), so several blocks share one
    std::vector<std::function<int(hipStream_t)>> pending;
    // measured on the benchmark configuration -- round 1: 1 -> 6.06, 2 -> 5.97, 3 -> 5.91, 4 -> 5.99, 6 -> 6.14 ms per step;
    // round 3 (fence-free events, deferred join): 1 -> 5.19, 2 -> 5.156, 3 -> 5.168, 4 -> 5.29, 6 -> 5.39, 12 -> 5.38
    int every = 2;
    // Join deferred over several backward calls (p4c_side_stream_defer): the weight gradients left over at the end of one AR
    // step's backward -- the full-resolution ones, which find nothing to hide behind there -- then run beside the HBM-bound head of
    // the next AR step's chain instead of alone.  `calls` picks the dY buffer set; set_done[s] is recorded on the side stream after
    // the last weight-gradient launch of a call that used set s, and the next call that uses s waits for it (two calls later:
    // long signalled); dy_read is recorded after the 1x1 convolution's weight gradient, which reads the CALLER's dy buffer.
    bool defer = false;
    unsigned long long calls = 0;
    hipEvent_t set_done[2] = {nullptr, nullptr};
    bool set_recorded[2] = {false, false};
    hipEvent_t dy_read = nullptr;
    int own_events() {
        if (dy_read) return P4C_OK;
        for (int i = 0; i < 2; ++i) P4C_CHECK_HIP(hipEventCreateWithFlags(&set_done[i], kOrderFlags));
        P4C_CHECK_HIP(hipEventCreateWithFlags(&dy_read, kOrderFlags));
        return P4C_OK;
    }
    int init() {
        if (stream || external) return P4C_OK;
        const char* e = diag_env("P4C_SIDE_STREAM");
        enabled = !(e && e[0] == '0');
        if (const char* n = diag_env("P4C_SIDE_EVERY")) { const int v = atoi(n); if (v > 0) every = v; }
        if (!enabled) return P4C_OK;
        // (A/B switch P4C_SIDE_PRIO: -1 = lowest, 1 = highest queue priority for the weight-gradient stream)
        const char* pr = diag_env("P4C_SIDE_PRIO");
        if (pr && pr[0] != '0') {
            int lo = 0, hi = 0;
            P4C_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            P4C_CHECK_HIP(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, atoi(pr) < 0 ? lo : hi));
        } else {
            P4C_CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        }
        return P4C_OK;
    }
    // run the deferred launches on the side stream after everything enqueued on `from` so far (ONE event for all of them)
    int flush(hipStream_t from) {
        if (pending.empty()) return P4C_OK;
        P4C_TRY(order(from, stream));
        count_side_launches((long long)pending.size());
        for (auto& job : pending) P4C_TRY(job(stream));
        pending.clear();
        return P4C_OK;
    }
    int event(hipEvent_t* ev) {
        if (next == events.size()) {
            if (external)
                return fail(P4C_ERR_INVALID, "p4c_halfunet_backward: the caller's %zu ordering events are used up "
                                             "(p4c_set_side_stream: pass at least 8)", events.size());
            hipEvent_t e;
            P4C_CHECK_HIP(hipEventCreateWithFlags(&e, kOrderFlags));
            events.push_back(e);
        }
        *ev = events[next++];
        return P4C_OK;
    }
    // make `to` wait for everything enqueued on `from` so far
    int order(hipStream_t from, hipStream_t to) {
        hipEvent_t ev;
        P4C_TRY(event(&ev));
        P4C_CHECK_HIP(hipEventRecord(ev, from));
        P4C_CHECK_HIP(hipStreamWaitEvent(to, ev, 0));
        return P4C_OK;
    }
};
thread_local SideStream g_side;
// p4c_side_stream_enable is PROCESS-wide: a host sets it from its main thread, the backward runs on autograd's device
// thread, whose own (thread-local) SideStream would never see a thread-local flag.  -1 = never asked: each thread's default.
std::atomic<int> g_side_enable_request{-1};
// weight-gradient jobs this process has issued to a side stream (p4c_side_stream_launch_count: what the single-stream test asserts on)
std::atomic<long long> g_side_launches{0};
void count_side_launches(long long n) { g_side_launches.fetch_add(n); }
int apply_side_request() {
    const int req = g_side_enable_request.load();
    if (req < 0 || g_side.external) return P4C_OK;
    if (req && !g_side.stream) P4C_CHECK_HIP(hipStreamCreateWithFlags(&g_side.stream, hipStreamNonBlocking));
    g_side.enabled = req != 0;
    return P4C_OK;
}

// The dA (gradient wrt the post-ReLU activation) of every conv block lives in the block's own buffer of the current set
// (Layout::DY): its producer -- the data gradient of the next block, enc_out_bwd, the 1x1 data gradient -- writes it there, the
// normalisation backward turns it into dY in place (or leaves that to the consumers, below), and the weight gradient reads it from
// the side stream while the main stream has long moved on.
inline void* block_grad(const WS& ws, int i) { return ws.g(ws.L.DY[g_side.calls & 1][i]); }

// In-kernel finish of a normalisation backward's pass 1 (kernels.hpp: BwdFin; BatchNorm, bf16 storage): ticket word 1 of the plan's
// counters (word 0: the forward's BatchFin), or nullptr = every pass is finished by a norm_bwd_finalize launch
inline unsigned int* bwd_ticket(const p4c_halfunet_desc& d, const WS& ws) {
    return (d.norm == 0 && d.dtype == P4C_BF16) ? reinterpret_cast<unsigned int*>(ws.f(ws.L.tickets)) + 1 : nullptr;
}
inline BwdFin bwd_fin(const p4c_halfunet_desc& d, const WS& ws, int i, const float* params, float* grads, int training) {
    const Layout& L = ws.L;
    const int lev = conv_level(i);
    return BwdFin{bwd_ticket(d, ws), params + L.gamma[i], grads + L.gamma[i], grads + L.beta[i], ws.f(L.k1i[g_side.calls & 1][i]),
                  ws.f(L.k2i[g_side.calls & 1][i]), (float)d.B * (float)((int64_t)L.Hk[lev] * L.Wk[lev]), d.B,
                  ((d.norm == 1) || training) ? 1 : 0};
}

// backward through [conv i -> norm -> relu] given dA in block_grad(i):
//   grads of gamma/beta/weight accumulated; if `din` != null: din = dL/d(conv input).
// pre_nblk > 0: the producer of dA took pass 1 of the normalisation backward (partial sums in ws.nbwdp).  Where the consumers can
// (conv_bf16_norm_bwd_fused_ok, and see `dgrad_takes_pass1`), pass 2 is theirs: no norm_bwd_apply launch, no dY map -- the
// data-gradient row kernel and the weight-gradient kernel form dY = alpha * g + beta * y + delta while they load dA and y.
int conv_block_bwd(const p4c_halfunet_desc& d, const WS& ws, int i, const void* in, const Norm* in_norm, void* din,
                   const float* params, float* grads, int training, hipStream_t st, int pre_nblk = 0, int* next_nblk = nullptr,
                   bool pre_finalized = false) {
    const Layout& L = ws.L;
    const int lev = conv_level(i), H = L.Hk[lev], W = L.Wk[lev];
    Norm nm = norm_at(ws, i, d.B);
    const int stats_training = (d.norm == 1) || training;
    void* g = block_grad(ws, i);
    const int cip = conv_cin_pad(d, i);
    // (the data-gradient launch of this block must not also take pass 1 of the next normalisation: both loaders in one kernel
    // exceed the register file -- so the blocks whose input is a pooled / summed map: conv 10, 2, 4, 6)
    const bool dgrad_takes_pass1 = din && in_norm && i > 0;
    // Consumer-side pass 2 for EVERY block: where the data-gradient launch would also take pass 1 of the next normalisation (both
    // loaders in one kernel exceed the register file) it gives that up and a norm_bwd_reduce launch (2 reads) takes it -- instead of
    // a norm_bwd_apply launch (2 reads + 1 write) here: 4.96 -> 4.91 ms per step.  P4C_NB_ALL=0: the earlier split.
    static const bool nb_all = [] { const char* e = diag_env("P4C_NB_ALL"); return !(e && e[0] == '0'); }();
    // (the first convolution's 96-channel input: its weight gradient runs as a 64-channel chunk + a half-empty one, both on the
    // role-split kernel -- conv_wgrad_bf16_takes_nb -- and its data gradient, state channels only, is a 64 -> 64 launch)
    const bool nbf = (!dgrad_takes_pass1 || nb_all) && d.compute == P4C_BF16 && conv_bf16_norm_bwd_fused_ok(d.dtype, NF, d.B, H, W) &&
                     (cip == NF || (nb_all && diag_env("P4C_NO_NB0") == nullptr && conv_wgrad_bf16_takes_nb(d.dtype, cip, 3, d.B)));
    P4C_TRY(norm_bwd(d.dtype, g, ws.act(L.Y[i]), nm.scale, nm.shift, nm.mean, nm.rstd, params + L.gamma[i], 1, d.B,
                     (int64_t)H * W, d.norm, d.groups, stats_training, ws.f(L.nbwdp), ws.f(L.k1i[g_side.calls & 1][i]), ws.f(L.k2i[g_side.calls & 1][i]),
                     grads + L.gamma[i], grads + L.beta[i], nbf ? nullptr : g, st, pre_nblk, bwd_ticket(d, ws), pre_finalized));
    const NormBwdCoef nb{ws.act(L.Y[i]), params + L.gamma[i], nm.scale, nm.shift, nm.rstd, nm.mean,
                         ws.f(L.k1i[g_side.calls & 1][i]), ws.f(L.k2i[g_side.calls & 1][i])};
    int64_t ntiles = (int64_t)d.B * conv_tiles_per_sample(H, W);
    const int G = ntiles < L.G ? (int)ntiles : L.G;
    // the weight gradient needs dY (complete at this point of the main stream) and the saved input: every block has its own
    // gradient buffer, so the launch can be deferred and share its ordering event with the next blocks'
    const int compute = d.compute, dtype = d.dtype, Bn = d.B, cin = conv_cin(d, i);
    const float* isc = in_norm ? in_norm->scale : nullptr;
    const float* ish = in_norm ? in_norm->shift : nullptr;
    const int irelu = in_norm ? 1 : 0;
    float* wpart = ws.f(L.wgradp[i]);
    float* gw = grads + L.w[i];
    auto job = [=](hipStream_t s) {
        return conv_wgrad(compute, dtype, in, cip, 3, isc, ish, irelu, g, wpart, G, Bn, H, W, NF, cin, gw, s, nbf ? &nb : nullptr);
    };
    if (g_side.enabled) {
        g_side.pending.push_back(job);
        // (block 1 closes a batch whatever its size: the full-resolution weight gradient of conv 1 then runs beside the chain of
        // block 0 instead of after it, and only block 0's is left over when the main stream ends -- with the join deferred over
        // the AR steps that left-over is exposed once per optimizer step: 240 -> ~60 us)
        const char* f1 = diag_env("P4C_FLUSH_AT_1");   // (A/B switch)
        const bool at1 = i == 1 && !g_side.external && !(f1 && f1[0] == '0');
        if ((int)g_side.pending.size() >= g_side.every || at1) P4C_TRY(g_side.flush(st));
    } else {
        P4C_TRY(job(st));
    }
    if (next_nblk) *next_nblk = 0;
    if (din) {
        // the input of this convolution is relu(norm(Y[i-1])): its data gradient IS the dA of layer i-1's normalisation backward,
        // whose pass 1 (sums of g and g * xhat) the ring / row kernel takes while it stores the gradient rows
        const char* fe = diag_env("P4C_NO_FUSED_REDUCE");   // (read per call: the parity test switches it)
        const bool fuse_off = fe && fe[0] == '1';
        // Both roles in one launch -- pass 2 of this block's normalisation backward in the loader, pass 1 of the next block's in the
        // drain -- do not fit the register file at four rows per interval (137 spilled registers; 59 with the roles' constants in
        // LDS: 55 -> 120 us per launch, the step 4.79 -> 5.36 ms).  At TWO rows per interval (half-size loader images, no spills:
        // the row kernel's IV) the launch works -- 70 us at full resolution against 50 + a 35 us norm_bwd_reduce launch, 15 launches
        // fewer per step -- but the step does not get faster: 4.60 vs 4.61 ms at 2x512x512x60, 4.88 vs 4.82 at the Titan shape
        // (profiles/r04_step_ab_runs.txt block 20: the weight-gradient stream pays for what the chain saves).  Kept behind
        // P4C_NB_BST=1, parity-tested both ways (tests/test_bwd_infin_gpu.py).
        const char* nbe = diag_env("P4C_NB_BST");   // (read per call: the parity test switches it)
        const bool nb_bst = nbe && nbe[0] == '1';
        const bool fuse = !fuse_off && next_nblk && in_norm && i > 0 && d.compute == P4C_BF16 && (!(nbf && dgrad_takes_pass1) || nb_bst) &&
                          conv_bf16_bwd_stats_ok(d.dtype, d.B, H, W);
        const RingBwdStats bst{in, in_norm ? in_norm->scale : nullptr, in_norm ? in_norm->shift : nullptr,
                               in_norm ? in_norm->mean : nullptr, in_norm ? in_norm->rstd : nullptr};
        P4C_TRY(conv_fwd(d.compute, d.dtype, g, NF, wslot(ws, NCONV + i), 3, nullptr, nullptr, 0, din, 64, fuse ? ws.f(L.nbwdp) : nullptr,
                         d.B, H, W, 1, st, nullptr, fuse ? &bst : nullptr, fuse ? next_nblk : nullptr, nbf ? &nb : nullptr));
    }
    return P4C_OK;
}

}  // namespace
}  // namespace p4c

using namespace p4c;

extern "C" int p4c_set_side_stream(p4c_stream_t side, void* const* events, int n_events) {
    // per calling thread (the backward plan runs on autograd's thread): replaces the lazily created stream / event pool
    SideStream& sd = g_side;
    if (!side) {   // back to the library-owned defaults (created on first use)
        if (sd.external) { sd.stream = nullptr; sd.events.clear(); sd.external = false; sd.enabled = true; }
        return P4C_OK;
    }
    P4C_CHECK_ARG(events && n_events >= 8, "p4c_set_side_stream: needs >= 8 ordering events (hipEvent_t, timing disabled)");
    sd.stream = as_stream(side);
    sd.events.assign(reinterpret_cast<hipEvent_t const*>(events), reinterpret_cast<hipEvent_t const*>(events) + n_events);
    sd.next = 0;
    sd.external = true;
    sd.enabled = true;
    return P4C_OK;
}

extern "C" int64_t p4c_halfunet_param_count(const p4c_halfunet_desc* d) {
    if (check_desc(d) != P4C_OK) return -1;
    Layout L;
    make_layout(*d, L);
    return L.nparams;
}

extern "C" int p4c_halfunet_workspace_bytes(const p4c_halfunet_desc* d, size_t* saved_bytes, size_t* scratch_bytes) {
    P4C_TRY(check_desc(d));
    Layout L;
    make_layout(*d, L);
    if (saved_bytes) *saved_bytes = (size_t)L.saved_bytes;
    if (scratch_bytes) *scratch_bytes = (size_t)L.scratch_bytes;
    return P4C_OK;
}

extern "C" int p4c_halfunet_prepare_weights(const p4c_halfunet_desc* dp, const float* params, void* scratchv,
                                            p4c_stream_t stream) {
    P4C_TRY(check_desc(dp));
    P4C_CHECK_ARG(params && scratchv, "p4c_halfunet_prepare_weights: null pointer");
    Layout L;
    make_layout(*dp, L);
    const WS ws{L, nullptr, (char*)scratchv};
    return prepare_weights(*dp, ws, params, 3, as_stream(stream));
}

extern "C" int p4c_halfunet_forward(const p4c_halfunet_desc* dp, const void* x, const float* params, float* running,
                                    void* y, void* savedv, void* scratchv, int training, p4c_stream_t stream) {
    P4C_TRY(check_desc(dp));
    P4C_CHECK_ARG(x && params && (y || dp->skip_out_conv) && savedv && scratchv, "p4c_halfunet_forward: null pointer");
    const p4c_halfunet_desc& d = *dp;
    Layout L;
    make_layout(d, L);
    hipStream_t st = as_stream(stream);
    const WS ws{L, (char*)savedv, (char*)scratchv};
    if (!d.weights_prepared) P4C_TRY(prepare_weights(d, ws, params, 1, st));

    // encoder (a persistent one-launch form of levels 2 .. 4 with grid-wide barriers was built in round 4, measured slower -- 925
    // against 673 us per forward -- and removed in round 5: profiles/r04_step_ab_runs.txt block 25, profiles/HISTORY.md)
    for (int k = 0; k < NLEV; ++k) {
        const void* in = k == 0 ? x : ws.act(L.P[k]);
        P4C_TRY(conv_block_fwd(d, ws, 2 * k, in, nullptr, params, running, training, st));
        Norm n1 = norm_at(ws, 2 * k, d.B);
        P4C_TRY(conv_block_fwd(d, ws, 2 * k + 1, ws.act(L.Y[2 * k]), &n1, params, running, training, st));
        if (k + 1 < NLEV) {
            Norm n2 = norm_at(ws, 2 * k + 1, d.B);
            P4C_TRY(pool_fwd(d.dtype, ws.act(L.Y[2 * k + 1]), n2.scale, n2.shift, d.B, L.Hk[k], L.Wk[k], ws.act(L.P[k + 1]), st));
        }
    }
    // up-sample every level to full resolution and sum
    {
        const void* ys[NLEV];
        const float *sc[NLEV], *sh[NLEV];
        for (int k = 0; k < NLEV; ++k) {
            Norm n2 = norm_at(ws, 2 * k + 1, d.B);
            ys[k] = ws.act(L.Y[2 * k + 1]); sc[k] = n2.scale; sh[k] = n2.shift;
        }
        P4C_TRY(upsum_fwd(d.dtype, ys, sc, sh, d.B, d.H, d.W, ws.act(L.S), st));
    }
    // decoder
    P4C_TRY(conv_block_fwd(d, ws, 10, ws.act(L.S), nullptr, params, running, training, st));
    Norm nd1 = norm_at(ws, 10, d.B);
    P4C_TRY(conv_block_fwd(d, ws, 11, ws.act(L.Y[10]), &nd1, params, running, training, st));
    Norm nd2 = norm_at(ws, 11, d.B);
    // 1x1 output conv on relu(norm(Y_dec2)); last activation = Identity  (skip_out_conv: the caller fuses it with the AR step)
    if (d.skip_out_conv) return P4C_OK;
    P4C_TRY(conv_fwd(d.compute, d.dtype, ws.act(L.Y[11]), NF, wslot(ws, 2 * NCONV), 1, nd2.scale, nd2.shift, 1, y, NF, nullptr, d.B, d.H, d.W, 1, st));
    return P4C_OK;
}

extern "C" int p4c_halfunet_tail(const p4c_halfunet_desc* dp, const float* params, void* savedv, const void** a, const float** a_scale,
                                 const float** a_shift, const float** wout) {
    P4C_TRY(check_desc(dp));
    P4C_CHECK_ARG(params && savedv && a && a_scale && a_shift && wout, "p4c_halfunet_tail: null pointer");
    Layout L;
    make_layout(*dp, L);
    const WS ws{L, (char*)savedv, nullptr};
    const Norm n = norm_at(ws, 11, dp->B);
    *a = ws.act(L.Y[11]);
    *a_scale = n.scale;
    *a_shift = n.shift;
    *wout = params + L.wout;
    return P4C_OK;
}

extern "C" int p4c_halfunet_backward(const p4c_halfunet_desc* dp, const void* x, const float* params, const void* dy,
                                     void* dx, float* grads, void* savedv, void* scratchv, int training,
                                     p4c_stream_t stream) {
    P4C_TRY(check_desc(dp));
    P4C_CHECK_ARG(x && params && dy && grads && savedv && scratchv, "p4c_halfunet_backward: null pointer");
    const p4c_halfunet_desc& d = *dp;
    P4C_CHECK_ARG(dx || d.dx_channels == 0, "p4c_halfunet_backward: dx is null but dx_channels > 0");
    Layout L;
    make_layout(d, L);
    hipStream_t st = as_stream(stream);
    const WS ws{L, (char*)savedv, (char*)scratchv};
    void *G0 = ws.g(L.G0), *G1 = ws.g(L.G1), *G2 = ws.g(L.G2), *TB = ws.g(L.TB);
    if (!d.weights_prepared) P4C_TRY(prepare_weights(d, ws, params, 2, st));
    P4C_TRY(g_side.init());
    P4C_TRY(apply_side_request());
    g_side.next = 0;
    g_side.pending.clear();
    // (inside a HIP-graph capture nothing may depend on events recorded before it: every call then joins for itself and the
    // cross-call events are left alone)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool plain_stream = g_side.enabled && !g_side.external && cap == hipStreamCaptureStatusNone;
    const bool deferring = plain_stream && g_side.defer;
    if (plain_stream) {
        P4C_TRY(g_side.own_events());
        ++g_side.calls;
        const int set = (int)(g_side.calls & 1);
        if (g_side.set_recorded[set]) P4C_CHECK_HIP(hipStreamWaitEvent(st, g_side.set_done[set], 0));   // (the call before the previous one)
    }
    struct BwdPhase { BwdPhase() { prof_set_backward(true); } ~BwdPhase() { prof_set_backward(false); } } bwd_phase;
    // the per-workgroup partials of this call's 14 weight gradients are reduced by ONE launch at the end of the call (they were 14
    // dependent ~8 us launches on the weight-gradient stream, 42 per 3-step rollout); P4C_WGRAD_BATCH=0: one launch each, as before
    static const bool batch_reduce = [] { const char* e = diag_env("P4C_WGRAD_BATCH"); return !(e && e[0] == '0'); }();
    WgradCollect reduce_jobs;
    struct Collect {
        explicit Collect(WgradCollect* c) { wgrad_collect_into(c); }
        ~Collect() { wgrad_collect_into(nullptr); }
    } collect(batch_reduce ? &reduce_jobs : nullptr);

    // ---- output 1x1 conv
    Norm nd2 = norm_at(ws, 11, d.B);
    int pre11 = 0;
    // bf16 storage: data gradient, pass 1 of conv 11's normalisation backward and the weight gradient in ONE pass over dy and Y[11] on
    // the main stream (out_conv_bwd.hip) -- nothing on the weight-gradient stream reads the caller's dy then
    const bool fused_out = d.compute == P4C_BF16 && out_conv_bwd_ok(d.dtype, d.B, (int64_t)d.H * d.W);
    if (fused_out) {
        float* wpart = ws.f(L.ocbp[g_side.calls & 1]);
        P4C_TRY(launch_out_conv_bwd(dy, wslot(ws, 2 * NCONV + 1), ws.act(L.Y[11]), nd2.scale, nd2.shift, nd2.mean, nd2.rstd,
                                    block_grad(ws, 11), ws.f(L.nbwdp), wpart, d.B, (int64_t)d.H * d.W, st, &pre11));
        P4C_TRY(wgrad_reduce(wpart, d.B * pre11, 1, NF, 0, NF, d.cout, NF, grads + L.wout, st));   // (recorded for the batch, or launched here)
    } else {
        int64_t ntiles = (int64_t)d.B * conv_tiles_per_sample(d.H, d.W);
        const int G = ntiles < L.G ? (int)ntiles : L.G;
        hipStream_t wst = st;
        if (g_side.enabled) {
            P4C_TRY(g_side.order(st, g_side.stream));   // after everything the caller enqueued before this call (dy, ...)
            wst = g_side.stream;
            count_side_launches(1);
        }
        P4C_TRY(conv_wgrad(d.compute, d.dtype, ws.act(L.Y[11]), NF, 1, nd2.scale, nd2.shift, 1, dy, ws.f(L.wgradp[NCONV]), G, d.B, d.H,
                           d.W, d.cout, NF, grads + L.wout, wst));
        if (deferring) P4C_CHECK_HIP(hipEventRecord(g_side.dy_read, g_side.stream));
        // the 1x1 data gradient IS the dA of conv 11's normalisation backward: on the row kernel it takes pass 1 of it
        // (sums of g and g * xhat) while it stores the rows, instead of a norm_bwd_reduce launch over dA and y
        const char* fe = diag_env("P4C_NO_FUSED_REDUCE");
        const bool fuse11 = !(fe && fe[0] == '1') && d.compute == P4C_BF16 && conv_bf16_is_rows(d.dtype, NF, 1, 1, NF, d.B, d.H, d.W) &&
                            conv_bf16_bwd_stats_ok(d.dtype, d.B, d.H, d.W);
        if (fuse11) {
            const RingBwdStats bst{ws.act(L.Y[11]), nd2.scale, nd2.shift, nd2.mean, nd2.rstd};
            P4C_TRY(conv_fwd(d.compute, d.dtype, dy, NF, wslot(ws, 2 * NCONV + 1), 1, nullptr, nullptr, 0, block_grad(ws, 11), NF,
                             ws.f(L.nbwdp), d.B, d.H, d.W, 1, st, nullptr, &bst, &pre11));
        } else {
            P4C_TRY(conv_fwd(d.compute, d.dtype, dy, NF, wslot(ws, 2 * NCONV + 1), 1, nullptr, nullptr, 0, block_grad(ws, 11), NF, nullptr,
                             d.B, d.H, d.W, 1, st));
        }
    }
    // ---- decoder (every block's dA goes to that block's own buffer: block_grad)
    Norm nd1 = norm_at(ws, 10, d.B);
    int nxt = 0;
    P4C_TRY(conv_block_bwd(d, ws, 11, ws.act(L.Y[10]), &nd1, block_grad(ws, 10), params, grads, training, st, pre11, &nxt));
    P4C_TRY(conv_block_bwd(d, ws, 10, ws.act(L.S), nullptr, G0, params, grads, training, st, nxt));
    // G0 = dS, kept until the last level

    // ---- encoder levels, deepest first
    // x pass of all four up-sampling adjoints from one read of dS
    void* tx[4];
    {
        int64_t off = 0;
        for (int k = 1; k < NLEV; ++k) { tx[k - 1] = (char*)TB + off * L.esz; off += L.n[0] * NF >> k; }
        P4C_TRY(up_bwd_x4(d.dtype, G0, d.B, d.H, d.W, tx, st));
    }
    void *a = G1, *b = G2;  // a: holds dP_{k+1} on entry (k < 4)
    for (int k = NLEV - 1; k >= 0; --k) {
        const int Hk = L.Hk[k], Wk = L.Wk[k];
        Norm n2 = norm_at(ws, 2 * k + 1, d.B);
        Norm n1 = norm_at(ws, 2 * k, d.B);
        const void* dP = (k + 1 < NLEV) ? a : nullptr;
        // (bf16 storage, statistics of the batch in use: enc_out_bwd also takes pass 1 of conv2's normalisation backward)
        const char* fe = diag_env("P4C_NO_FUSED_REDUCE");
        const bool fuse_off = fe && fe[0] == '1';
        const bool fuse = !fuse_off && d.dtype == P4C_BF16;
        int pre = 0;
        bool finished = false;   // (few slots: enc_out_bwd's last workgroup also finishes the pass -- BwdFin)
        float* part = fuse ? ws.f(L.nbwdp) : nullptr;
        const BwdFin fin = bwd_fin(d, ws, 2 * k + 1, params, grads, training);
        if (k > 0) {
            P4C_TRY(enc_out_bwd(d.dtype, tx[k - 1], d.H, 1 << k, nullptr, dP, ws.act(L.Y[2 * k + 1]), n2.scale, n2.shift, d.B, Hk, Wk,
                                block_grad(ws, 2 * k + 1), st, n2.mean, n2.rstd, part, &pre, &fin, &finished));
        } else {
            P4C_TRY(enc_out_bwd(d.dtype, nullptr, d.H, 1, G0, dP, ws.act(L.Y[1]), n2.scale, n2.shift, d.B, Hk, Wk, block_grad(ws, 1), st,
                                n2.mean, n2.rstd, part, &pre, &fin, &finished));
        }
        // conv2 of the block: input = relu(norm1(Y_k1)); its data gradient is conv1's dA
        int nxt1 = 0;
        P4C_TRY(conv_block_bwd(d, ws, 2 * k + 1, ws.act(L.Y[2 * k]), &n1, block_grad(ws, 2 * k), params, grads, training, st, pre, &nxt1,
                               finished));
        // conv1 of the block: input = P_k (k>0) or x; its data gradient is dP_k (into the other of the two rotating buffers)
        if (k > 0) {
            P4C_TRY(conv_block_bwd(d, ws, 2 * k, ws.act(L.P[k]), nullptr, b, params, grads, training, st, nxt1));
            void* t = a; a = b; b = t;  // dP_k now in a
        } else {
            P4C_TRY(conv_block_bwd(d, ws, 0, x, nullptr, d.dx_channels > 0 ? dx : nullptr, params, grads, training, st, nxt1));
        }
    }
    // join: the caller's stream continues only after every weight gradient of this call has been accumulated -- unless the
    // caller defers that to p4c_side_stream_join (then only what the next call may overwrite is ordered: the caller's dy)
    if (g_side.enabled) {
        // (the batched reduction also reads partials the main stream wrote -- out_conv_bwd: ordered after it even with nothing pending)
        if (g_side.pending.empty()) P4C_TRY(g_side.order(st, g_side.stream));
        P4C_TRY(g_side.flush(st));
        P4C_TRY(wgrad_reduce_batch(reduce_jobs, g_side.stream));
        if (plain_stream) {
            const int set = (int)(g_side.calls & 1);
            P4C_CHECK_HIP(hipEventRecord(g_side.set_done[set], g_side.stream));
            g_side.set_recorded[set] = true;
        }
        if (deferring) {
            if (!fused_out) P4C_CHECK_HIP(hipStreamWaitEvent(st, g_side.dy_read, 0));
        } else
            P4C_TRY(g_side.order(g_side.stream, st));
    } else {
        P4C_TRY(wgrad_reduce_batch(reduce_jobs, st));
    }
    return P4C_OK;
}

// Weight gradients beside the backward chain (on: the default) or on the caller's stream like everything else (off).  For hosts
// that cannot keep two streams fed -- eight ranks sharing one host -- and for the single-stream HIP-graph capture (a two-stream
// capture is replayed over four hardware queues and loses the overlap: profiles/r04_graph_vs_eager.txt).  Call between steps only:
// nothing of a previous step may be in flight on the side stream (p4c_side_stream_join / a stream synchronisation first).
extern "C" int p4c_side_stream_enable(int on) {
    g_side_enable_request.store(on != 0 ? 1 : 0);     // every thread's next p4c_halfunet_backward applies it (the calling thread: now)
    P4C_TRY(g_side.init());
    return apply_side_request();
}

extern "C" long long p4c_side_stream_launch_count(void) { return g_side_launches.load(); }

extern "C" int p4c_side_stream_defer(int on) {
    g_side.defer = on != 0;
    return P4C_OK;
}

extern "C" int p4c_side_stream_join(p4c_stream_t stream) {
    SideStream& sd = g_side;
    if (!sd.enabled || !sd.stream || sd.external) return P4C_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(as_stream(stream), &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone) return P4C_OK;   // (captured calls joined for themselves)
    sd.next = 0;
    return sd.order(sd.stream, as_stream(stream));
}

// ------------------------------------------------------------------------------ single-op entry points (tests, reuse)
extern "C" int p4c_prep_weights(const float* w, int CO, int CI, int ks, int transpose_flip, int M_pad, int K_pad,
                                void* out, int compute, p4c_stream_t stream) {
    P4C_CHECK_ARG(w && out, "p4c_prep_weights: null pointer");
    P4C_CHECK_ARG((ks == 1 || ks == 3) && M_pad % 64 == 0 && K_pad % 32 == 0, "p4c_prep_weights: bad ks / padding");
    P4C_CHECK_ARG(compute == P4C_F32 || compute == P4C_BF16, "p4c_prep_weights: bad compute type");
    return prep_w(compute, w, CO, CI, ks, transpose_flip, M_pad, K_pad, out, as_stream(stream));
}

extern "C" int p4c_conv_stat_tiles(int compute, int storage, int CI, int B, int H, int W) {
    return stat_tiles(compute, storage, CI, B, H, W);
}

extern "C" int p4c_conv_stat_tiles_ks(int compute, int storage, int CI, int ks, int B, int H, int W) {
    return stat_tiles(compute, storage, CI, B, H, W, ks);
}

extern "C" int p4c_conv_kernel_kind(int compute, int storage, int CI, int ks, int B, int H, int W) {
    if (compute != P4C_BF16) return 0;
    if (conv_bf16_is_rows(storage, CI, ks, 1, 64, B, H, W)) return 2;
    return conv_bf16_is_ring(storage, CI, ks, 1, 64, B, H, W) ? 1 : 0;
}

extern "C" int p4c_conv_fwd(const void* in, int compute, int storage, int CI, const void* wprep, int ks,
                            const float* in_scale, const float* in_shift, int in_relu, const float* bias, void* out,
                            int out_cs, float* stat_partial, int B, int H, int W, int m_blocks, p4c_stream_t stream) {
    P4C_CHECK_ARG(in && wprep && out, "p4c_conv_fwd: null pointer");
    P4C_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "p4c_conv_fwd: scale and shift go together");
    P4C_CHECK_ARG(m_blocks >= 1 && out_cs >= 64 * m_blocks && out_cs % 4 == 0, "p4c_conv_fwd: bad out_cs / m_blocks");
    P4C_CHECK_ARG(!stat_partial || m_blocks == 1, "p4c_conv_fwd: statistics need m_blocks == 1");
    P4C_CHECK_ARG(compute == P4C_F32 || compute == P4C_BF16, "p4c_conv_fwd: bad compute type %d", compute);
    P4C_CHECK_ARG(storage == P4C_F32 || (storage == P4C_BF16 && compute == P4C_BF16), "p4c_conv_fwd: bad storage type");
    if (compute == P4C_BF16 && bias) return fail(P4C_ERR_UNSUPPORTED, "p4c_conv_fwd: bias is not implemented for P4C_BF16");
    if (compute == P4C_BF16)
        return conv_fwd_bf16(in, storage, CI, wprep, ks, in_scale, in_shift, in_relu, out, out_cs, stat_partial, B, H, W,
                             m_blocks, as_stream(stream));
    return conv_fwd_f32((const float*)in, CI, (const float*)wprep, ks, in_scale, in_shift, in_relu, bias, (float*)out, out_cs,
                        stat_partial, B, H, W, m_blocks, as_stream(stream));
}

extern "C" size_t p4c_conv_wgrad_workspace_bytes(int CI_pad, int ks) {
    return (size_t)wgrad_partial_floats(CI_pad, ks, num_cus()) * sizeof(float);
}

extern "C" int p4c_conv_wgrad(const void* in, int compute, int storage, int CI_pad, int ks, const float* in_scale,
                              const float* in_shift, int in_relu, const void* dout, int CO, int CI, float* grad,
                              void* workspace, int B, int H, int W, p4c_stream_t stream) {
    P4C_CHECK_ARG(in && dout && grad && workspace, "p4c_conv_wgrad: null pointer");
    P4C_CHECK_ARG(CO <= 64 && CI <= CI_pad, "p4c_conv_wgrad: CO must be <= 64 and CI <= CI_pad");
    P4C_CHECK_ARG(compute == P4C_F32 || compute == P4C_BF16, "p4c_conv_wgrad: bad compute type");
    P4C_CHECK_ARG(storage == P4C_F32 || (storage == P4C_BF16 && compute == P4C_BF16), "p4c_conv_wgrad: bad storage type");
    int64_t ntiles = (int64_t)B * conv_tiles_per_sample(H, W);
    const int G = ntiles < num_cus() ? (int)ntiles : num_cus();
    return conv_wgrad(compute, storage, in, CI_pad, ks, in_scale, in_shift, in_relu, dout, (float*)workspace, G, B, H, W, CO,
                      CI, grad, as_stream(stream));
}

extern "C" int p4c_conv_wgrad_nb(const void* in, const float* in_scale, const float* in_shift, int in_relu, const void* dA, const void* y,
                                 const float* gamma, const float* nscale, const float* nshift, const float* rstd, const float* mean,
                                 const float* k1, const float* k2, int CO, int CI, float* grad, void* workspace, int B, int H, int W,
                                 p4c_stream_t stream) {
    P4C_CHECK_ARG(in && dA && y && gamma && nscale && nshift && rstd && mean && k1 && k2 && grad && workspace, "p4c_conv_wgrad_nb: null pointer");
    P4C_CHECK_ARG(CO <= 64 && CI <= 64, "p4c_conv_wgrad_nb: CO and CI must be <= 64");
    P4C_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "p4c_conv_wgrad_nb: scale and shift go together");
    if (!conv_wgrad_bf16_takes_nb(P4C_BF16, 64, 3, B)) return fail(P4C_ERR_UNSUPPORTED, "p4c_conv_wgrad_nb: unsupported batch size %d", B);
    int64_t ntiles = (int64_t)B * conv_tiles_per_sample(H, W);
    const int G = ntiles < num_cus() ? (int)ntiles : num_cus();
    const NormBwdCoef nb{y, gamma, nscale, nshift, rstd, mean, k1, k2};
    return conv_wgrad(P4C_BF16, P4C_BF16, in, 64, 3, in_scale, in_shift, in_relu, dA, (float*)workspace, G, B, H, W, CO, CI, grad,
                      as_stream(stream), &nb);
}

extern "C" int p4c_out_conv_bwd_slots(int B, int64_t N) { return out_conv_bwd_slots(B, N); }

extern "C" size_t p4c_out_conv_bwd_workspace_bytes(int B, int64_t N) {
    return (size_t)B * out_conv_bwd_slots(B, N) * 4096 * sizeof(float);
}

extern "C" int p4c_out_conv_bwd(const void* dy, const void* wprep_dgrad, const void* y, const float* scale, const float* shift,
                                const float* mean, const float* rstd, void* dA, float* stat_partial, int CO, float* grad_w, void* workspace,
                                int B, int64_t N, p4c_stream_t stream) {
    P4C_CHECK_ARG(dy && wprep_dgrad && y && scale && shift && mean && rstd && dA && stat_partial && grad_w && workspace,
                  "p4c_out_conv_bwd: null pointer");
    P4C_CHECK_ARG(CO > 0 && CO <= 64, "p4c_out_conv_bwd: CO must be in 1..64");
    if (!out_conv_bwd_ok(P4C_BF16, B, N)) return fail(P4C_ERR_UNSUPPORTED, "p4c_out_conv_bwd: unsupported shape (B %d, N %lld)", B, (long long)N);
    int nblk = 0;
    P4C_TRY(launch_out_conv_bwd(dy, wprep_dgrad, y, scale, shift, mean, rstd, dA, stat_partial, (float*)workspace, B, N, as_stream(stream), &nblk));
    return wgrad_reduce((const float*)workspace, B * nblk, 1, 64, 0, 64, CO, 64, grad_w, as_stream(stream));
}

extern "C" int p4c_upsample_sum_bwd_x(int storage, const void* dS, int B, int H, int W, void* tx1, void* tx2, void* tx3, void* tx4,
                                     p4c_stream_t stream) {
    P4C_CHECK_ARG(dS && tx1 && tx2 && tx3 && tx4, "p4c_upsample_sum_bwd_x: null pointer");
    P4C_CHECK_ARG(storage == P4C_F32 || storage == P4C_BF16, "p4c_upsample_sum_bwd_x: storage must be P4C_F32 or P4C_BF16");
    P4C_CHECK_ARG(B > 0 && H > 0 && W >= 16 && W % 16 == 0, "p4c_upsample_sum_bwd_x: W must be a positive multiple of 16");
    void* tx[4] = {tx1, tx2, tx3, tx4};
    return up_bwd_x4(storage, dS, B, H, W, tx, as_stream(stream));
}

extern "C" int p4c_conv_wgrad_kernel_kind(int storage, int B, int H, int W) {
    int64_t ntiles = (int64_t)B * conv_tiles_per_sample(H, W);
    const int G = ntiles < num_cus() ? (int)ntiles : num_cus();
    return conv_wgrad_rows_ok(storage, 64, 0, 64, 3, G, B, H, W) ? 1 : 0;
}

// ---- plain bf16 convolution on feature maps with fewer than 64 channels, in place (row kernel; SwinUNetR's 24- / 48-channel levels)
extern "C" int p4c_conv_compact_supported(int in_c, int out_c, int ks, int B, int H, int W) {
    return (conv_rows_compact_ok(P4C_BF16, in_c, out_c, ks, B, H, W) && B <= 32) ? 1 : 0;
}

extern "C" int p4c_conv_fwd_compact(const void* in, int in_c, const void* wprep, int ks, void* out, int out_c, int B, int H, int W,
                                    p4c_stream_t stream) {
    P4C_CHECK_ARG(in && wprep && out, "p4c_conv_fwd_compact: null pointer");
    return launch_conv_bf16_rows_compact(in, in_c, wprep, ks, out, out_c, B, H, W, as_stream(stream));
}

extern "C" int p4c_conv_wgrad_compact(const void* in, int in_c, int ks, const void* dout, int dout_c, int CO, int CI, float* grad,
                                      void* workspace, int B, int H, int W, p4c_stream_t stream) {
    P4C_CHECK_ARG(in && dout && grad && workspace, "p4c_conv_wgrad_compact: null pointer");
    P4C_CHECK_ARG(CO <= dout_c && CI <= in_c, "p4c_conv_wgrad_compact: CO <= dout_c and CI <= in_c");
    int64_t ntiles = (int64_t)B * conv_tiles_per_sample(H, W);
    const int G = ntiles < num_cus() ? (int)ntiles : num_cus();
    return conv_wgrad_bf16_compact(in, in_c, ks, dout, dout_c, (float*)workspace, G, B, H, W, CO, CI, grad, as_stream(stream));
}
