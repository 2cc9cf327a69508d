// The encoder's heads and the latent gate of the model.py flavour, train mode, as ONE launch forward and ONE backward.
//
//   log_var  = log(softplus(v))                                   /root/reference/train/model.py:54-55   (v = variance_estimator(x))
//   s1       = selection_layer1(mean)   (ld -> 1, per token)      model.py:56-57
//   logits   = selection_layer2(s1) + 1 (hw -> 1, per frame)      model.py:58
//   sel      = round(sigmoid(logits + log(u / (1 - u))))          /root/reference/train/layers.py:238-252 (temperature 1, round_ste :226-236)
//   z        = mean + eps * exp(log_var / 2)                      model.py:124-128
//   KL       = mean_{t,hw,c}[0.5 (e^lv - 1 - lv + mu^2) m_t / len]  /root/reference/train/rl_nonadversarial.py:146-147 (per frame here)
//   comp     = fill * (1 - sel) + z * sel                         model.py:133
//
// As framework ops this is ~17 launches forward and ~28 backward over 6 MB tensors and (b, t) scalars; every launch of a replayed
// hipGraph costs ~5 us however little it does.  One workgroup per frame: the selection of a frame needs all of its tokens (phase A),
// the gate of every token needs the selection (phase B); the frame (48 KB per operand at hw = 256, ld = 96) is re-read from L2.
// The rounding points of the unfused path are kept (bf16 Linear outputs, bf16 softplus / log results) so that the discrete selection
// and the values downstream agree with it; gradients are accumulated in fp32 and rounded once.
// Parameter gradients leave as one partial row per frame (folded by the caller's grouped fold): no atomics, fixed order.
#include "common.hpp"

namespace {

constexpr int EH_MAX_THREADS = 1024;
constexpr int EH_MAX_HW = 4096;              // w2 / s1 rows in LDS
constexpr int EH_MAX_LD = 1024;

struct EhDims { int T, HW, LD, CG, TY; long mask_pitch; int dbg; };

int g_eh_debug = 0;        // vvae_encoder_head_debug: attribution builds of the backward's rounding points (tests / tools only)

__device__ __forceinline__ float bfr(float v) { return bf2f(f2bf(v)); }

// softplus as the framework computes it (beta 1, threshold 20), in the compute dtype: result rounded to bf16.  Hardware exp / log
// (v_exp_f32 / v_log_f32, ~1e-6 relative) instead of the correctly rounded library calls: the result is rounded to 8 bits anyway, and
// the library forms made these kernels VALU-bound (50 us each for 6 MB of operands; ~300 instructions per element).  log(1 + e) loses
// the small e in the addition, so below v = -4 (e < 0.018) the series e - e^2/2 + e^3/3 stands in for log1p (relative error e^3/4 < 2e-6).
__device__ __forceinline__ float softplus_bf(float v)
{
    if (v > 20.f) return bfr(v);
    const float e = __expf(v);
    return bfr(v < -4.f ? e * (1.f - e * (0.5f - e * (1.f / 3.f))) : __logf(1.f + e));
}

// sum over the workgroup in fixed order (lanes by butterfly, waves by index) -> every thread gets the total
__device__ __forceinline__ float block_total(float v, float* red, float* out_slot)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) t += red[i];
        *out_slot = t;
    }
    __syncthreads();
    return *out_slot;
}

__device__ __forceinline__ float seq_len(const float* __restrict__ mrow, int T)
{
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += mrow[t];
    return fmaxf(s, 1.0f);
}

// The rl flavour (RL = true; reference train/rl_model.py:50-60,119-147): the selection is the PROBABILITY sigmoid(logits) (no Gumbel noise, no
// rounding), every clip is doubled into a pair (samples 2k, 2k + 1) whose members draw their own Bernoulli frame mask u < probability and
// gate the shared latent with it; log-variance and mean are returned pair-doubled as well.  ``u``: (2B, T) uniforms; ``logvar`` / ``comp``: (2B, T,
// HW, LD); ``mean2`` likewise; ``mask2`` (2B, T) the sampled frame masks; ``sel_out`` (2B, T) the pair-doubled probability; ``y_out`` the logits; ``kl_frame`` (B, T).
struct EhRl { bf16_t* mean2; float* mask2; };

// LDS: w2b[HW] | pt[HW * CG] | red[16] | slot[4]
template <bool RL>
__global__ __launch_bounds__(EH_MAX_THREADS) void encoder_head_fwd_kernel(
    const bf16_t* __restrict__ mean, const bf16_t* __restrict__ v, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ u, const float* __restrict__ eps,
    const float* __restrict__ mask, const float* __restrict__ fill, bf16_t* __restrict__ logvar, bf16_t* __restrict__ comp,
    float* __restrict__ sel_out, float* __restrict__ y_out, float* __restrict__ s1_out, float* __restrict__ kl_frame, EhDims d, EhRl rl)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w2b = lds;
    float* pt = w2b + d.HW;
    float* red = pt + (long)d.HW * d.CG;
    float* slot = red + 16;
    const int f = blockIdx.x, b = f / d.T, t = f % d.T;
    const int tid = threadIdx.x, cg = tid % d.CG, ty = tid / d.CG;
    const bool active = ty < d.TY;
    const long base = (long)f * d.HW * d.LD + cg * 8;

    for (int j = tid; j < d.HW; j += blockDim.x) w2b[j] = bfr(w2[j]);
    float w1r[8], fr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { w1r[e] = bfr(w1[cg * 8 + e]); fr[e] = bfr(fill[cg * 8 + e]); }

    // ---- phase A: per-token dot with w1 (partial per channel group, summed per token in channel order) ----
    if (active)
        for (int j = ty; j < d.HW; j += d.TY) {
            float m[8];
            VecIO<bf16_t, 8>::load(mean + base + (long)j * d.LD, m);
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a += m[e] * w1r[e];
            pt[j * d.CG + cg] = a;
        }
    __syncthreads();
    const float b1b = bfr(b1[0]);
    float acc = 0.f;
    for (int j = tid; j < d.HW; j += blockDim.x) {
        float a = 0.f;
        for (int c = 0; c < d.CG; ++c) a += pt[j * d.CG + c];
        const float s = bfr(a + b1b);
        s1_out[(long)f * d.HW + j] = s;
        acc += s * w2b[j];
    }
    const float dot2 = block_total(acc, red, slot);
    const float logits = bfr(bfr(dot2 + bfr(b2[0])) + 1.f);
    float y, sel;
    bool keep2[2] = {false, false};
    if (RL) {
        y = logits;
        sel = bfr(1.f / (1.f + expf(-y)));                    // the probability, an array of the compute dtype (rl_model.py:59)
#pragma unroll
        for (int p = 0; p < 2; ++p) keep2[p] = u[(long)(2 * b + p) * d.T + t] < sel;          // rl_model.py:141-142
        if (tid < 2) rl.mask2[(long)(2 * b + tid) * d.T + t] = keep2[tid] ? 1.f : 0.f;
    } else {
        float uc = u[f];
        uc = uc < 1e-20f ? 1e-20f : (uc > 1.f - 1e-20f ? 1.f - 1e-20f : uc);
        y = logits + logf(uc / (1.f - uc));
        sel = rintf(1.f / (1.f + expf(-y)));
    }
    if (tid == 0) {
        y_out[f] = y;
        if (RL) { sel_out[(long)(2 * b) * d.T + t] = sel; sel_out[(long)(2 * b + 1) * d.T + t] = sel; }      // the pair-doubled probability
        else sel_out[f] = sel;
    }

    // ---- phase B: log-variance, reparameterisation, gate, KL ----
    float kl = 0.f;
    if (active)
        for (int j = ty; j < d.HW; j += d.TY) {
            const long g = base + (long)j * d.LD;
            float m[8], vv[8], lv[8], o[8];
            VecIO<bf16_t, 8>::load(mean + g, m);
            VecIO<bf16_t, 8>::load(v + g, vv);
            const float4 e0 = *reinterpret_cast<const float4*>(eps + g), e1 = *reinterpret_cast<const float4*>(eps + g + 4);
            const float ee[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
            float zz[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                lv[e] = bfr(__logf(softplus_bf(vv[e])));
                zz[e] = m[e] + ee[e] * __expf(0.5f * lv[e]);
                o[e] = sel != 0.f ? zz[e] : fr[e];
                kl += 0.5f * (__expf(lv[e]) - 1.f - lv[e] + m[e] * m[e]);
            }
            if (RL) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {                  // the pair's rows: frame (2b + p, t) of the doubled batch
                    const long g2 = ((long)(2 * b + p) * d.T + t) * d.HW * d.LD + cg * 8 + (long)j * d.LD;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = keep2[p] ? zz[e] : fr[e];
                    VecIO<bf16_t, 8>::store(logvar + g2, lv);
                    VecIO<bf16_t, 8>::store(rl.mean2 + g2, m);
                    VecIO<bf16_t, 8>::store(comp + g2, o);
                }
            } else {
                VecIO<bf16_t, 8>::store(logvar + g, lv);
                VecIO<bf16_t, 8>::store(comp + g, o);
            }
        }
    const float klt = block_total(kl, red, slot);
    if (tid == 0) {
        const float* mrow = mask + (long)b * d.mask_pitch;
        const float kf = klt * mrow[t] / (seq_len(mrow, d.T) * (float)d.T * (float)d.HW * (float)d.LD);
        if (RL) {                                             // both members of a pair share mean / log-variance / mask: the same KL share
            kl_frame[(long)(2 * b) * d.T + t] = kf;
            kl_frame[(long)(2 * b + 1) * d.T + t] = kf;
        } else kl_frame[f] = kf;
    }
}

// LDS: w2b[HW] | colred[TY * LD] | red[16] | slot[4]
// part1 (F, LD) = dW1, part2 (F, HW) = dW2, part3 (F, LD) = d fill, partb (2, F, 4) = [db1 0 0 0] then [db2 0 0 0]: this frame's row of each
// (row widths are multiples of four floats: what the caller's fold kernels take).
// RL: ``logvar`` is the pair-doubled forward output (row 2b is read), ``dcomp`` (2B, T, HW, LD) and ``sel_in`` = the frame masks (2B, T): the latent's
// gradient is the sum of the pair members that kept the frame, the fill token's the sum of those that dropped it; no gradient flows through the
// sampled mask (rl_model.py:141-144); ``dsel`` (2B, T): the gradient at the pair-doubled probability (its two rows add up); ``y_in`` the logits;
// ``gkl`` as in the model flavour, (B, T).
template <bool RL>
__global__ __launch_bounds__(EH_MAX_THREADS) void encoder_head_bwd_kernel(
    const bf16_t* __restrict__ mean, const bf16_t* __restrict__ v, const bf16_t* __restrict__ logvar, const float* __restrict__ eps,
    const float* __restrict__ mask, const float* __restrict__ fill, const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ y_in,
    const float* __restrict__ s1_in, const float* __restrict__ sel_in, const bf16_t* __restrict__ dcomp, const float* __restrict__ dsel,
    const float* __restrict__ gkl, long gkl_pitch_b, long gkl_pitch_t, const bf16_t* __restrict__ dlv_ext, bf16_t* __restrict__ dmean, bf16_t* __restrict__ dv,
    float* __restrict__ part1, float* __restrict__ part2, float* __restrict__ part3, float* __restrict__ partb, EhDims d)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w2b = lds;
    float* colred = w2b + d.HW;
    float* red = colred + (long)d.TY * d.LD;
    float* slot = red + 16;
    const int f = blockIdx.x, b = f / d.T, t = f % d.T;
    const int tid = threadIdx.x, cg = tid % d.CG, ty = tid / d.CG;
    const bool active = ty < d.TY;
    const long base = (long)f * d.HW * d.LD + cg * 8;
    const long f2 = (long)(2 * b) * d.T + t;                     // RL: frame (2b, t) of the doubled batch; its pair partner is d.T frames further
    const long base2 = f2 * d.HW * d.LD + cg * 8, pair = (long)d.T * d.HW * d.LD;
    const float sel = RL ? 0.f : sel_in[f];
    const bool keep = sel != 0.f;
    const bool k0 = RL && sel_in[f2] != 0.f, k1 = RL && sel_in[f2 + d.T] != 0.f;
    const long lvbase = RL ? base2 : base;                       // where this frame's log-variance lies

    for (int j = tid; j < d.HW; j += blockDim.x) w2b[j] = bfr(w2[j]);
    float w1r[8], fr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { w1r[e] = bfr(w1[cg * 8 + e]); fr[e] = bfr(fill[cg * 8 + e]); }

    // ---- phase 1: d sel through the gate = sum dcomp * (z - fill); d fill = sum_tokens dcomp * (1 - sel) ----
    float ds = 0.f, dfa[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dfa[e] = 0.f;
    if (!RL && active && dcomp)
        for (int j = ty; j < d.HW; j += d.TY) {
            const long g = base + (long)j * d.LD;
            float m[8], lvv[8], dc[8];
            VecIO<bf16_t, 8>::load(mean + g, m);
            VecIO<bf16_t, 8>::load(logvar + g, lvv);
            VecIO<bf16_t, 8>::load(dcomp + g, dc);
            const float4 e0 = *reinterpret_cast<const float4*>(eps + g), e1 = *reinterpret_cast<const float4*>(eps + g + 4);
            const float ee[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float z = m[e] + ee[e] * __expf(0.5f * lvv[e]);
                ds += dc[e] * (z - fr[e]);
                if (!keep) dfa[e] += dc[e];
            }
        }
    const float dsg = block_total(ds, red, slot);
    // round_ste: identity; sigmoid'; the .float() of the logits casts the gradient back to the compute dtype
    const float yy = y_in[f];
    const float sg = 1.f / (1.f + expf(-yy));
    const float dl_f = (dsg + (dsel ? dsel[f] : 0.f)) * sg * (1.f - sg);
    const float dl = RL ? bfr((dsel ? dsel[f2] + dsel[f2 + d.T] : 0.f) * sg * (1.f - sg)) : ((d.dbg & 1) ? dl_f : bfr(dl_f));

    // ---- phase 2: token / element gradients ----
    const float* mrow = mask + (long)b * d.mask_pitch;
    float kscale = 0.f;
    if (gkl) {
        const float gk = RL ? gkl[(long)(2 * b) * gkl_pitch_b + (long)t * gkl_pitch_t] + gkl[(long)(2 * b + 1) * gkl_pitch_b + (long)t * gkl_pitch_t]
                            : gkl[(long)b * gkl_pitch_b + (long)t * gkl_pitch_t];
        kscale = gk * mrow[t] / (seq_len(mrow, d.T) * (float)d.T * (float)d.HW * (float)d.LD);
    }
    float dw1[8], db1 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dw1[e] = 0.f;
    if (active)
        for (int j = ty; j < d.HW; j += d.TY) {
            const long g = base + (long)j * d.LD;
            const float dsi = (d.dbg & 2) ? dl * w2b[j] : bfr(dl * w2b[j]);
            float m[8], vv[8], lvv[8], dc[8], dm[8], dvv[8], dx[8];
            VecIO<bf16_t, 8>::load(mean + g, m);
            VecIO<bf16_t, 8>::load(v + g, vv);
            VecIO<bf16_t, 8>::load(logvar + lvbase + (long)j * d.LD, lvv);
            if (RL) {
                float d0[8], d1[8];
                if (dcomp) {
                    VecIO<bf16_t, 8>::load(dcomp + base2 + (long)j * d.LD, d0);
                    VecIO<bf16_t, 8>::load(dcomp + base2 + pair + (long)j * d.LD, d1);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { d0[e] = 0.f; d1[e] = 0.f; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    dc[e] = (k0 ? d0[e] : 0.f) + (k1 ? d1[e] : 0.f);
                    dfa[e] += (k0 ? 0.f : d0[e]) + (k1 ? 0.f : d1[e]);
                }
            } else if (dcomp && keep) VecIO<bf16_t, 8>::load(dcomp + g, dc);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) dc[e] = 0.f;
            }
            if (dlv_ext) VecIO<bf16_t, 8>::load(dlv_ext + g, dx);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) dx[e] = 0.f;
            }
            const float4 e0 = *reinterpret_cast<const float4*>(eps + g), e1 = *reinterpret_cast<const float4*>(eps + g + 4);
            const float ee[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float var = softplus_bf(vv[e]);
                const float lv = lvv[e];
                const float elv = __expf(lv);
                dm[e] = dc[e] + kscale * m[e] + dsi * w1r[e];
                const float dlv = dc[e] * ee[e] * 0.5f * __expf(0.5f * lv) + kscale * 0.5f * (elv - 1.f) + dx[e];
                const float dsp = vv[e] > 20.f ? 1.f : __builtin_amdgcn_rcpf(1.f + __expf(-vv[e]));          // softplus'
                dvv[e] = dlv * __builtin_amdgcn_rcpf(var) * dsp;
                dw1[e] += m[e] * dsi;
            }
            VecIO<bf16_t, 8>::store(dmean + g, dm);
            VecIO<bf16_t, 8>::store(dv + g, dvv);
            if (cg == 0) {
                db1 += dsi;
                part2[(long)f * d.HW + j] = s1_in[(long)f * d.HW + j] * dl;
            }
        }
    const float db1t = block_total(db1, red, slot);
    if (tid < 8) {
        const long F = gridDim.x;
        partb[(tid < 4 ? 0 : F * 4) + (long)f * 4 + (tid & 3)] = (tid & 3) ? 0.f : (tid < 4 ? db1t : dl);
    }
    // per-channel sums over the token lanes, fixed order: dW1 then d fill through the same LDS image
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (active) {
#pragma unroll
            for (int e = 0; e < 8; ++e) colred[ty * d.LD + cg * 8 + e] = pass ? dfa[e] : dw1[e];
        }
        __syncthreads();
        for (int c = tid; c < d.LD; c += blockDim.x) {
            float a = 0.f;
            for (int r = 0; r < d.TY; ++r) a += colred[r * d.LD + c];
            if (pass) part3[(long)f * d.LD + c] = a;
            else part1[(long)f * d.LD + c] = a;
        }
        __syncthreads();
    }
}

bool eh_dims(int B, int T, int HW, int LD, long mask_pitch, EhDims& d, int& threads)
{
    if (B <= 0 || T <= 0 || HW <= 0 || HW > EH_MAX_HW || HW % 4 || LD <= 0 || LD > EH_MAX_LD || LD % 8 || (mask_pitch != 0 && mask_pitch < T)) return false;
    d.T = T; d.HW = HW; d.LD = LD; d.CG = LD / 8; d.mask_pitch = mask_pitch; d.dbg = g_eh_debug;
    d.TY = EH_MAX_THREADS / d.CG;
    if (d.TY > HW) d.TY = HW;
    threads = ((d.CG * d.TY + 63) / 64) * 64;
    return true;
}

}  // namespace

// 1 when the fused heads cover the shape (bf16 operands are the caller's business), else 0.
extern "C" int vvae_encoder_head_ok(int B, int T, int HW, int LD)
{
    EhDims d; int th;
    if (!eh_dims(B, T, HW, LD, T, d, th)) return 0;
    const size_t fwd = ((size_t)HW + (size_t)HW * d.CG + 20) * 4, bwd = ((size_t)HW + (size_t)d.TY * LD + 20) * 4;
    return fwd <= 64 * 1024 && bwd <= 64 * 1024;
}

// Test / attribution hook (tools/r04_sel1_attribution.py): which of the backward kernel's chosen bf16 rounding points are SKIPPED (kept fp32).
// bit 0: d logits (the gradient handed to selection_layer2, a bf16 array in the reference's mixed-precision run); bit 1: d s1 = d logits * w2
// (the gradient handed to selection_layer1).  0 (default) = the shipped kernel.
extern "C" int vvae_encoder_head_debug(int flags)
{
    if (flags < 0 || flags > 3) return VVAE_ERR_BAD_ARG;
    g_eh_debug = flags;
    return 0;
}

// mean, v (pre-softplus) bf16 (B, T, HW, LD) contiguous; w1 (LD), b1 (1), w2 (HW), b2 (1), fill (LD) fp32 masters; u fp32 (B*T) uniform;
// eps fp32 (B, T, HW, LD) normal; mask fp32 rows of T with pitch mask_pitch (elements) per sample (0: one row for all samples).  HW % 4 == 0, LD % 8 == 0.
// -> logvar, comp bf16 (B, T, HW, LD); sel, y fp32 (B*T) (y = the noisy logit, kept for the backward); s1 fp32 (B*T, HW);
//    kl_frame fp32 (B*T): summed over a sample's frames it is the per-sample KL term.
extern "C" int vvae_encoder_head_fwd(const void* mean, const void* v, const float* w1, const float* b1, const float* w2, const float* b2,
                                     const float* u, const float* eps, const float* mask, long mask_pitch, const float* fill, void* logvar,
                                     void* comp, float* sel, float* y, float* s1, float* kl_frame, int B, int T, int HW, int LD, void* stream)
{
    EhDims d; int threads;
    if (!mean || !v || !w1 || !b1 || !w2 || !b2 || !u || !eps || !mask || !fill || !logvar || !comp || !sel || !y || !s1 || !kl_frame ||
        !eh_dims(B, T, HW, LD, mask_pitch, d, threads) || !vvae_encoder_head_ok(B, T, HW, LD) ||
        ((uintptr_t)mean | (uintptr_t)v | (uintptr_t)eps | (uintptr_t)logvar | (uintptr_t)comp) % 16) return VVAE_ERR_BAD_ARG;
    const size_t lds = ((size_t)HW + (size_t)HW * d.CG + 20) * 4;
    hipLaunchKernelGGL(encoder_head_fwd_kernel<false>, dim3(B * T), dim3(threads), lds, (hipStream_t)stream, (const bf16_t*)mean, (const bf16_t*)v,
                       w1, b1, w2, b2, u, eps, mask, fill, (bf16_t*)logvar, (bf16_t*)comp, sel, y, s1, kl_frame, d, EhRl{nullptr, nullptr});
    VVAE_LAUNCH_CHECK();
    return 0;
}

// logvar: the forward's output.  dcomp bf16 (B, T, HW, LD) or NULL; dsel fp32 (B*T) or NULL; gkl fp32 or NULL: the gradient of kl_frame[b][t] at gkl[b * gkl_pitch_b + t * gkl_pitch_t]
// (pitches (1, 0): one gradient per sample, as the loss tail hands it over); dlv_ext bf16 (B, T, HW, LD) or NULL (a gradient arriving at logvar from elsewhere).
// -> dmean, dv bf16 (B, T, HW, LD); part1 (B*T, LD) = dW1, part2 (B*T, HW) = dW2, part3 (B*T, LD) = d fill, partb (2, B*T, 4) = [db1 0 0 0] rows then
// [db2 0 0 0] rows: one partial row per frame each, to be summed over rows by the caller.
extern "C" int vvae_encoder_head_bwd(const void* mean, const void* v, const void* logvar, const float* eps, const float* mask, long mask_pitch, const float* fill,
                                     const float* w1, const float* w2, const float* y, const float* s1, const float* sel, const void* dcomp,
                                     const float* dsel, const float* gkl, long gkl_pitch_b, long gkl_pitch_t, const void* dlv_ext, void* dmean, void* dv,
                                     float* part1, float* part2, float* part3, float* partb, int B, int T, int HW, int LD, void* stream)
{
    EhDims d; int threads;
    if (!mean || !v || !logvar || !eps || !mask || !fill || !w1 || !w2 || !y || !s1 || !sel || !dmean || !dv || !part1 || !part2 || !part3 || !partb ||
        !eh_dims(B, T, HW, LD, mask_pitch, d, threads) || !vvae_encoder_head_ok(B, T, HW, LD) ||
        ((uintptr_t)mean | (uintptr_t)v | (uintptr_t)logvar | (uintptr_t)eps | (uintptr_t)dcomp | (uintptr_t)dlv_ext | (uintptr_t)dmean | (uintptr_t)dv) % 16)
        return VVAE_ERR_BAD_ARG;
    const size_t lds = ((size_t)HW + (size_t)d.TY * LD + 20) * 4;
    hipLaunchKernelGGL(encoder_head_bwd_kernel<false>, dim3(B * T), dim3(threads), lds, (hipStream_t)stream, (const bf16_t*)mean, (const bf16_t*)v,
                       (const bf16_t*)logvar, eps, mask, fill, w1, w2, y, s1, sel, (const bf16_t*)dcomp, dsel, gkl, gkl_pitch_b, gkl_pitch_t, (const bf16_t*)dlv_ext, (bf16_t*)dmean,
                       (bf16_t*)dv, part1, part2, part3, partb, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// The rl flavour's heads, reparameterisation, KL, pair doubling, Bernoulli frame masks and latent gate in one launch each way (reference
// train/rl_model.py:50-60,119-147).  mean, v bf16 (B, T, HW, LD); u fp32 (2B, T) uniforms; eps fp32 (B, T, HW, LD); mask rows of T per CLIP.
// -> logvar2, mean2, comp2 bf16 (2B, T, HW, LD) (samples 2k, 2k + 1 are a pair); prob2 fp32 (2B T) = sigmoid(logits) rounded to bf16, pair-doubled; mask2 fp32
//    (2B T) = u < prob; y fp32 (B T) the logits and s1 fp32 (B T, HW) (kept for backward); kl_frame2 fp32 (2B, T): summed over a row's frames it is the
//    per-sample KL term of that member of the doubled batch.
extern "C" int vvae_encoder_head_rl_fwd(const void* mean, const void* v, const float* w1, const float* b1, const float* w2, const float* b2,
                                        const float* u2, const float* eps, const float* mask, long mask_pitch, const float* fill, void* logvar2,
                                        void* mean2, void* comp2, float* prob, float* mask2, float* y, float* s1, float* kl_frame2, int B, int T, int HW,
                                        int LD, void* stream)
{
    EhDims d; int threads;
    if (!mean || !v || !w1 || !b1 || !w2 || !b2 || !u2 || !eps || !mask || !fill || !logvar2 || !mean2 || !comp2 || !prob || !mask2 || !y || !s1 ||
        !kl_frame2 || !eh_dims(B, T, HW, LD, mask_pitch, d, threads) || !vvae_encoder_head_ok(B, T, HW, LD) ||
        ((uintptr_t)mean | (uintptr_t)v | (uintptr_t)eps | (uintptr_t)logvar2 | (uintptr_t)mean2 | (uintptr_t)comp2) % 16) return VVAE_ERR_BAD_ARG;
    const size_t lds = ((size_t)HW + (size_t)HW * d.CG + 20) * 4;
    hipLaunchKernelGGL(encoder_head_fwd_kernel<true>, dim3(B * T), dim3(threads), lds, (hipStream_t)stream, (const bf16_t*)mean, (const bf16_t*)v,
                       w1, b1, w2, b2, u2, eps, mask, fill, (bf16_t*)logvar2, (bf16_t*)comp2, prob, y, s1, kl_frame2, d, EhRl{(bf16_t*)mean2, mask2});
    VVAE_LAUNCH_CHECK();
    return 0;
}

// logvar2: the forward's pair-doubled output; mask2 (2B T) its frame masks; dcomp2 bf16 (2B, T, HW, LD) or NULL; dprob2 fp32 (2B T) or NULL: the gradient
// at the pair-doubled probability; gkl fp32 or NULL: the gradient of kl_frame2[i][t] at gkl[i * gkl_pitch_b + t * gkl_pitch_t], i < 2B; dlv_ext bf16
// (B, T, HW, LD) or NULL.  -> dmean, dv bf16 (B, T, HW, LD) and the partial rows of vvae_encoder_head_bwd (one per frame of the B T).
extern "C" int vvae_encoder_head_rl_bwd(const void* mean, const void* v, const void* logvar2, const float* eps, const float* mask, long mask_pitch,
                                        const float* fill, const float* w1, const float* w2, const float* y, const float* s1, const float* mask2,
                                        const void* dcomp2, const float* dprob2, const float* gkl, long gkl_pitch_b, long gkl_pitch_t, const void* dlv_ext,
                                        void* dmean, void* dv, float* part1, float* part2, float* part3, float* partb, int B, int T, int HW, int LD,
                                        void* stream)
{
    EhDims d; int threads;
    if (!mean || !v || !logvar2 || !eps || !mask || !fill || !w1 || !w2 || !y || !s1 || !mask2 || !dmean || !dv || !part1 || !part2 || !part3 || !partb ||
        !eh_dims(B, T, HW, LD, mask_pitch, d, threads) || !vvae_encoder_head_ok(B, T, HW, LD) ||
        ((uintptr_t)mean | (uintptr_t)v | (uintptr_t)logvar2 | (uintptr_t)eps | (uintptr_t)dcomp2 | (uintptr_t)dlv_ext | (uintptr_t)dmean | (uintptr_t)dv) % 16)
        return VVAE_ERR_BAD_ARG;
    const size_t lds = ((size_t)HW + (size_t)d.TY * LD + 20) * 4;
    hipLaunchKernelGGL(encoder_head_bwd_kernel<true>, dim3(B * T), dim3(threads), lds, (hipStream_t)stream, (const bf16_t*)mean, (const bf16_t*)v,
                       (const bf16_t*)logvar2, eps, mask, fill, w1, w2, y, s1, mask2, (const bf16_t*)dcomp2, dprob2, gkl, gkl_pitch_b, gkl_pitch_t,
                       (const bf16_t*)dlv_ext, (bf16_t*)dmean, (bf16_t*)dv, part1, part2, part3, partb, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// ---- the rl flavour's latent gate (reference train/rl_model.py:136-145): every clip is doubled into a pair (samples 2k, 2k + 1 share z), each
// member draws its own Bernoulli frame mask from the selection probabilities, and comp = fill (1 - mask) + z mask.  As framework ops:
// repeat_interleave of the fp32 latent, a comparison, a cast and four elementwise launches over (2b, t, hw, ld) fp32 tensors (50 MB each).
//   fwd: mask[i][t] = u[i][t] < prob[i / 2][t];  comp[i][t][tok][c] = mask ? z[i / 2][t][tok][c] : fill[c]      (bf16, what the decoder reads)
//   bwd: dz[k][t][tok][c] = sum_p mask[2k + p][t] dcomp[2k + p][t][tok][c];  d fill[c] = sum over everything of (1 - mask) dcomp
namespace {

// per = hw * ld elements of a frame (a multiple of 8); grid-stride over the 8-element pieces of comp
__global__ __launch_bounds__(256) void rl_gate_fwd_kernel(const float* __restrict__ z, const float* __restrict__ prob, const float* __restrict__ u,
                                                          const float* __restrict__ fill, bf16_t* __restrict__ comp, float* __restrict__ mask_out,
                                                          int B2, int T, long per, int LD)
{
    const long pieces = (long)B2 * T * (per / 8);
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < pieces; it += (long)gridDim.x * 256) {
        const long e = it * 8, f = e / per, r = e - f * per;                  // frame f = i * T + t of the doubled batch
        const int i = (int)(f / T), t = (int)(f - (long)i * T);
        const bool keep = u[f] < prob[(long)(i >> 1) * T + t];
        if (r == 0) mask_out[f] = keep ? 1.f : 0.f;
        float o[8];
        if (keep) {
            const float* zp = z + ((long)(i >> 1) * T + t) * per + r;
            const float4 a = *reinterpret_cast<const float4*>(zp), b = *reinterpret_cast<const float4*>(zp + 4);
            o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
        } else {
            const int c = (int)(r % LD);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = fill[c + k];
        }
        VecIO<bf16_t, 8>::store(comp + e, o);
    }
}

// grid (blocks, B = B2 / 2): a block walks 8-element pieces of one clip's (t, tok, c) volume; part (gridDim.y * gridDim.x, LD) = its d fill sums.
// blockDim = the largest multiple of LD / 8 that is <= 256 (252 threads for LD = 96): a thread keeps its channel group.
__global__ __launch_bounds__(256) void rl_gate_bwd_kernel(const bf16_t* __restrict__ dcomp, const float* __restrict__ mask, float* __restrict__ dz,
                                                          float* __restrict__ part, int T, long per, int LD, long pieces_per_block)
{
    __shared__ float red[256][8];
    const int k = blockIdx.y;
    const long pieces = (long)T * (per / 8);
    const long pbeg = (long)blockIdx.x * pieces_per_block;
    long pend = pbeg + pieces_per_block;
    if (pend > pieces) pend = pieces;
    const int cg = LD / 8;
    float df[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) df[j] = 0.f;
    for (long it = pbeg + threadIdx.x; it < pend; it += blockDim.x) {
        const long e = it * 8;
        const int t = (int)(e / per);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const long i = 2L * k + p;
            float g[8];
            VecIO<bf16_t, 8>::load(dcomp + (i * T) * per + e, g);
            const bool keep = mask[i * T + t] != 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { if (keep) acc[j] += g[j]; else df[j] += g[j]; }
        }
        float* zp = dz + ((long)k * T) * per + e;
        *reinterpret_cast<float4*>(zp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(zp + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    // per-channel sums over the threads that share a channel group (threadIdx.x % cg), fixed order
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = df[j];
    __syncthreads();
    if ((int)threadIdx.x < cg) {
        float* pr = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * LD + threadIdx.x * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = 0.f;
            for (int r = threadIdx.x; r < (int)blockDim.x; r += cg) a += red[r][j];
            pr[j] = a;
        }
    }
}

inline bool rl_gate_shape(int B2, int T, long per, int LD) { return B2 > 0 && (B2 & 1) == 0 && T > 0 && LD > 0 && LD % 8 == 0 && LD / 8 <= 256 && per > 0 && per % LD == 0; }

}  // namespace

// 1 if the rl gate kernels take the shape (LD a multiple of 8, at most 2048).
extern "C" int vvae_rl_gate_ok(int B2, int T, long per, int LD) { return rl_gate_shape(B2, T, per, LD) ? 1 : 0; }

// Rows of the d fill partial buffer vvae_rl_gate_bwd writes (each LD floats).
extern "C" int vvae_rl_gate_blocks(int B2, int T, long per)
{
    if (B2 <= 0 || T <= 0 || per <= 0) return 0;
    long pieces = (long)T * (per / 8), nb = (pieces + 2047) / 2048;
    const long cap = 2048 / (B2 / 2 > 0 ? B2 / 2 : 1);
    if (nb > cap) nb = cap > 0 ? cap : 1;
    return (int)(nb * (B2 / 2));
}

// z fp32 (B2/2, T, per); prob fp32 (B2/2, T); u fp32 (B2, T) uniform; fill fp32 (LD) -> comp bf16 (B2, T, per), mask fp32 (B2, T).
extern "C" int vvae_rl_gate_fwd(const float* z, const float* prob, const float* u, const float* fill, void* comp, float* mask, int B2, int T, long per,
                                int LD, void* stream)
{
    if (!z || !prob || !u || !fill || !comp || !mask || !rl_gate_shape(B2, T, per, LD) || ((uintptr_t)z % 16) || ((uintptr_t)comp % 16)) return VVAE_ERR_BAD_ARG;
    const long pieces = (long)B2 * T * (per / 8);
    long blocks = (pieces + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rl_gate_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, prob, u, fill, (bf16_t*)comp, mask, B2, T, per, LD);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// dcomp bf16 (B2, T, per); mask fp32 (B2, T) from the forward -> dz fp32 (B2/2, T, per), part fp32 (vvae_rl_gate_blocks(...), LD): d fill partial rows.
extern "C" int vvae_rl_gate_bwd(const void* dcomp, const float* mask, float* dz, float* part, int B2, int T, long per, int LD, void* stream)
{
    if (!dcomp || !mask || !dz || !part || !rl_gate_shape(B2, T, per, LD) || ((uintptr_t)dcomp % 16) || ((uintptr_t)dz % 16)) return VVAE_ERR_BAD_ARG;
    const int B = B2 / 2;
    const int nbx = vvae_rl_gate_blocks(B2, T, per) / B;
    const long pieces = (long)T * (per / 8);
    const int cg = LD / 8, nth = 256 / cg * cg;                     // whole channel-group runs per pass: a thread's channel group stays put
    long ppb = (pieces + nbx - 1) / nbx;
    ppb = (ppb + nth - 1) / nth * nth;
    hipLaunchKernelGGL(rl_gate_bwd_kernel, dim3((unsigned)nbx, (unsigned)B), dim3(nth), 0, (hipStream_t)stream, (const bf16_t*)dcomp, mask, dz, part, T, per,
                       LD, ppb);
    VVAE_LAUNCH_CHECK();
    return 0;
}
