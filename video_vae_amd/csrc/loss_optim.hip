// Reparameterise + KL, masked MSE/MAE, and the clip-by-global-norm + Adam update: fused HBM-streaming kernels.
//
// reparam : z = mean + eps * exp(log_var / 2)                    /root/reference/train/model.py:124-128
// KL      : mean_{t,hw,c}[ 0.5 (e^lv - 1 - lv + mu^2) m_t / len ]  /root/reference/train/rl_nonadversarial.py:146-147
// MSE/MAE : mean_{h,w,c}[ sum_t ((v - r) m_t)^2 / len ], |.| too   rl_nonadversarial.py:114-121
// optimiser: optax.chain(clip_by_global_norm(1.0), adam(...))      rl_nonadversarial.py:248-251
//
// One read of each operand; per-sample sums go wave-shuffle -> LDS -> one partial per workgroup, folded in fixed order by a
// second tiny kernel (no float atomics: every loss term and the gradient norm are bitwise reproducible run to run and rank to rank).
#include "common.hpp"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    __syncthreads();
    return t;   // valid on thread 0
}

__device__ __forceinline__ float seq_len(const float* __restrict__ mrow, int T) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += mrow[t];
    return fmaxf(s, 1.0f);
}

// grid (chunks, B).  per = hw*c elements per frame; M = T*per elements per sample.
template <typename T_>
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(const T_* __restrict__ mean, const T_* __restrict__ logvar,
                                                             const float* __restrict__ eps, const float* __restrict__ mask,
                                                             float* __restrict__ z, float* __restrict__ kl_part, int T, long per,
                                                             int elems_per_block)
{
    float* kl = kl_part;
    __shared__ float red[4];
    const int b = blockIdx.y;
    const long M = (long)T * per;
    const long beg = (long)blockIdx.x * elems_per_block;
    long end = beg + elems_per_block; if (end > M) end = M;
    const float* mrow = mask ? mask + (long)b * T : nullptr;
    float acc = 0.f;
    for (long i = beg + threadIdx.x; i < end; i += 256) {
        const long g = (long)b * M + i;
        const float mu = ldf(mean + g), lv = ldf(logvar + g);
        if (z) z[g] = mu + eps[g] * __expf(0.5f * lv);
        if (kl) {
            const float m = mrow ? mrow[i / per] : 1.f;
            acc += 0.5f * (__expf(lv) - 1.f - lv + mu * mu) * m;
        }
    }
    if (kl) {
        const float t = block_sum(acc, red);
        if (threadIdx.x == 0) {
            const float len = mrow ? seq_len(mrow, T) : (float)T;
            kl_part[(long)b * gridDim.x + blockIdx.x] = t / (len * (float)M);
        }
    }
}

template <typename T_>
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(const T_* __restrict__ mean, const T_* __restrict__ logvar,
                                                             const float* __restrict__ eps, const float* __restrict__ mask,
                                                             const float* __restrict__ dz, const float* __restrict__ gkl,
                                                             T_* __restrict__ dmean, T_* __restrict__ dlogvar, int T, long per,
                                                             int elems_per_block)
{
    const int b = blockIdx.y;
    const long M = (long)T * per;
    const long beg = (long)blockIdx.x * elems_per_block;
    long end = beg + elems_per_block; if (end > M) end = M;
    const float* mrow = mask ? mask + (long)b * T : nullptr;
    float kscale = 0.f;
    if (gkl) {
        const float len = mrow ? seq_len(mrow, T) : (float)T;
        kscale = gkl[b] / (len * (float)M);
    }
    for (long i = beg + threadIdx.x; i < end; i += 256) {
        const long g = (long)b * M + i;
        const float mu = ldf(mean + g), lv = ldf(logvar + g);
        float dm = 0.f, dl = 0.f;
        if (dz) { const float d = dz[g]; dm = d; dl = d * eps[g] * 0.5f * __expf(0.5f * lv); }
        if (gkl) {
            const float m = mrow ? mrow[i / per] : 1.f;
            dm += kscale * m * mu;
            dl += kscale * m * 0.5f * (__expf(lv) - 1.f);
        }
        stf(dmean + g, dm);
        stf(dlogvar + g, dl);
    }
}

// grid (chunks, B).  P = h*w*c elements per frame.  video sample index = b / video_div (pair doubling).
template <typename T_, int VEC>
__global__ __launch_bounds__(256) void masked_mse_mae_fwd_kernel(const T_* __restrict__ video, const T_* __restrict__ recon,
                                                                 const float* __restrict__ mask, float* __restrict__ mse_part,
                                                                 float* __restrict__ mae_part, int T, long P, int video_div,
                                                                 int elems_per_block)
{
    __shared__ float red[4];
    const int b = blockIdx.y;
    const long M = (long)T * P;
    const long beg = (long)blockIdx.x * elems_per_block;
    long end = beg + elems_per_block; if (end > M) end = M;
    const float* mrow = mask + (long)b * T;
    const T_* vs = video + (long)(b / video_div) * M;
    const T_* rs = recon + (long)b * M;
    float a2 = 0.f, a1 = 0.f;
    for (long i = beg + (long)threadIdx.x * VEC; i < end; i += 256 * VEC) {
        float v[VEC], r[VEC];
        VecIO<T_, VEC>::load(vs + i, v);
        VecIO<T_, VEC>::load(rs + i, r);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float m = mrow[(i + k) / P];
            const float e = (v[k] - r[k]) * m;
            a2 += e * e; a1 += fabsf(e);
        }
    }
    const float t2 = block_sum(a2, red);
    const float t1 = block_sum(a1, red);
    if (threadIdx.x == 0) {
        const float sc = 1.f / (seq_len(mrow, T) * (float)P);
        mse_part[(long)b * gridDim.x + blockIdx.x] = t2 * sc;
        mae_part[(long)b * gridDim.x + blockIdx.x] = t1 * sc;
    }
}

template <typename T_, int VEC>
__global__ __launch_bounds__(256) void masked_mse_mae_bwd_kernel(const T_* __restrict__ video, const T_* __restrict__ recon,
                                                                 const float* __restrict__ mask, const float* __restrict__ gmse,
                                                                 const float* __restrict__ gmae, T_* __restrict__ drecon, int T, long P,
                                                                 int video_div, int elems_per_block)
{
    const int b = blockIdx.y;
    const long M = (long)T * P;
    const long beg = (long)blockIdx.x * elems_per_block;
    long end = beg + elems_per_block; if (end > M) end = M;
    const float* mrow = mask + (long)b * T;
    const T_* vs = video + (long)(b / video_div) * M;
    const T_* rs = recon + (long)b * M;
    T_* ds = drecon + (long)b * M;
    const float sc = 1.f / (seq_len(mrow, T) * (float)P);
    const float g2 = (gmse ? gmse[b] : 0.f) * sc, g1 = (gmae ? gmae[b] : 0.f) * sc;
    for (long i = beg + (long)threadIdx.x * VEC; i < end; i += 256 * VEC) {
        float v[VEC], r[VEC];
        VecIO<T_, VEC>::load(vs + i, v);
        VecIO<T_, VEC>::load(rs + i, r);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float m = mrow[(i + k) / P];
            const float e = (v[k] - r[k]) * m;                       // d/dr of e = -m
            const float sg = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
            r[k] = -m * (2.f * e * g2 + sg * g1);
        }
        VecIO<T_, VEC>::store(ds + i, r);
    }
}

// out[b] = sum_c part[b][c], c in fixed order (lane-strided, then the wave's shuffle tree): the second half of the per-sample
// loss reductions above.  grid B, one wave.
__global__ __launch_bounds__(64) void sum_chunks_kernel(const float* __restrict__ part, int nchunks, float* __restrict__ out)
{
    const float* row = part + (long)blockIdx.x * nchunks;
    float acc = 0.f;
    for (int c = threadIdx.x; c < nchunks; c += 64) acc += row[c];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// two partial arrays of B rows each, laid one behind the other, folded by one launch (grid 2B): rows [0, B) -> out0, [B, 2B) -> out1
__global__ __launch_bounds__(64) void sum_chunks2_kernel(const float* __restrict__ part, int nchunks, float* __restrict__ out0,
                                                        float* __restrict__ out1, int B)
{
    const float* row = part + (long)blockIdx.x * nchunks;
    float acc = 0.f;
    for (int c = threadIdx.x; c < nchunks; c += 64) acc += row[c];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) { if ((int)blockIdx.x < B) out0[blockIdx.x] = acc; else out1[blockIdx.x - B] = acc; }
}

// The scalar end of the recon + KL loss (reference train/legacy/training_loop_adversarial.py:100-124), value AND gradients in one
// launch: as framework ops this tail is ~45 kernels of a few hundred bytes each per step, every one a 5 us dispatch.
//   len_b = max(sum_t mask, 1); density_b = sum_t sel * mask / len_b; d_b = density_b - 1 / max_rate; m_b = d_b < 0 ? R d_b : d_b
//   selection_loss = mean_b m_b^2;  loss = mean_b mse + gamma1 selection_loss + gamma2 mean_b kl
// out[5] = loss, MSE, selection_loss, kl_loss, mean density.   grads = [d loss / d mse_b (B) | d / d kl_b (B) | d / d sel_bt (B T)].
// One workgroup; sample b is thread b's (strided), the means are summed by thread 0 in index order: deterministic.
constexpr int TAIL_MAX_B = 1024;
__global__ __launch_bounds__(256) void loss_tail_plain_kernel(const float* __restrict__ mse, int mse_cols, const float* __restrict__ kl, int kl_cols,
                                                             const float* __restrict__ sel, const float* __restrict__ mask, int B, int T,
                                                             float inv_max_rate, float magnify, float gamma1, float gamma2,
                                                             float* __restrict__ out, float* __restrict__ grads)
{
    __shared__ float sq[TAIL_MAX_B], dens[TAIL_MAX_B], msum[TAIL_MAX_B], ksum[TAIL_MAX_B];
    __shared__ float red[256];
    const float invB = 1.f / (float)B;
    // per-sample MSE / KL terms from their partial sums.  Few samples with many partials each (B = 4: 512 + 16): the whole workgroup
    // sums one sample at a time -- thread i takes columns i, i + 256, ..., then a fixed tree -- instead of one thread walking 2 048 dependent
    // loads (93 us); many samples with few partials: one thread per sample.  Either way a fixed order.
    if (mse_cols + kl_cols > 16 && B <= 64) {
        for (int b = 0; b < B; ++b)
            for (int which = 0; which < 2; ++which) {
                const float* p = which ? kl + (long)b * kl_cols : mse + (long)b * mse_cols;
                const int n = which ? kl_cols : mse_cols;
                float a = 0.f;
                for (int c = threadIdx.x; c < n; c += 256) a += p[c];
                red[threadIdx.x] = a;
                __syncthreads();
                for (int st = 128; st > 0; st >>= 1) {
                    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
                    __syncthreads();
                }
                if (threadIdx.x == 0) (which ? ksum : msum)[b] = red[0];
                __syncthreads();
            }
    } else {
        for (int b = threadIdx.x; b < B; b += 256) {
            float kb = 0.f, mb = 0.f;
            for (int c = 0; c < kl_cols; ++c) kb += kl[(long)b * kl_cols + c];
            for (int c = 0; c < mse_cols; ++c) mb += mse[(long)b * mse_cols + c];
            msum[b] = mb; ksum[b] = kb;
        }
    }
    for (int b = threadIdx.x; b < B; b += 256) {
        float len = 0.f, ssum = 0.f;
        for (int t = 0; t < T; ++t) { const float m = mask[b * T + t]; len += m; ssum += sel[b * T + t] * m; }
        len = fmaxf(len, 1.f);
        const float density = ssum / len;
        const float d = density - inv_max_rate;
        const float slope = d < 0.f ? magnify : 1.f;
        const float m = d * slope;
        sq[b] = m * m;
        dens[b] = density;
        const float gd = gamma1 * invB * 2.f * m * slope / len;     // d loss / d ssum
        for (int t = 0; t < T; ++t) grads[2 * B + b * T + t] = gd * mask[b * T + t];
        grads[b] = invB;
        grads[B + b] = gamma2 * invB;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, k = 0.f, q = 0.f, dn = 0.f;
        for (int b = 0; b < B; ++b) { a += msum[b]; k += ksum[b]; q += sq[b]; dn += dens[b]; }
        a *= invB; k *= invB; q *= invB; dn *= invB;
        out[0] = a + gamma1 * q + gamma2 * k;
        out[1] = a; out[2] = q; out[3] = k; out[4] = dn;
    }
}

// The scalar end of the pair / REINFORCE loss of the rl flavour (reference train/rl_nonadversarial.py:130-186), value and gradients in ONE launch
// (as framework ops: ~100 kernels on (2b,), (b, 2, t) and scalar tensors in every step).  Samples 2k, 2k + 1 are a pair.
//   len_i = max(sum_t mask, 1); density_i = sum_t action * mask / len_i; d = density_i - 1 / max_rate; m = d < 0 ? R d : d; sel_loss_i = m^2
//   psl_i = mse_i + g3 perc_i + g1 sel_loss_i + g2 kl_i + g4 mae_i
//   pair: mean, population std + 1e-6 -> disadvantage_i = (psl_i - mean) / std                      (no gradient: stop_gradient)
//   raw_it = clip(|sel_it + action_it - 1|, 1e-6, 1 - 1e-6); probs_i = prod_t raw_it / stop_gradient(raw_it) = 1 (its gradient is what counts)
//   loss = mean_i psl_i + w mean_i (probs_i disadvantage_i)
// out[9] = loss, MSE, perceptual, selection_loss, kl_loss, mean density, mean trajectory probability (prod_t of the masked raw_it), rl_loss, MAE.
// grads = [d / d mse_i | d / d mae_i | d / d perc_i | d / d kl_i | d / d sel_it]  (B2 each, then B2 * T).
// mse / mae arrive as (B2, cols) partial sums (cols >= 1); perc may be NULL (zeros).  One workgroup, fixed summation order.
__global__ __launch_bounds__(256) void loss_tail_rl_kernel(const float* __restrict__ mse, const float* __restrict__ mae, int cols,
                                                          const float* __restrict__ perc, const float* __restrict__ kl, int kl_cols,
                                                          const float* __restrict__ sel, const float* __restrict__ act, const float* __restrict__ mask,
                                                          int B2, int T, float inv_max_rate, float magnify, float g1, float g2, float g3, float g4,
                                                          float w, float* __restrict__ out, float* __restrict__ grads)
{
    __shared__ float psl[TAIL_MAX_B], ms[TAIL_MAX_B], ma[TAIL_MAX_B], sq[TAIL_MAX_B], dens[TAIL_MAX_B], dis[TAIL_MAX_B], traj[TAIL_MAX_B], klr[TAIL_MAX_B];
    const float invB = 1.f / (float)B2;
    for (int i = threadIdx.x; i < B2; i += 256) {
        float a = 0.f, c = 0.f;
        for (int k = 0; k < cols; ++k) { a += mse[(long)i * cols + k]; c += mae[(long)i * cols + k]; }
        ms[i] = a; ma[i] = c;
        float kb = 0.f;                                        // kl arrives as (B2, kl_cols) partial sums of the per-sample term (one per frame from ops.encoder_head_rl)
        for (int k = 0; k < kl_cols; ++k) kb += kl[(long)i * kl_cols + k];
        klr[i] = kb;
        float len = 0.f, ssum = 0.f, tp = 1.f;
        for (int t = 0; t < T; ++t) {
            const float m = mask[i * T + t];
            len += m; ssum += act[i * T + t] * m;
            const float raw = fminf(fmaxf(fabsf(sel[i * T + t] + act[i * T + t] - 1.f), 1e-6f), 1.f - 1e-6f);
            tp *= m != 0.f ? raw : 1.f;
        }
        len = fmaxf(len, 1.f);
        const float density = ssum / len;
        const float d = density - inv_max_rate;
        const float mm = d < 0.f ? d * magnify : d;
        sq[i] = mm * mm; dens[i] = density; traj[i] = tp;
        psl[i] = a + g3 * (perc ? perc[i] : 0.f) + g1 * sq[i] + g2 * kb + g4 * c;
        grads[i] = invB; grads[B2 + i] = g4 * invB; grads[2 * B2 + i] = g3 * invB; grads[3 * B2 + i] = g2 * invB;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < B2 / 2; p += 256) {
        const float x0 = psl[2 * p], x1 = psl[2 * p + 1];
        const float mean = 0.5f * (x0 + x1);
        const float sd = sqrtf(0.5f * ((x0 - mean) * (x0 - mean) + (x1 - mean) * (x1 - mean))) + 1e-6f;
        dis[2 * p] = (x0 - mean) / sd; dis[2 * p + 1] = (x1 - mean) / sd;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B2; i += 256) {
        const float gi = w * invB * dis[i];
        for (int t = 0; t < T; ++t) {
            const float x = sel[i * T + t] + act[i * T + t] - 1.f;
            const float ax = fabsf(x);
            const bool inside = ax >= 1e-6f && ax <= 1.f - 1e-6f;           // clamp passes the gradient on [min, max]
            const float raw = fminf(fmaxf(ax, 1e-6f), 1.f - 1e-6f);
            const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
            grads[4 * B2 + i * T + t] = (mask[i * T + t] != 0.f && inside) ? gi * sg / raw : 0.f;
        }
    }
    if (threadIdx.x == 0) {
        float a = 0.f, c = 0.f, pc = 0.f, q = 0.f, k = 0.f, dn = 0.f, tp = 0.f, rl = 0.f, ps = 0.f;
        for (int i = 0; i < B2; ++i) {
            a += ms[i]; c += ma[i]; pc += perc ? perc[i] : 0.f; q += sq[i]; k += klr[i]; dn += dens[i]; tp += traj[i]; rl += dis[i]; ps += psl[i];
        }
        out[0] = ps * invB + w * rl * invB;
        out[1] = a * invB; out[2] = pc * invB; out[3] = q * invB; out[4] = k * invB; out[5] = dn * invB; out[6] = tp * invB; out[7] = rl * invB;
        out[8] = c * invB;
    }
}

// ---------------------------------------------------------------------------------------------- optimiser
constexpr int SQN_MAX_BLOCKS = 1024;

// part[blockIdx] = sum of squares of this workgroup's grid-stride share, fixed order (no atomics).
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n, double* __restrict__ part)
{
    __shared__ float red[4];
    float acc = 0.f;
    const long n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 t = g4[i];
        acc += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) { const float t = g[n4 * 4 + threadIdx.x]; acc += t * t; }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = (double)t;
}

// p, m, v updated in place; optional bf16 shadow copy of p.  gscale: grads are multiplied by gscale first (1/world).
// clip: g *= max_norm/||g|| only if ||g|| >= max_norm, with ||g|| = gscale*sqrt(sum of gnorm_part)  (optax semantics).
// Every workgroup folds the <= 1024 partial sums of squares itself, in one fixed order (8 KB out of L2), so all of them -- and all
// ranks of a data-parallel job, which hold the same all-reduced gradient -- apply the bitwise-same clip factor.
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ p_bf16, long n,
                                                        const double* __restrict__ gnorm_part, int nparts,
                                                        double* __restrict__ gnorm_sq_out, float gscale, float max_norm, float lr,
                                                        float b1, float b2, float eps, float c1, float c2)
{
    float clip = gscale;
    if (gnorm_part) {
        __shared__ double dred[256];
        double a = 0.0;
        for (int i = threadIdx.x; i < nparts; i += 256) a += gnorm_part[i];
        dred[threadIdx.x] = a;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) dred[threadIdx.x] += dred[threadIdx.x + st];
            __syncthreads();
        }
        const double tot = dred[0];
        if (gnorm_sq_out && blockIdx.x == 0 && threadIdx.x == 0) *gnorm_sq_out = tot;
        const float gn = gscale * (float)sqrt(tot);
        if (gn >= max_norm) clip = gscale * max_norm / gn;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * clip;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float pi = p[i] - lr * (mi / c1) / (sqrtf(vi / c2) + eps);
        p[i] = pi;
        if (p_bf16) p_bf16[i] = f2bf(pi);
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = f2bf(x[i]);
}

inline int pick_epb(long M, int B) {
    long want = 2048 / (B > 0 ? B : 1); if (want < 1) want = 1;
    long epb = (M + want - 1) / want;
    epb = ((epb + 2047) / 2048) * 2048;
    return (int)epb;
}

}  // namespace

// fp32 scratch floats the per-sample loss reductions below need for their per-workgroup partials (M = elements per sample).
extern "C" size_t vvae_loss_part_floats(int B, long M)
{
    if (B <= 0 || M <= 0) return 0;
    return (size_t)2 * B * ceil_div(M, (long)pick_epb(M, B));
}

// mean/logvar (B, T, per) in `dtype`; eps, z fp32; mask fp32 (B,T) or NULL; kl fp32 [B] overwritten.  z or kl may be NULL.
// part: vvae_loss_part_floats(B, T * per) floats of scratch (needed when kl != NULL).
extern "C" int vvae_reparam_kl_fwd(const void* mean, const void* logvar, const float* eps, const float* mask, float* z, float* kl,
                                   float* part, int B, int T, long per, int dtype, void* stream)
{
    if (!mean || !logvar || (!z && !kl) || (z && !eps) || (kl && !part) || B <= 0 || T <= 0 || per <= 0) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)T * per;
    const int epb = pick_epb(M, B);
    dim3 grid(ceil_div(M, epb), B);
    float* klp = kl ? part : nullptr;
    if (dtype == VVAE_DT_F32)
        hipLaunchKernelGGL((reparam_kl_fwd_kernel<float>), grid, dim3(256), 0, s, (const float*)mean, (const float*)logvar, eps, mask, z, klp, T, per, epb);
    else if (dtype == VVAE_DT_BF16)
        hipLaunchKernelGGL((reparam_kl_fwd_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)mean, (const bf16_t*)logvar, eps, mask, z, klp, T, per, epb);
    else return VVAE_ERR_BAD_ARG;
    if (kl) hipLaunchKernelGGL(sum_chunks_kernel, dim3(B), dim3(64), 0, s, part, (int)grid.x, kl);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// dz fp32 (or NULL), gkl fp32 [B] (or NULL) -> dmean, dlogvar in `dtype` (overwritten).
extern "C" int vvae_reparam_kl_bwd(const void* mean, const void* logvar, const float* eps, const float* mask, const float* dz,
                                   const float* gkl, void* dmean, void* dlogvar, int B, int T, long per, int dtype, void* stream)
{
    if (!mean || !logvar || !dmean || !dlogvar || (dz && !eps) || B <= 0 || T <= 0 || per <= 0) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)T * per;
    const int epb = pick_epb(M, B);
    dim3 grid(ceil_div(M, epb), B);
    if (dtype == VVAE_DT_F32)
        hipLaunchKernelGGL((reparam_kl_bwd_kernel<float>), grid, dim3(256), 0, s, (const float*)mean, (const float*)logvar, eps, mask, dz, gkl, (float*)dmean, (float*)dlogvar, T, per, epb);
    else if (dtype == VVAE_DT_BF16)
        hipLaunchKernelGGL((reparam_kl_bwd_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)mean, (const bf16_t*)logvar, eps, mask, dz, gkl, (bf16_t*)dmean, (bf16_t*)dlogvar, T, per, epb);
    else return VVAE_ERR_BAD_ARG;
    VVAE_LAUNCH_CHECK();
    return 0;
}

// video (B/video_div, T, P), recon (B, T, P) in `dtype`; mask fp32 (B,T); mse, mae fp32 [B] overwritten -- or both NULL: the caller takes
// the partial sums themselves, part = [mse (B, chunks) | mae (B, chunks)] with chunks = vvae_loss_part_floats / (2 B) (vvae_loss_tail_plain
// adds a sample's partials up itself).  part: vvae_loss_part_floats(B, T * P) floats of scratch.
extern "C" int vvae_masked_mse_mae_fwd(const void* video, const void* recon, const float* mask, float* mse_out, float* mae_out,
                                       float* part, int B, int T, long P, int video_div, int dtype, void* stream)
{
    if (!video || !recon || !mask || (!mse_out) != (!mae_out) || !part || B <= 0 || T <= 0 || P <= 0 || video_div <= 0) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)T * P;
    const int epb = pick_epb(M, B);
    dim3 grid(ceil_div(M, epb), B);
    float* mse = part;
    float* mae = part + (size_t)B * grid.x;
    const bool al = ((uintptr_t)video % 16) == 0 && ((uintptr_t)recon % 16) == 0;
    if (dtype == VVAE_DT_F32) {
        if (al && M % 4 == 0) hipLaunchKernelGGL((masked_mse_mae_fwd_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)video, (const float*)recon, mask, mse, mae, T, P, video_div, epb);
        else hipLaunchKernelGGL((masked_mse_mae_fwd_kernel<float, 1>), grid, dim3(256), 0, s, (const float*)video, (const float*)recon, mask, mse, mae, T, P, video_div, epb);
    } else if (dtype == VVAE_DT_BF16) {
        if (al && M % 8 == 0) hipLaunchKernelGGL((masked_mse_mae_fwd_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t*)video, (const bf16_t*)recon, mask, mse, mae, T, P, video_div, epb);
        else hipLaunchKernelGGL((masked_mse_mae_fwd_kernel<bf16_t, 1>), grid, dim3(256), 0, s, (const bf16_t*)video, (const bf16_t*)recon, mask, mse, mae, T, P, video_div, epb);
    } else return VVAE_ERR_BAD_ARG;
    if (mse_out) hipLaunchKernelGGL(sum_chunks2_kernel, dim3(2 * B), dim3(64), 0, s, part, (int)grid.x, mse_out, mae_out, B);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// mse_ps fp32 (B, mse_cols), kl_ps fp32 (B, kl_cols): partial sums of the per-sample MSE / KL terms; selection, mask fp32 (B, T) contiguous.
// out fp32 [5], grads fp32 [2 B + B T] (see loss_tail_plain_kernel).
extern "C" int vvae_loss_tail_plain(const float* mse_ps, int mse_cols, const float* kl_ps, int kl_cols, const float* selection, const float* mask, int B, int T,
                                    float max_compression_rate, float magnify_negatives_rate, float gamma1, float gamma2, float* out,
                                    float* grads, void* stream)
{
    if (!mse_ps || mse_cols <= 0 || !kl_ps || kl_cols <= 0 || !selection || !mask || !out || !grads || B <= 0 || B > TAIL_MAX_B || T <= 0 || !(max_compression_rate > 0.f))
        return VVAE_ERR_BAD_ARG;
    hipLaunchKernelGGL(loss_tail_plain_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mse_ps, mse_cols, kl_ps, kl_cols, selection, mask, B, T,
                       1.f / max_compression_rate, magnify_negatives_rate, gamma1, gamma2, out, grads);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// The rl flavour's loss tail (see loss_tail_rl_kernel).  mse, mae fp32 (B2, cols); perc fp32 [B2] or NULL; kl fp32 (B2, kl_cols) partial sums; sel (probabilities), act
// (sampled actions), mask fp32 (B2, T) contiguous; B2 even, <= 1024.  out fp32 [9]; grads fp32 [4 B2 + B2 T].
extern "C" int vvae_loss_tail_rl(const float* mse, const float* mae, int cols, const float* perc, const float* kl, int kl_cols, const float* sel, const float* act,
                                 const float* mask, int B2, int T, float max_compression_rate, float magnify_negatives_rate, float gamma1,
                                 float gamma2, float gamma3, float gamma4, float rl_loss_weight, float* out, float* grads, void* stream)
{
    if (!mse || !mae || cols <= 0 || !kl || kl_cols <= 0 || !sel || !act || !mask || !out || !grads || B2 <= 0 || (B2 & 1) || B2 > TAIL_MAX_B || T <= 0 ||
        !(max_compression_rate > 0.f)) return VVAE_ERR_BAD_ARG;
    hipLaunchKernelGGL(loss_tail_rl_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mse, mae, cols, perc, kl, kl_cols, sel, act, mask, B2, T,
                       1.f / max_compression_rate, magnify_negatives_rate, gamma1, gamma2, gamma3, gamma4, rl_loss_weight, out, grads);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// gmse / gmae fp32 [B] (either may be NULL) -> drecon (B, T, P) in `dtype`.
extern "C" int vvae_masked_mse_mae_bwd(const void* video, const void* recon, const float* mask, const float* gmse, const float* gmae,
                                       void* drecon, int B, int T, long P, int video_div, int dtype, void* stream)
{
    if (!video || !recon || !mask || !drecon || B <= 0 || T <= 0 || P <= 0 || video_div <= 0) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)T * P;
    const int epb = pick_epb(M, B);
    dim3 grid(ceil_div(M, epb), B);
    const bool al = ((uintptr_t)video % 16) == 0 && ((uintptr_t)recon % 16) == 0 && ((uintptr_t)drecon % 16) == 0;
    if (dtype == VVAE_DT_F32) {
        if (al && M % 4 == 0) hipLaunchKernelGGL((masked_mse_mae_bwd_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)video, (const float*)recon, mask, gmse, gmae, (float*)drecon, T, P, video_div, epb);
        else hipLaunchKernelGGL((masked_mse_mae_bwd_kernel<float, 1>), grid, dim3(256), 0, s, (const float*)video, (const float*)recon, mask, gmse, gmae, (float*)drecon, T, P, video_div, epb);
    } else if (dtype == VVAE_DT_BF16) {
        if (al && M % 8 == 0) hipLaunchKernelGGL((masked_mse_mae_bwd_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t*)video, (const bf16_t*)recon, mask, gmse, gmae, (bf16_t*)drecon, T, P, video_div, epb);
        else hipLaunchKernelGGL((masked_mse_mae_bwd_kernel<bf16_t, 1>), grid, dim3(256), 0, s, (const bf16_t*)video, (const bf16_t*)recon, mask, gmse, gmae, (bf16_t*)drecon, T, P, video_div, epb);
    } else return VVAE_ERR_BAD_ARG;
    VVAE_LAUNCH_CHECK();
    return 0;
}

// Number of fp64 partials vvae_sqnorm_partials writes for a buffer of n floats (<= 1024).
extern "C" int vvae_sqnorm_blocks(long n)
{
    if (n <= 0) return 0;
    long blocks = n / 1024 + 1; if (blocks > SQN_MAX_BLOCKS) blocks = SQN_MAX_BLOCKS;
    return (int)blocks;
}

// part[0 .. vvae_sqnorm_blocks(n)) (fp64, device) = per-workgroup sums of g[i]^2, each in a fixed order; vvae_adam_clip_step folds them.
extern "C" int vvae_sqnorm_partials(const float* g, long n, double* part, void* stream)
{
    if (!g || !part || n <= 0 || ((uintptr_t)g % 16) != 0) return VVAE_ERR_BAD_ARG;
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)vvae_sqnorm_blocks(n)), dim3(256), 0, (hipStream_t)stream, g, n, part);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// One optimizer.update over a flat fp32 buffer.  gnorm_part: nparts device fp64 partial sums of squares of the *unscaled* grads
// (vvae_sqnorm_partials) or NULL (no clip); gnorm_sq_out: device fp64 that receives their sum (for logging), or NULL.
extern "C" int vvae_adam_clip_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, const double* gnorm_part,
                                   int nparts, double* gnorm_sq_out, float gscale, float max_norm, float lr, float b1, float b2,
                                   float eps, long count, void* stream)
{
    if (!p || !g || !m || !v || n <= 0 || count < 1 || (gnorm_part && (nparts <= 0 || nparts > SQN_MAX_BLOCKS))) return VVAE_ERR_BAD_ARG;
    const float c1 = 1.f - powf(b1, (float)count), c2 = 1.f - powf(b2, (float)count);
    long blocks = n / 1024 + 1; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, n,
                       gnorm_part, nparts, gnorm_sq_out, gscale, max_norm, lr, b1, b2, eps, c1, c2);
    VVAE_LAUNCH_CHECK();
    return 0;
}

extern "C" int vvae_cast_f32_to_bf16(const float* x, void* y, long n, void* stream)
{
    if (!x || !y || n <= 0) return VVAE_ERR_BAD_ARG;
    long blocks = n / 1024 + 1; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
    VVAE_LAUNCH_CHECK();
    return 0;
}

namespace {
// out[c] = sum_r part[r][c], r in fixed order: the fold of the per-workgroup partial rows the LayerNorm / attention backward
// kernels emit (parameter gradients without float atomics).  Block = 8 column quads x 32 row lanes; a thread keeps 16 float4
// loads in flight (the partials are L2-resident: latency, not bandwidth, is what a 1.5-million-element fold pays for).
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ part, int rows, int cols, float* __restrict__ out)
{
    __shared__ float4 red[32][8];
    const int cq = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = (blockIdx.x * 8 + cq) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols) {
        for (int r0 = rl; r0 < rows; r0 += 32 * 16) {
            float4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = r0 + 32 * i;
                v[i] = r < rows ? *reinterpret_cast<const float4*>(part + (long)r * cols + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
        }
    }
    red[rl][cq] = s;
    __syncthreads();
    for (int st = 16; st > 0; st >>= 1) {
        if (rl < st) {
            const float4 a = red[rl][cq], b = red[rl + st][cq];
            red[rl][cq] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        __syncthreads();
    }
    if (rl == 0 && c < cols) *reinterpret_cast<float4*>(out + c) = red[0][cq];
}
}  // namespace

// part: fp32 (rows, cols) contiguous, cols % 4 == 0; out: fp32 (cols) = column sums, deterministic order.
extern "C" int vvae_sum_rows(const float* part, int rows, int cols, float* out, void* stream)
{
    if (!part || !out || rows <= 0 || cols <= 0 || cols % 4 || ((uintptr_t)part % 16) || ((uintptr_t)out % 16)) return VVAE_ERR_BAD_ARG;
    hipLaunchKernelGGL(sum_rows_kernel, dim3(ceil_div(cols, 32)), dim3(256), 0, (hipStream_t)stream, part, rows, cols, out);
    VVAE_LAUNCH_CHECK();
    return 0;
}

namespace {
// Grouped fold: up to 64 partial buffers in one launch (the dgamma / dbeta / q-k-scale partial rows of a whole backward pass).
// Entry e: out[c] = sum_r part_e[r * cols_e + c]; columns < n0_e go to d0_e[c], the rest to d1_e[c - n0_e] (or nowhere if d1_e is
// NULL).  Blocks [block_start_e, block_start_{e+1}) belong to entry e, 128 columns each (cols_e % 4 == 0); fixed row order.
constexpr int FOLD_MAX = 64;
struct FoldEntry { const float* part; float* d0; float* d1; int rows, cols, n0, block_start; };
struct FoldArgs { FoldEntry e[FOLD_MAX]; int n; };

__global__ __launch_bounds__(256) void fold_rows_grouped_kernel(FoldArgs g)
{
    // 256 threads = 32 float4 columns (128 columns of the entry) x 8 row lanes; 8 sixteen-byte loads in flight per thread
    __shared__ float4 red[8][32];
    int ei = 0;
    for (int i = 1; i < g.n; ++i) ei = (int)blockIdx.x >= g.e[i].block_start ? i : ei;
    const FoldEntry& E = g.e[ei];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = (((int)blockIdx.x - E.block_start) * 32 + cl) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < E.cols) {
        for (int r0 = rl; r0 < E.rows; r0 += 8 * 8) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + 8 * i;
                v[i] = r < E.rows ? *reinterpret_cast<const float4*>(E.part + (long)r * E.cols + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
        }
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < E.cols) {
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* q = reinterpret_cast<const float*>(&red[0][cl]) + e;          // element e of row lane j at q[j * 128]
            t[e] = ((q[0] + q[128]) + (q[256] + q[384])) + ((q[512] + q[640]) + (q[768] + q[896]));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ce = c + e;
            if (ce < E.n0) E.d0[ce] = t[e];
            else if (E.d1) E.d1[ce - E.n0] = t[e];
        }
    }
}
}  // namespace

// n <= 64 partial buffers folded in one launch.  part[i]: fp32 (rows[i], cols[i]) contiguous; d0[i] receives columns [0, n0[i]),
// d1[i] (or NULL) columns [n0[i], cols[i]).  Host arrays of device pointers / ints.  Fixed row order: deterministic.
extern "C" int vvae_fold_rows_grouped(const void* const* part, float* const* d0, float* const* d1, const int* rows, const int* cols,
                                      const int* n0, int n, void* stream)
{
    if (!part || !d0 || !d1 || !rows || !cols || !n0 || n <= 0 || n > FOLD_MAX) return VVAE_ERR_BAD_ARG;
    FoldArgs g;
    g.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!part[i] || !d0[i] || rows[i] <= 0 || cols[i] <= 0 || n0[i] <= 0 || n0[i] > cols[i]) return VVAE_ERR_BAD_ARG;
        g.e[i] = FoldEntry{(const float*)part[i], d0[i], d1[i], rows[i], cols[i], n0[i], blocks};
        if (cols[i] % 4 || ((uintptr_t)part[i] % 16)) return VVAE_ERR_BAD_ARG;
        blocks += ceil_div(cols[i], 128);
    }
    hipLaunchKernelGGL(fold_rows_grouped_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
    VVAE_LAUNCH_CHECK();
    return 0;
}

namespace {
// Grouped copy: up to 64 contiguous fp32 ranges in one launch (gradients landing in the optimizer's flat buffer).
// Blocks [block_start_e, block_start_{e+1}) belong to entry e, 1024 floats each.
struct CopyEntry { const float* src; float* dst; long n; int block_start; };
struct CopyArgs { CopyEntry e[FOLD_MAX]; int n; };

__global__ __launch_bounds__(256) void copy_grouped_kernel(CopyArgs g)
{
    int ei = 0;
    for (int i = 1; i < g.n; ++i) ei = (int)blockIdx.x >= g.e[i].block_start ? i : ei;
    const CopyEntry& E = g.e[ei];
    const long i0 = ((long)((int)blockIdx.x - E.block_start) * 256 + threadIdx.x) * 4;
    if (i0 + 4 <= E.n && (((uintptr_t)E.src | (uintptr_t)E.dst) & 15) == 0) {
        *reinterpret_cast<float4*>(E.dst + i0) = *reinterpret_cast<const float4*>(E.src + i0);
    } else {
        for (long i = i0; i < i0 + 4 && i < E.n; ++i) E.dst[i] = E.src[i];
    }
}
}  // namespace

// n <= 64 contiguous fp32 ranges copied in one launch: dst[i][0..count[i]) = src[i][0..count[i]).  Host arrays.
extern "C" int vvae_copy_grouped(const float* const* src, float* const* dst, const long* count, int n, void* stream)
{
    if (!src || !dst || !count || n <= 0 || n > FOLD_MAX) return VVAE_ERR_BAD_ARG;
    CopyArgs g;
    g.n = n;
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i] || count[i] <= 0) return VVAE_ERR_BAD_ARG;
        g.e[i] = CopyEntry{src[i], dst[i], count[i], (int)blocks};
        blocks += ceil_div(count[i], 1024L);
        if (blocks > 0x7fffffffL) return VVAE_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(copy_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    VVAE_LAUNCH_CHECK();
    return 0;
}

namespace {
template <int O> __global__ void xor_lane_selftest_kernel(const float* __restrict__ x, float* __restrict__ y, double* __restrict__ yd)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    y[i] = xor_lane<O>(x[i]);
    yd[i] = xor_lane<O>((double)x[i] * 1.000000001);
}
}  // namespace

// Test hook: y[i] = x[i ^ o] within each group of 64 floats (n a multiple of 64), yd[i] = (double)x[i ^ o] * 1.000000001: the
// VALU-only lane exchange every reduction of this library is built on (common.hpp: xor_lane), checked against index arithmetic.
extern "C" int vvae_selftest_xor_lane(const float* x, float* y, double* yd, int n, int o, void* stream)
{
    if (!x || !y || !yd || n <= 0 || n % 64) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(n / 64), block(64);
    switch (o) {
        case 1: hipLaunchKernelGGL(xor_lane_selftest_kernel<1>, grid, block, 0, s, x, y, yd); break;
        case 2: hipLaunchKernelGGL(xor_lane_selftest_kernel<2>, grid, block, 0, s, x, y, yd); break;
        case 4: hipLaunchKernelGGL(xor_lane_selftest_kernel<4>, grid, block, 0, s, x, y, yd); break;
        case 8: hipLaunchKernelGGL(xor_lane_selftest_kernel<8>, grid, block, 0, s, x, y, yd); break;
        case 16: hipLaunchKernelGGL(xor_lane_selftest_kernel<16>, grid, block, 0, s, x, y, yd); break;
        case 32: hipLaunchKernelGGL(xor_lane_selftest_kernel<32>, grid, block, 0, s, x, y, yd); break;
        default: return VVAE_ERR_BAD_ARG;
    }
    VVAE_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------- SiLU stream
// a = silu(h) between the two Linear layers of the MLP (reference train/layers.py:186-189), 42 x 50 MB per step: one 16-byte vector
// per lane and iteration, 4 in flight per thread (the framework's elementwise kernel runs this at 4.0 TB/s).
namespace {
__global__ __launch_bounds__(256) void silu_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, long nvec)
{
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float f[8];
            VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&v[u]), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] * sigmoidf_(f[e]);
            VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(&v[u]), f);
            y[i + u * stride] = v[u];
        }
    }
    for (; i < nvec; i += stride) {
        uint4 v = x[i];
        float f[8];
        VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&v), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * sigmoidf_(f[e]);
        VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(&v), f);
        y[i] = v;
    }
}
}  // namespace

// y = silu(x), bf16, n elements (a multiple of 8), both 16-byte aligned and contiguous.
extern "C" int vvae_silu_bf16(const void* x, void* y, long n, void* stream)
{
    if (!x || !y || n <= 0 || n % 8 || ((uintptr_t)x % 16) || ((uintptr_t)y % 16)) return VVAE_ERR_BAD_ARG;
    const long nvec = n / 8;
    long blocks = (nvec + 256 * 4 - 1) / (256 * 4);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(silu_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, nvec);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------- transposed weight shadows
// dst_i (cols_i, rows_i) = src_i (rows_i, cols_i)^T for up to 64 bf16 matrices in one launch: the (out, in) copies of the Linear
// kernels that the own NT GEMM multiplies in the forward pass (fc1 + SiLU, out-projection + residual), refreshed once per optimizer
// step from the bf16 shadow the Adam kernel writes.  64 x 64 tiles through LDS, 16-byte accesses on both sides.
namespace {
constexpr int TR_MAX = 64;
struct TrEntry { const bf16_t* src; bf16_t* dst; int rows, cols, tiles_c, tile_start; };
// start[] apart from the entries: the search for a workgroup's matrix is then four wide scalar loads, not 63 dependent ones
struct TrArgs { int start[TR_MAX]; TrEntry e[TR_MAX]; int n; };

__global__ __launch_bounds__(256) void transpose_grouped_kernel(TrArgs g)
{
    __shared__ bf16_t tile[64][64 + 8];
    int ei = 0;
#pragma unroll
    for (int i = 1; i < TR_MAX; ++i) ei += (int)blockIdx.x >= g.start[i] ? 1 : 0;     // start[] is non-decreasing, INT_MAX behind n
    const TrEntry& E = g.e[ei];
    const int t = blockIdx.x - E.tile_start;
    const int r0 = (t / E.tiles_c) * 64, c0 = (t % E.tiles_c) * 64;
    const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;            // 32 rows x 8 chunks of 8 elements per pass
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int r = tr + 32 * ps;
        const uint4 v = *reinterpret_cast<const uint4*>(E.src + (long)(r0 + r) * E.cols + c0 + tc);
        *reinterpret_cast<uint4*>(&tile[r][tc]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int c = tr + 32 * ps;                                          // output row = source column
        bf16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = tile[tc + e][c];
        *reinterpret_cast<uint4*>(E.dst + (long)(c0 + c) * E.rows + r0 + tc) = *reinterpret_cast<const uint4*>(o);
    }
}
}  // namespace

// n <= 64 matrices, rows_i and cols_i multiples of 64, 16-byte aligned, contiguous.  Host arrays of device pointers / ints.
extern "C" int vvae_transpose_grouped_bf16(const void* const* src, void* const* dst, const int* rows, const int* cols, int n, void* stream)
{
    if (!src || !dst || !rows || !cols || n <= 0 || n > TR_MAX) return VVAE_ERR_BAD_ARG;
    TrArgs g;
    g.n = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i] || rows[i] <= 0 || cols[i] <= 0 || rows[i] % 64 || cols[i] % 64 || ((uintptr_t)src[i] % 16) || ((uintptr_t)dst[i] % 16))
            return VVAE_ERR_BAD_ARG;
        g.e[i] = TrEntry{(const bf16_t*)src[i], (bf16_t*)dst[i], rows[i], cols[i], cols[i] / 64, tiles};
        g.start[i] = tiles;
        tiles += (rows[i] / 64) * (cols[i] / 64);
    }
    for (int i = n; i < TR_MAX; ++i) g.start[i] = 0x7fffffff;
    hipLaunchKernelGGL(transpose_grouped_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, g);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// Measurement aid (ops.KernelTimer): occupy the stream for ~us microseconds with one wave that watches the 100 MHz realtime counter.
// bench.py brackets a tagged launch with two HIP events; on an idle stream the first event's timestamp is taken the moment it is
// enqueued, microseconds before the host has finished enqueueing the kernel behind it, and the interval reads kernel + host launch
// latency.  Behind this gate the event, the kernel and the closing event are all queued before the first of them executes, so the
// interval is the kernel's (what rocprofv3 reports, and what a replayed graph's back-to-back launches see).
namespace {
__global__ void delay_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

extern "C" int vvae_delay_us(int us, void* stream)
{
    if (us <= 0 || us > 1000) return VVAE_ERR_BAD_ARG;
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)us * 100ull);
    VVAE_LAUNCH_CHECK();
    return 0;
}
