// Shared device helpers for the video-VAE HIP kernels (gfx950 / CDNA4 only).
//
// Storage dtypes: float and bf16 (raw uint16 bits).  All arithmetic is fp32;
// bf16 is a storage/operand format.  Tensors are channels-last
// (n, t, h, w, c) seen as (voxel, channel) rows with a row pitch `ld >= C`
// in elements, so channel slices of a wider buffer (the concat-elision
// buffers of UpBlock3D) are first-class operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VVAE_DT_F32 0
#define VVAE_DT_BF16 1

#define VVAE_ERR_BAD_ARG 1001     // outside hipError_t's range
#define VVAE_ERR_WORKSPACE 1002
#define VVAE_ERR_LIBRARY 1003     // a library call (hipBLASLt) failed

typedef uint16_t bf16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// f32 -> bf16, round-to-nearest-even, NaN stays NaN (a plain cast lowers to v_cvt_pk_bf16_f32).
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = f2bf(v); }

// Round a fp32 value to the storage dtype and back (what the next consumer will see).
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf2f(f2bf(v)); }
// Two values at once: one v_cvt_pk_bf16_f32 rounds both (a scalar round_to spends one on each), then a shift and a mask unpack.
template <typename T> __device__ __forceinline__ void round2(float& a, float& b);
template <> __device__ __forceinline__ void round2<float>(float&, float&) {}
template <> __device__ __forceinline__ void round2<bf16_t>(float& a, float& b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    const uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    a = __uint_as_float(w << 16);
    b = __uint_as_float(w & 0xffff0000u);
}

// s_barrier that hipcc's scheduler may not move instructions across.  MFMAs are register-only instructions and nothing in the language orders
// them against a barrier: in the ping-pong GEMM kernels (two wave groups alternating a read phase and an MFMA phase between raw s_barriers)
// hipcc sank most of a phase's MFMAs BELOW the barrier that closes the phase -- 2 of 36 above / 34 below in gemm_tn256, 4 / 44 in gemm_pp --
// so both groups' MFMAs met on the matrix pipe in one segment and the other segment had none (round 4, tools/pp_ablation.py).
__device__ __forceinline__ void vvae_phase_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ---- vector access: VEC channels per lane (VEC*sizeof(T) = 16 B when aligned) -------------
template <typename T, int VEC> struct VecIO;

template <> struct VecIO<float, 4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct VecIO<float, 1> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};
template <> struct VecIO<bf16_t, 8> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        uint4 t = *reinterpret_cast<const uint4*>(p);
        uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct VecIO<bf16_t, 4> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
        uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
        uint2 t;
        t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(p) = t;
    }
};
template <> struct VecIO<bf16_t, 1> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[1]) { v[0] = bf2f(*p); }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[1]) { *p = f2bf(v[0]); }
};

template <typename T> struct VecWidth;                       // widest vector = 16 bytes
template <> struct VecWidth<float> { static constexpr int value = 4; };
template <> struct VecWidth<bf16_t> { static constexpr int value = 8; };

// ---- cross-lane exchange on the VALU only (DPP + v_permlane16/32_swap), never through the LDS crossbar ---------------
// xor_lane<O>(v): the value lane (id ^ O) holds, O in {1, 2, 4, 8, 16, 32} -- what __shfl_xor(v, O, 64) returns.
// Why not __shfl_xor: hipcc lowers it to ds_bpermute_b32, an LDS-pipeline instruction (LGKM counter, ~50+ cycles of latency per
// butterfly stage, and it competes for the LDS port with the kernels that stage tiles there).  DPP moves and the permlane swaps are
// ordinary vector instructions: one or two issue slots per stage, no wait, bitwise the same data movement.
template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int O> __device__ __forceinline__ unsigned xor_lane_u32(unsigned v) {
    static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "one address bit at a time");
    if constexpr (O == 1) return dpp_u32<0xB1>(v);                       // quad_perm [1,0,3,2]
    else if constexpr (O == 2) return dpp_u32<0x4E>(v);                  // quad_perm [2,3,0,1]
    else if constexpr (O == 4) return dpp_u32<0x1B>(dpp_u32<0x141>(v));  // row_half_mirror (i -> 7-i), then quad_perm [3,2,1,0]: i -> i^4
    else if constexpr (O == 8) return dpp_u32<0x128>(v);                 // row_ror:8 (16-lane rows: (i + 8) % 16 = i ^ 8)
    else if constexpr (O == 16) {
        // v_permlane16_swap vdst, src: rows 1, 3 of vdst trade places with rows 0, 2 of src.  With both operands = v:
        // r[0] = rows {0,0,2,2} of v, r[1] = rows {1,1,3,3}: an odd row's partner is in r[0], an even row's in r[1].
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        // v_permlane32_swap: lanes 32-63 of vdst trade places with lanes 0-31 of src: r[0] = {lo, lo}, r[1] = {hi, hi}
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}
template <int O> __device__ __forceinline__ float xor_lane(float v) { return __uint_as_float(xor_lane_u32<O>(__float_as_uint(v))); }
template <int O> __device__ __forceinline__ double xor_lane(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = xor_lane_u32<O>((unsigned)u), hi = xor_lane_u32<O>((unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// Butterfly all-reduce over the lane-address bits HI, HI/2, ..., LO (powers of two): every lane ends with the sum over the lanes
// that differ from it only in those bits, bitwise identical in all of them (each step adds the same two numbers on both sides).
template <int HI, int LO, typename F> __device__ __forceinline__ F butterfly_sum(F v) {
    static_assert(HI >= LO && LO >= 1, "bit range");
    v += xor_lane<HI>(v);
    if constexpr (HI > LO) return butterfly_sum<HI / 2, LO>(v);
    else return v;
}
// the threads of a block must be laid out so that lane = threadIdx.x % 64 (every launch in this library is 1-D)

// ---- wave (64-lane) reductions -------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) { return butterfly_sum<32, 1>(v); }
__device__ __forceinline__ double wave_sum(double v) { return butterfly_sum<32, 1>(v); }

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (ten VALU instructions per element in the GroupNorm+SiLU kernels)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// host-side helpers
static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Zero `n` 32-bit words with a KERNEL (not hipMemsetAsync): inside a captured hipGraph a memset node followed by an
// atomically-accumulating kernel was observed to race on ROCm 7.2 (stale words in ~1 of 3 replays); a fill kernel is an
// ordinary kernel node with ordinary stream ordering.
__global__ static void vvae_zero_words_kernel(uint32_t* __restrict__ p, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t vvae_zero_async(void* p, size_t bytes, hipStream_t s)
{
    const long n = (long)(bytes / 4);
    if (n <= 0) return hipSuccess;
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(vvae_zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)p, n);
    return hipGetLastError();
}
// Fold per-workgroup partial rows in fixed order: out[c] = sum_r part[r * stride + c] for c < ncols; columns < n0 go to out0[c],
// the rest to out1[c - n0].  Block = 32 columns x 8 row lanes, 8 loads in flight per thread (the partials are L2-resident:
// latency, not bandwidth, is what the fold pays for).  grid = ceil(ncols / 32).
__global__ static void vvae_reduce_rows_kernel(const float* __restrict__ part, int rows, long stride, int ncols, float* __restrict__ out0, int n0,
                                               float* __restrict__ out1)
{
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < ncols) {
        for (int r0 = rl; r0 < rows; r0 += 8 * 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + 8 * i;
                v[i] = r < rows ? part[(long)r * stride + c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
        }
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < ncols) {
        const float t = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
        if (c < n0) out0[c] = t;
        else if (out1) out1[c - n0] = t;
    }
}

#define VVAE_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
