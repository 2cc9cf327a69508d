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

// ---- wave (64-lane) reductions -------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

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
