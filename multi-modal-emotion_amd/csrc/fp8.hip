// FP8 (OCP e4m3) operand preparation for the fp8 GEMM path (BASELINE config 5: videomae-large, 32 frames).
//
// Per-tensor "current" scaling, sync free: amax -> scale = 448 / amax lives in a device scalar that the quantiser and the GEMM epilogue
// read, so nothing returns to the host and the whole step stays graph-capturable.
//   tav_fp8_amax      x [rows, cols] (f32 / bf16, row stride ld) -> scales[0] = 448/amax (quantisation scale), scales[1] = amax/448
//                     (dequantisation factor the GEMM multiplies back), scales[2] = amax
//   tav_fp8_quantize  q[r][c] = e4m3(x[r][c] * scales[0])  and, optionally, the TRANSPOSED copy qt[c][r] with the token axis padded with
//                     zeros to rows_pad: the weight-gradient dW = dY^T X is then the same NT GEMM (K = tokens) as every other product,
//                     so one fp8 kernel serves forward, dgrad and wgrad
//   tav_splitk_reduce out = sum_s slabs[s]   (the token axis of a wgrad is split over workgroups; fixed order, no atomics)
// v_cvt_pk_fp8_f32 rounds to nearest even (OCP e4m3fn on gfx950; MI300's fnuz encoding is not used anywhere); saturation to +-448 is done in
// software (pack4_fp8): the instruction turns out-of-range inputs into NaN.
#include "common.h"
#include "tavhip_internal.h"

namespace tav {

constexpr float FP8_MAX = 448.0f;

template <typename T>
__global__ __launch_bounds__(256) void fp8_amax_partial_kernel(const T* __restrict__ x, float* __restrict__ part, long rows, int cols4, long ld) {
    // grid-stride over 16-B (f32) / 8-B (bf16) groups of 4 elements; one partial per workgroup
    __shared__ float red[4];
    const long n4 = rows * cols4;
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long r = i / cols4; const int c = (int)(i - r * cols4) * 4;
        const f32x4 v = ld4(x + r * ld + c);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void fp8_amax_final_kernel(const float* __restrict__ part, int n, float* __restrict__ scales) {
    __shared__ float red[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, part[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const bool ok = m > 0.f && m < 3.0e38f;              // all-zero (or non-finite) tensors quantise with scale 1
        scales[0] = ok ? FP8_MAX / m : 1.f;
        scales[1] = ok ? m / FP8_MAX : 1.f;
        scales[2] = m;
    }
}

TAV_DEV uint32_t pack4_fp8(f32x4 v) {
    // clamp first: v_cvt_pk_fp8_f32 does NOT saturate on this target as configured (measured round 4: an input beyond +-448 converts to NaN, 0x7f),
    // and under delayed scaling a tensor may outgrow the maximum its scale was made from
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_fmed3f(v[k], -FP8_MAX, FP8_MAX);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
    return (uint32_t)w;
}

// 64 x 64 tiles: row-major copy with 4-byte stores (64-B row segments) and, through LDS, the transposed copy.
// DELAYED (ABI v5, tav_fp8_quantize_delayed): `scales` is a 4-float STATE {448/amax, amax/448, amax, running amax of THIS pass}: the tensor is
// quantised with the scale the previous step left there and its own absolute maximum is gathered on the way (one atomic max per wave on
// state[3], as integer bits: |x| >= 0) for tav_fp8_roll_states to turn into the next step's scale -- ONE pass over the tensor (2 B read + 1 B
// written per element) instead of the amax passes + the quantiser (2 + 2 + 1), and one launch instead of three.  Saturating conversion: a
// value that outgrew last step's maximum clips to +-448.
template <typename T, bool DELAYED>
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const T* __restrict__ x, float* __restrict__ scales, uint8_t* __restrict__ q,
                                                           uint8_t* __restrict__ qt, int rows, int cols, long ld, long ld_q, long ld_qt, int rows_pad) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[64][68];
    const float sc = scales[0];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int cq = threadIdx.x & 15, rh = threadIdx.x >> 4;           // 16 column groups of 4, 16 row slots
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + rh + 16 * i, c = c0 + 4 * cq;
        uint32_t w = 0;
        if (r < rows && c < cols) {
            const f32x4 raw = ld4(x + (long)r * ld + c);
            if constexpr (DELAYED) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(raw[0]), fabsf(raw[1])), fmaxf(fabsf(raw[2]), fabsf(raw[3]))));
            w = pack4_fp8(raw * sc);
            if (q) *reinterpret_cast<uint32_t*>(q + (long)r * ld_q + c) = w;
        }
        *reinterpret_cast<uint32_t*>(&tile[rh + 16 * i][4 * cq]) = w;   // rows past the end stay zero: they are the K padding of qt
    }
    if constexpr (DELAYED) {
        // One atomic max per wave on ONE address would be 187 000 serialised atomics for a [46 832, 4096] tensor (~2 ms: round 4 measured the step at
        // 282 ms that way).  Look first: a wave whose maximum does not beat the running one -- all but a handful, the running maximum of N tiles is
        // beaten ~ln N times -- leaves it alone.  The look is a relaxed agent-scope load (a stale value only costs a needless atomic, never a lost one).
        amax = wave_max(amax);
        if ((threadIdx.x & 63) == 0 && amax > 0.f && amax < 3.0e38f) {
            unsigned* slot = reinterpret_cast<unsigned*>(scales + 3);
            const unsigned mine = __float_as_uint(amax);
            if (mine > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, mine);
        }
    }
    if (!qt) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + rh + 16 * i, r = r0 + 4 * cq;
        if (c < cols && r < rows_pad) {
            const int cl = rh + 16 * i;
            const uint32_t w = (uint32_t)tile[4 * cq][cl] | ((uint32_t)tile[4 * cq + 1][cl] << 8) | ((uint32_t)tile[4 * cq + 2][cl] << 16) |
                               ((uint32_t)tile[4 * cq + 3][cl] << 24);
            *reinterpret_cast<uint32_t*>(qt + (long)c * ld_qt + r) = w;
        }
    }
}

// Row-major copy only (no transposed copy: every activation / gradient of the default fp8 policy): a streaming kernel -- 16-B loads of 8 bf16 (or two
// of 4 f32), four groups in flight per thread, 8-B stores -- instead of the 64 x 64 transposing tiles above, whose 8-B loads ran at ~3 TB/s.
template <typename T> TAV_DEV void ld8(const T* p, f32x4& a, f32x4& b);
template <> TAV_DEV void ld8<float>(const float* p, f32x4& a, f32x4& b) { a = ld4(p); b = ld4(p + 4); }
template <> TAV_DEV void ld8<bf16>(const bf16* p, f32x4& a, f32x4& b) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w = *reinterpret_cast<const u32x4*>(p);
    a = f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
    b = f32x4{__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xffff0000u), __uint_as_float(w.w << 16), __uint_as_float(w.w & 0xffff0000u)};
}
template <typename T, bool DELAYED>
__global__ __launch_bounds__(256) void fp8_quantize_stream_kernel(const T* __restrict__ x, float* __restrict__ scales, uint8_t* __restrict__ q, unsigned rows,
                                                                  unsigned cols8, long ld, long ld_q) {
    const float sc = scales[0];
    const unsigned n8 = rows * cols8, stride = gridDim.x * 256u;
    float amax = 0.f;
    constexpr int U = 4;
    for (unsigned i0 = blockIdx.x * 256u + threadIdx.x; i0 < n8; i0 += U * stride) {
        f32x4 a[U], b[U];
        long qo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + u * stride;
            qo[u] = -1;
            if (i < n8) {
                const unsigned r = i / cols8, c = (i - r * cols8) * 8u;
                ld8<T>(x + (long)r * ld + c, a[u], b[u]);
                qo[u] = (long)r * ld_q + c;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (qo[u] < 0) continue;
            if constexpr (DELAYED) {
                const f32x4 m = __builtin_elementwise_max(__builtin_elementwise_abs(a[u]), __builtin_elementwise_abs(b[u]));
                amax = fmaxf(amax, fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])));
            }
            uint2 w;
            w.x = pack4_fp8(a[u] * sc); w.y = pack4_fp8(b[u] * sc);
            *reinterpret_cast<uint2*>(q + qo[u]) = w;
        }
    }
    if constexpr (DELAYED) {
        amax = wave_max(amax);
        if ((threadIdx.x & 63) == 0 && amax > 0.f && amax < 3.0e38f) {      // (as in fp8_quantize_kernel: look before the atomic)
            unsigned* slot = reinterpret_cast<unsigned*>(scales + 3);
            const unsigned mine = __float_as_uint(amax);
            if (mine > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, mine);
        }
    }
}
template <typename T, bool DELAYED>
static bool quantize_stream(const void* x, int64_t rows, int64_t cols, int64_t ld, float* scales, void* q, int64_t ld_q, hipStream_t st) {
    // the streaming form needs 16-B aligned groups of 8 and 32-bit group indices
    if (cols % 8 || ld % 8 || ld_q % 8 || rows * (cols / 8) >= (1ll << 31) || ((uintptr_t)x & 15) || ((uintptr_t)q & 7)) return false;
    const long n8 = rows * (cols / 8);
    long nb = (n8 + 256 * 4 - 1) / (256 * 4);
    nb = nb < 1 ? 1 : (nb > 8192 ? 8192 : nb);
    hipLaunchKernelGGL((fp8_quantize_stream_kernel<T, DELAYED>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, scales, (uint8_t*)q, (unsigned)rows,
                       (unsigned)(cols / 8), (long)ld, (long)ld_q);
    return true;
}

// every state whose running maximum moved this step takes it over as its scale; the others (unused this step, or just calibrated) stay
__global__ void fp8_roll_states_kernel(float* __restrict__ states, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* st = states + 4l * i;
    const float m = st[3];
    if (m > 0.f && m < 3.0e38f) { st[0] = FP8_MAX / m; st[1] = m / FP8_MAX; st[2] = m; }
    st[3] = 0.f;
}

__global__ void fp8_splitk_reduce_kernel(const float* __restrict__ S, float* __restrict__ out, int nsplit, long n4, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) v += ld4(S + ((long)s * n4 + i) * 4);
    if (accumulate) v += ld4(out + i * 4);
    st4(out + i * 4, v);
}

}  // namespace tav
using namespace tav;
#define ST ((hipStream_t)stream)

extern "C" int tav_fp8_amax_partials(int64_t rows, int64_t cols) {
    const long n4 = rows * (cols / 4);
    long nb = (n4 + 256 * 8 - 1) / (256 * 8);                 // >= 8 groups per thread
    return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
}
extern "C" int tav_fp8_amax(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld, float* partials, float* scales, void* stream) {
    if (!x || !partials || !scales) return TAV_ERR_NULL;
    if (rows <= 0 || cols <= 0 || cols % 4) return TAV_ERR_SHAPE;
    if (ld % 4) return TAV_ERR_ALIGN;
    const int nb = tav_fp8_amax_partials(rows, cols);
    if (dtype == TAV_BF16) hipLaunchKernelGGL((fp8_amax_partial_kernel<bf16>), dim3(nb), dim3(256), 0, ST, (const bf16*)x, partials, (long)rows, (int)(cols / 4), (long)ld);
    else if (dtype == TAV_F32) hipLaunchKernelGGL((fp8_amax_partial_kernel<float>), dim3(nb), dim3(256), 0, ST, (const float*)x, partials, (long)rows, (int)(cols / 4), (long)ld);
    else return TAV_ERR_DTYPE;
    hipLaunchKernelGGL(fp8_amax_final_kernel, dim3(1), dim3(256), 0, ST, (const float*)partials, nb, scales);
    return tav_last_error();
}
extern "C" int tav_fp8_quantize(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld, const float* scales, void* q, int64_t ld_q,
                                void* qt, int64_t ld_qt, int64_t rows_pad, void* stream) {
    if (!x || !scales || (!q && !qt)) return TAV_ERR_NULL;
    if (rows <= 0 || cols <= 0 || cols % 4) return TAV_ERR_SHAPE;
    if (ld % 4 || (q && ld_q % 4) || (qt && (ld_qt % 4 || rows_pad % 4 || rows_pad < rows || ld_qt < rows_pad))) return TAV_ERR_ALIGN;
    const long rp = qt ? rows_pad : rows;
    dim3 grid(tav_cdiv(cols, 64), tav_cdiv(rp, 64));
    float* sc = const_cast<float*>(scales);
    if (!qt && dtype == TAV_BF16 && quantize_stream<bf16, false>(x, rows, cols, ld, sc, q, ld_q, ST)) return tav_last_error();
    if (!qt && dtype == TAV_F32 && quantize_stream<float, false>(x, rows, cols, ld, sc, q, ld_q, ST)) return tav_last_error();
    if (dtype == TAV_BF16) hipLaunchKernelGGL((fp8_quantize_kernel<bf16, false>), grid, dim3(256), 0, ST, (const bf16*)x, sc, (uint8_t*)q, (uint8_t*)qt, (int)rows, (int)cols, (long)ld, (long)ld_q, (long)ld_qt, (int)rp);
    else if (dtype == TAV_F32) hipLaunchKernelGGL((fp8_quantize_kernel<float, false>), grid, dim3(256), 0, ST, (const float*)x, sc, (uint8_t*)q, (uint8_t*)qt, (int)rows, (int)cols, (long)ld, (long)ld_q, (long)ld_qt, (int)rp);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
extern "C" int tav_fp8_quantize_delayed(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld, float* state, void* q, int64_t ld_q,
                                        void* qt, int64_t ld_qt, int64_t rows_pad, void* stream) {
    if (!x || !state || (!q && !qt)) return TAV_ERR_NULL;
    if (rows <= 0 || cols <= 0 || cols % 4) return TAV_ERR_SHAPE;
    if (ld % 4 || (q && ld_q % 4) || (qt && (ld_qt % 4 || rows_pad % 4 || rows_pad < rows || ld_qt < rows_pad))) return TAV_ERR_ALIGN;
    const long rp = qt ? rows_pad : rows;
    dim3 grid(tav_cdiv(cols, 64), tav_cdiv(rp, 64));
    if (!qt && dtype == TAV_BF16 && quantize_stream<bf16, true>(x, rows, cols, ld, state, q, ld_q, ST)) return tav_last_error();
    if (!qt && dtype == TAV_F32 && quantize_stream<float, true>(x, rows, cols, ld, state, q, ld_q, ST)) return tav_last_error();
    if (dtype == TAV_BF16) hipLaunchKernelGGL((fp8_quantize_kernel<bf16, true>), grid, dim3(256), 0, ST, (const bf16*)x, state, (uint8_t*)q, (uint8_t*)qt, (int)rows, (int)cols, (long)ld, (long)ld_q, (long)ld_qt, (int)rp);
    else if (dtype == TAV_F32) hipLaunchKernelGGL((fp8_quantize_kernel<float, true>), grid, dim3(256), 0, ST, (const float*)x, state, (uint8_t*)q, (uint8_t*)qt, (int)rows, (int)cols, (long)ld, (long)ld_q, (long)ld_qt, (int)rp);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
extern "C" int tav_fp8_roll_states(float* states, int64_t n, void* stream) {
    if (!states) return TAV_ERR_NULL;
    if (n <= 0) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(fp8_roll_states_kernel, dim3(tav_cdiv(n, 256)), dim3(256), 0, ST, states, (int)n);
    return tav_last_error();
}
extern "C" int tav_splitk_reduce(const float* slabs, float* out, int32_t nsplit, int64_t n_elems, int32_t accumulate, void* stream) {
    if (!slabs || !out) return TAV_ERR_NULL;
    if (nsplit <= 0 || n_elems <= 0 || n_elems % 4) return TAV_ERR_SHAPE;
    const long n4 = n_elems / 4;
    hipLaunchKernelGGL(fp8_splitk_reduce_kernel, dim3(tav_cdiv(n4, 256)), dim3(256), 0, ST, slabs, out, nsplit, n4, accumulate);
    return tav_last_error();
}
