// wav2vec2 feature-encoder support kernels (HF wav2vec2/modeling_wav2vec2.py:275-419): activations are kept
// channels-last [B][T][C] so every conv after the first is a GEMM over overlapping rows (gemm.hip); this file holds
// the pieces that are not GEMMs: the C_in = 1 first conv, col2im for the strided input gradients, the zero-padded
// group-major copy for the grouped positional conv (:326-379) and its weight-norm reparametrisation.
#include "common.h"
#include "tavhip_internal.h"

namespace tav {

constexpr int C0_TT = 128;     // output steps per workgroup (conv0 forward): enough stores to amortise the weight loads of the prologue
constexpr int C0_BT = 512;     // output steps per workgroup of the conv0 weight gradient (4 time groups x 128 steps): half the partial blocks of 256 for the final reduce
constexpr int C0_MAXK = 16;

// y[b][t][c] = sum_j x[b][s*t + j] * w[c][j] + bias[c];  thread owns 4 consecutive channels {4q .. 4q+3, q = tid, tid + 256, ...} of the output
// steps of its time group (one 8-B / 16-B store per step and thread; one channel per thread -- 2-byte stores -- wrote the 524 MB at 1.7 TB/s)
template <typename TD>
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const float* __restrict__ wave, const float* __restrict__ w, const float* __restrict__ bias,
                                                        TD* __restrict__ y, int T_in, int T_out, int C, int K, int stride) {
    __shared__ float xs[C0_TT * 8 + C0_MAXK];
    const int b = blockIdx.y, t0 = blockIdx.x * C0_TT;
    const int nx = C0_TT * stride + K;
    for (int i = threadIdx.x; i < nx; i += 256) {
        const int src = t0 * stride + i;
        xs[i] = src < T_in ? wave[(long)b * T_in + src] : 0.f;
    }
    __syncthreads();
    const int nq = C >> 2;                                   // channel quads (host checks C % 4 == 0)
    const int qpt = nq < 256 ? nq : 256;                      // quads handled side by side; the remaining threads split the time steps
    const int tgroups = 256 / qpt > 0 ? 256 / qpt : 1;
    const int q0 = threadIdx.x % qpt, tg = threadIdx.x / qpt;
    if (tg >= tgroups) return;
    for (int q = q0; q < nq; q += qpt) {
        const int c = 4 * q;
        float wr[4][C0_MAXK];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < C0_MAXK; ++j) wr[k][j] = j < K ? w[(long)(c + k) * K + j] : 0.f;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = ld4(bias + c);
        for (int tt = tg; tt < C0_TT; tt += tgroups) {
            const int t = t0 + tt;
            if (t >= T_out) break;
            f32x4 a = bv;
#pragma unroll
            for (int j = 0; j < C0_MAXK; ++j)
                if (j < K) {
                    const float xv = xs[tt * stride + j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[k] += xv * wr[k][j];
                }
            st4(y + ((long)b * T_out + t) * C + c, a);
        }
    }
}

// partial[block][c][K+1]: dw[c][j] = sum_t dy[t][c] x[s*t+j], last column = bias gradient.
// Workgroup = one chunk of C0_BT output steps x 512 channels: thread (cq, tg) owns channels 8cq..8cq+7 and the steps tg, tg+4, ... of
// the chunk, reading 8 channels per load (16 B of bf16): 64 x 16-B loads per thread instead of 256 x 2-B ones (the old form was latency
// bound at 0.5 TB/s).  The four time groups are summed through LDS.
template <typename TD>
__global__ __launch_bounds__(256) void conv0_bwd_w_kernel(const float* __restrict__ wave, const TD* __restrict__ dy, float* __restrict__ partial,
                                                          int T_in, int T_out, int C, int K, int stride, int nchunks) {
    __shared__ float xs[C0_BT * 8 + C0_MAXK];
    __shared__ float red[2][64][89];                        // [slot][cq][8 channels x 11 accumulators] (+1: bank spread)
    const int b = blockIdx.y, t0 = blockIdx.x * C0_BT;
    const int nx = C0_BT * stride + K;
    for (int i = threadIdx.x; i < nx; i += 256) {
        const int src = t0 * stride + i;
        xs[i] = src < T_in ? wave[(long)b * T_in + src] : 0.f;
    }
    __syncthreads();
    const long blk = (long)b * nchunks + blockIdx.x;
    const int cq = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int tmax = (T_out - t0) < C0_BT ? (T_out - t0) : C0_BT;
    constexpr int KA = 11;                                   // accumulators per channel: K taps (<= 10 here) + bias
    for (int c0 = cq * 8; c0 < C; c0 += 512) {
        float acc[8][KA];
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int j = 0; j < KA; ++j) acc[e][j] = 0.f;
#pragma unroll 2
        for (int tt = tg; tt < tmax; tt += 4) {
            const TD* src = dy + ((long)b * T_out + t0 + tt) * C + c0;
            const f32x4 d0 = ld4(src), d1 = ld4(src + 4);
            const float d[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
            for (int j = 0; j < KA - 1; ++j) {
                const float xv = j < K ? xs[tt * stride + j] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e][j] += d[e] * xv;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e][KA - 1] += d[e];
        }
        // tree over the four time groups (= waves): 2,3 -> 0,1, then 1 -> 0
        if (tg >= 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int j = 0; j < KA; ++j) red[tg - 2][cq][e * KA + j] = acc[e][j];
        }
        __syncthreads();
        if (tg < 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int j = 0; j < KA; ++j) acc[e][j] += red[tg][cq][e * KA + j];
        }
        __syncthreads();
        if (tg == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int j = 0; j < KA; ++j) red[0][cq][e * KA + j] = acc[e][j];
        }
        __syncthreads();
        if (tg == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float* o = partial + (blk * C + c0 + e) * (K + 1);
#pragma unroll
                for (int j = 0; j < KA; ++j) {
                    const float v = acc[e][j] + red[0][cq][e * KA + j];
                    if (j < K) o[j] = v;
                    else if (j == KA - 1) o[K] = v;
                }
            }
        }
        __syncthreads();
    }
}
// 64 outputs x 4 partial groups per workgroup (the partial index is the slow axis: coalesced over outputs)
__global__ __launch_bounds__(256) void conv0_bwd_w_final_kernel(const float* __restrict__ partial, float* dw, float* dbias, int nblocks, int C, int K,
                                                                int accumulate) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, rg = threadIdx.x >> 6, idx = blockIdx.x * 64 + l, n = C * (K + 1);
    float s = 0.f;
    if (idx < n)
        for (int k = rg; k < nblocks; k += 4) s += partial[(long)k * n + idx];
    red[rg][l] = s;
    __syncthreads();
    if (rg != 0 || idx >= n) return;
    s = red[0][l] + red[1][l] + red[2][l] + red[3][l];
    const int c = idx / (K + 1), j = idx - c * (K + 1);
    if (j < K) dw[c * K + j] = accumulate ? dw[c * K + j] + s : s;
    else if (dbias) dbias[c] = accumulate ? dbias[c] + s : s;
}

// dx[b][tau][c] = sum_j [ (tau-j) % s == 0 and 0 <= (tau-j)/s < T_out ] dcol[b][(tau-j)/s][j*C + c]   (* gelu'(pre))
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, const T* __restrict__ pre, long n4, int T_in, int T_out, int C, int K,
                              int stride) {
    const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= n4) return;
    const long e = i4 * 4; const int c = (int)(e % C); const long bt = e / C; const int b = (int)(bt / T_in), tau = (int)(bt - (long)b * T_in);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < K; ++j) {
        const int d = tau - j;
        if (d < 0) break;
        if (d % stride) continue;
        const int t = d / stride;
        if (t >= T_out) continue;
        a += ld4(dcol + ((long)b * T_out + t) * ((long)K * C) + (long)j * C + c);
    }
    if (pre) {
        const f32x4 u = ld4(pre + e);
        a[0] *= gelu_grad_t<T>(u[0]); a[1] *= gelu_grad_t<T>(u[1]); a[2] *= gelu_grad_t<T>(u[2]); a[3] *= gelu_grad_t<T>(u[3]);
    }
    st4(dx + e, a);
}

// xg[b][g][p][cg] = (pad_l <= p < pad_l + T) ? x[b][p - pad_l][g*Cg + cg] : 0       (TP = pad_l + T + pad_r)
template <typename TS, typename TD>
__global__ void group_pad_kernel(const TS* __restrict__ x, TD* __restrict__ xg, long n4, int Tn, int H, int G, int pad_l, int TP) {
    const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= n4) return;
    const int Cg = H / G;
    const long e = i4 * 4; const int cg = (int)(e % Cg); long r = e / Cg; const int pp = (int)(r % TP); r /= TP; const int g = (int)(r % G); const int b = (int)(r / G);
    const int t = pp - pad_l;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t >= 0 && t < Tn) v = ld4(x + ((long)b * Tn + t) * H + g * Cg + cg);
    st4(xg + e, v);
}

// ---- weight norm (dim=2): w[co][ci][k] = g[k] * v[co][ci][k] / ||v[:,:,k]||
__global__ __launch_bounds__(128) void wn_colsum_partial_kernel(const float* __restrict__ a, const float* __restrict__ bmat, float* __restrict__ partial,
                                                                long rows, int K, int rows_per_block) {
    // partial[block][k] = sum over rows of a[row][k] * (bmat ? bmat[row][k] : a[row][k]);  thread <-> k (K <= 128)
    const int k = threadIdx.x;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block; r1 = r1 < rows ? r1 : rows;
    float s = 0.f;
    if (k < K) {
        if (bmat) {
#pragma unroll 8
            for (long r = r0; r < r1; ++r) s += a[r * K + k] * bmat[r * K + k];
        } else {
#pragma unroll 8
            for (long r = r0; r < r1; ++r) { const float av = a[r * K + k]; s += av * av; }
        }
    }
    if (k < K) partial[(long)blockIdx.x * K + k] = s;
}
__global__ void wn_colsum_final_kernel(const float* __restrict__ partial, float* out, int nblocks, int K, int do_sqrt) {
    const int k = threadIdx.x;
    if (k >= K) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;            // up to 512 partial rows in one workgroup: four loads in flight, not a serial chain
    int i = 0;
    for (; i + 3 < nblocks; i += 4) {
        s0 += partial[(long)i * K + k]; s1 += partial[(long)(i + 1) * K + k]; s2 += partial[(long)(i + 2) * K + k]; s3 += partial[(long)(i + 3) * K + k];
    }
    for (; i < nblocks; ++i) s0 += partial[(long)i * K + k];
    const float s = (s0 + s1) + (s2 + s3);
    out[k] = do_sqrt ? sqrtf(s) : s;
}
// GEMM layouts for the grouped conv (H = G*Cg channels, K taps):
//   w      [G][Cg_out][K][Cg_in]        forward / wgrad operand: row (g, co'), column kk*Cg + ci'
//   w_flip [G][Cg_in][K][Cg_out]        dgrad operand:          row (g, ci'), column kk'*Cg + co', kk' = K-1-kk
template <typename TD>
__global__ void wn_apply_kernel(const float* __restrict__ v, const float* __restrict__ gk, const float* __restrict__ norms, TD* __restrict__ w,
                                TD* __restrict__ w_flip, int H, int Cg, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)H * Cg * K;
    if (idx >= n) return;
    const int co = (int)(idx / ((long)Cg * K)); const int rem = (int)(idx - (long)co * Cg * K); const int ci = rem / K, kk = rem - ci * K;
    const float val = gk[kk] * v[idx] / norms[kk];
    const int g = co / Cg, col = co - g * Cg;
    if (w) ET<TD>::st(w + (((long)g * Cg + col) * K + kk) * Cg + ci, val);
    if (w_flip) ET<TD>::st(w_flip + (((long)g * Cg + ci) * K + (K - 1 - kk)) * Cg + col, val);
}
// dw_gemm f32 [G][Cg_out][K][Cg_in] -> dw_nat f32 [H][Cg][K] (nn.Conv1d order), so the reductions below are coalesced
__global__ void wn_dw_to_nat_kernel(const float* __restrict__ dw_gemm, float* __restrict__ dw_nat, int H, int Cg, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)H * Cg * K;
    if (idx >= n) return;
    const int co = (int)(idx / ((long)Cg * K)); const int rem = (int)(idx - (long)co * Cg * K); const int ci = rem / K, kk = rem - ci * K;
    const int g = co / Cg, col = co - g * Cg;
    dw_nat[idx] = dw_gemm[(((long)g * Cg + col) * K + kk) * Cg + ci];
}
// dv = g/||v|| * (dw - v * <dw,v>_k / ||v||^2) ; dg[k] = <dw,v>_k / ||v||
__global__ void wn_bwd_apply_kernel(const float* __restrict__ v, const float* __restrict__ gk, const float* __restrict__ norms,
                                    const float* __restrict__ dw_nat, const float* __restrict__ dots, float* __restrict__ dv, float* __restrict__ dg,
                                    long n, int K, int accumulate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) {
        const int kk = (int)(idx % K);
        const float nr = norms[kk];
        const float val = gk[kk] / nr * (dw_nat[idx] - v[idx] * dots[kk] / (nr * nr));
        dv[idx] = accumulate ? dv[idx] + val : val;
    }
    if (idx < K) { const float val = dots[idx] / norms[idx]; dg[idx] = accumulate ? dg[idx] + val : val; }
}

}  // namespace tav
using namespace tav;
#define ST ((hipStream_t)stream)
#define G1(n) dim3(tav_cdiv((n), 256)), dim3(256), 0, ST

extern "C" int tav_conv0_fwd(const float* wave, const float* w, const float* bias, void* y, int32_t dt, int64_t B, int64_t T_in, int64_t T_out, int64_t C,
                             int64_t K, int64_t stride, void* stream) {
    if (!wave || !w || !y) return TAV_ERR_NULL;
    if (B <= 0 || T_in <= 0 || T_out <= 0 || C <= 0 || C % 4 || K <= 0 || K > C0_MAXK || stride <= 0 || stride > 8) return TAV_ERR_SHAPE;
    if ((T_out - 1) * stride + K > T_in) return TAV_ERR_SHAPE;
    dim3 grid(tav_cdiv(T_out, C0_TT), (unsigned)B);
    if (dt == TAV_BF16) hipLaunchKernelGGL((conv0_fwd_kernel<bf16>), grid, dim3(256), 0, ST, wave, w, bias, (bf16*)y, (int)T_in, (int)T_out, (int)C, (int)K, (int)stride);
    else if (dt == TAV_F32) hipLaunchKernelGGL((conv0_fwd_kernel<float>), grid, dim3(256), 0, ST, wave, w, bias, (float*)y, (int)T_in, (int)T_out, (int)C, (int)K, (int)stride);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
extern "C" int tav_conv0_bwd_partials(int64_t B, int64_t T_out, int64_t C, int64_t K) { return (int)(B * ((T_out + C0_BT - 1) / C0_BT) * C * (K + 1)); }
extern "C" int tav_conv0_bwd_w(const float* wave, const void* dy, int32_t dt, float* dw, float* dbias, float* partials, int64_t B, int64_t T_in, int64_t T_out,
                               int64_t C, int64_t K, int64_t stride, int32_t accumulate, void* stream) {
    if (!wave || !dy || !dw || !partials) return TAV_ERR_NULL;
    if (B <= 0 || T_in <= 0 || T_out <= 0 || C <= 0 || C % 8 || K <= 0 || K > 10 || stride <= 0 || stride > 8) return TAV_ERR_SHAPE;
    const int nchunks = (int)((T_out + C0_BT - 1) / C0_BT);
    dim3 grid(nchunks, (unsigned)B);
    if (dt == TAV_BF16) hipLaunchKernelGGL((conv0_bwd_w_kernel<bf16>), grid, dim3(256), 0, ST, wave, (const bf16*)dy, partials, (int)T_in, (int)T_out, (int)C, (int)K, (int)stride, nchunks);
    else if (dt == TAV_F32) hipLaunchKernelGGL((conv0_bwd_w_kernel<float>), grid, dim3(256), 0, ST, wave, (const float*)dy, partials, (int)T_in, (int)T_out, (int)C, (int)K, (int)stride, nchunks);
    else return TAV_ERR_DTYPE;
    hipLaunchKernelGGL(conv0_bwd_w_final_kernel, dim3(tav_cdiv(C * (K + 1), 64)), dim3(256), 0, ST, partials, dw, dbias, nchunks * (int)B, (int)C, (int)K, accumulate);
    return tav_last_error();
}
extern "C" int tav_col2im_1d(const void* dcol, void* dx, const void* pre_act, int32_t dt, int64_t B, int64_t T_in, int64_t T_out, int64_t C, int64_t K,
                             int64_t stride, void* stream) {
    if (!dcol || !dx) return TAV_ERR_NULL;
    if (B <= 0 || T_in <= 0 || T_out <= 0 || C <= 0 || C % 4 || K <= 0 || stride <= 0) return TAV_ERR_SHAPE;
    const long n4 = B * T_in * C / 4;
    if (dt == TAV_BF16) hipLaunchKernelGGL((col2im_kernel<bf16>), G1(n4), (const bf16*)dcol, (bf16*)dx, (const bf16*)pre_act, n4, (int)T_in, (int)T_out, (int)C, (int)K, (int)stride);
    else if (dt == TAV_F32) hipLaunchKernelGGL((col2im_kernel<float>), G1(n4), (const float*)dcol, (float*)dx, (const float*)pre_act, n4, (int)T_in, (int)T_out, (int)C, (int)K, (int)stride);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
extern "C" int tav_group_pad(const void* x, int32_t sdt, void* xg, int32_t ddt, int64_t B, int64_t T, int64_t H, int64_t G, int64_t pad_l, int64_t pad_r,
                             void* stream) {
    if (!x || !xg) return TAV_ERR_NULL;
    if (B <= 0 || T <= 0 || H <= 0 || G <= 0 || H % G || (H / G) % 4 || pad_l < 0 || pad_r < 0) return TAV_ERR_SHAPE;
    const int TP = (int)(pad_l + T + pad_r);
    const long n4 = B * G * TP * (H / G) / 4;
    if (sdt == TAV_F32 && ddt == TAV_BF16) hipLaunchKernelGGL((group_pad_kernel<float, bf16>), G1(n4), (const float*)x, (bf16*)xg, n4, (int)T, (int)H, (int)G, (int)pad_l, TP);
    else if (sdt == TAV_F32 && ddt == TAV_F32) hipLaunchKernelGGL((group_pad_kernel<float, float>), G1(n4), (const float*)x, (float*)xg, n4, (int)T, (int)H, (int)G, (int)pad_l, TP);
    else if (sdt == TAV_BF16 && ddt == TAV_BF16) hipLaunchKernelGGL((group_pad_kernel<bf16, bf16>), G1(n4), (const bf16*)x, (bf16*)xg, n4, (int)T, (int)H, (int)G, (int)pad_l, TP);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
extern "C" int tav_weight_norm_partials(int64_t H, int64_t Cg) { long r = H * Cg; long nb = (r + 255) / 256; return (int)(nb > 512 ? 512 : nb); }
extern "C" int tav_weight_norm_fwd(const float* v, const float* g, float* norms, float* partials, void* w, void* w_flip, int32_t dt, int64_t H, int64_t Cg,
                                   int64_t K, void* stream) {
    if (!v || !g || !norms || !partials || (!w && !w_flip)) return TAV_ERR_NULL;
    if (H <= 0 || Cg <= 0 || K <= 0 || K > 128 || H % Cg) return TAV_ERR_SHAPE;
    const long rows = H * Cg;
    const int nb = tav_weight_norm_partials(H, Cg);
    const int rpb = (int)((rows + nb - 1) / nb);
    hipLaunchKernelGGL(wn_colsum_partial_kernel, dim3(nb), dim3(128), 0, ST, v, (const float*)nullptr, partials, rows, (int)K, rpb);
    hipLaunchKernelGGL(wn_colsum_final_kernel, dim3(1), dim3(128), 0, ST, partials, norms, nb, (int)K, 1);
    const long n = rows * K;
    if (dt == TAV_BF16) hipLaunchKernelGGL((wn_apply_kernel<bf16>), G1(n), v, g, norms, (bf16*)w, (bf16*)w_flip, (int)H, (int)Cg, (int)K);
    else if (dt == TAV_F32) hipLaunchKernelGGL((wn_apply_kernel<float>), G1(n), v, g, norms, (float*)w, (float*)w_flip, (int)H, (int)Cg, (int)K);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
// workspace: dw_nat (H*Cg*K) + partials (tav_weight_norm_partials * K) + dots (K) floats
extern "C" int tav_weight_norm_bwd(const float* v, const float* g, const float* norms, const float* dw_gemm, float* workspace, float* dv, float* dg,
                                   int64_t H, int64_t Cg, int64_t K, int32_t accumulate, void* stream) {
    if (!v || !g || !norms || !dw_gemm || !workspace || !dv || !dg) return TAV_ERR_NULL;
    if (H <= 0 || Cg <= 0 || K <= 0 || K > 128 || H % Cg) return TAV_ERR_SHAPE;
    const long rows = H * Cg, n = rows * K;
    const int nb = tav_weight_norm_partials(H, Cg);
    const int rpb = (int)((rows + nb - 1) / nb);
    float* dw_nat = workspace; float* partials = workspace + n; float* dots = partials + (long)nb * K;
    hipLaunchKernelGGL(wn_dw_to_nat_kernel, G1(n), dw_gemm, dw_nat, (int)H, (int)Cg, (int)K);
    hipLaunchKernelGGL(wn_colsum_partial_kernel, dim3(nb), dim3(128), 0, ST, (const float*)dw_nat, v, partials, rows, (int)K, rpb);
    hipLaunchKernelGGL(wn_colsum_final_kernel, dim3(1), dim3(128), 0, ST, partials, dots, nb, (int)K, 0);
    hipLaunchKernelGGL(wn_bwd_apply_kernel, G1(n), v, g, norms, (const float*)dw_nat, (const float*)dots, dv, dg, n, (int)K, accumulate);
    return tav_last_error();
}
