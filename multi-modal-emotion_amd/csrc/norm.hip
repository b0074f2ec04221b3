// LayerNorm (one wave64 per row, shuffle reductions, f32 statistics) and the wav2vec2 "group" norm
// (GroupNorm with one group per channel == normalisation over time per (batch, channel)), both HBM-bound.
// Replaces nn.LayerNorm / nn.GroupNorm call-sites of the path: reference utils/TAVFormer.py:237,239,108,118,
// models/tav.py:439-447, HF roberta:336-340,394-398, HF wav2vec2:275-323,422-434,611-726, HF videomae:326-357.
#include "common.h"
#include "tavhip_internal.h"

namespace tav {

constexpr int LN_MAXV = 4;   // float4 vectors per lane: W <= 64*4*4 = 1024

struct LnP {
    const void* x; const float* gamma; const float* beta; float* y_f32; void* y_lp; float* mean; float* rstd;
    const void* dy; const float* dx_add; float* dx_f32; void* dx_lp; float* partials;
    long rows; int W; long ld_x, ld_y, ld_dy, ld_dx; float eps; int act;
};

template <typename TX, typename TL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = p.W >> 2;
    const float invW = 1.f / (float)p.W;
    // (requesting the next row before this row's reductions, as ln_bwd_kernel does, made this kernel SLOWER: 33.5 -> 36.9 us at 46848 x 768 --
    // with up to eight waves per SIMD and one row per wave in flight it already runs at 6.4 TB/s)
    for (long row = (long)blockIdx.x * 4 + wave; row < p.rows; row += (long)gridDim.x * 4) {
        const TX* xr = reinterpret_cast<const TX*>(p.x) + row * p.ld_x;
        f32x4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane + 64 * j;
            v[j] = (c < nv) ? ld4(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
        }
        const float mean = wave_sum(s) * invW;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane + 64 * j;
            if (c < nv) { const f32x4 d = v[j] - mean; q += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]; }
        }
        const float rstd = rsqrtf(wave_sum(q) * invW + p.eps);
        if (lane == 0) { if (p.mean) p.mean[row] = mean; if (p.rstd) p.rstd[row] = rstd; }
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane + 64 * j;
            if (c < nv) {
                f32x4 y = (v[j] - mean) * rstd * ld4(p.gamma + 4 * c) + ld4(p.beta + 4 * c);
                if (p.act == 1) { y[0] = gelu_t<TL>(y[0]); y[1] = gelu_t<TL>(y[1]); y[2] = gelu_t<TL>(y[2]); y[3] = gelu_t<TL>(y[3]); }
                if (p.y_f32) st4(p.y_f32 + row * p.ld_y + 4 * c, y);
                if (p.y_lp) st4(reinterpret_cast<TL*>(p.y_lp) + row * p.ld_y + 4 * c, y);
            }
        }
    }
}

template <typename TX, typename TDY, typename TL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnP p) {
    __shared__ float red[4][2][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = p.W >> 2;
    const float invW = 1.f / (float)p.W;
    f32x4 dg[LN_MAXV], db[LN_MAXV], gam[LN_MAXV];
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        dg[j] = f32x4{0.f, 0.f, 0.f, 0.f}; db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = lane + 64 * j;
        gam[j] = c < nv ? ld4(p.gamma + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};   // row invariant (the stores below may alias as far as the compiler knows)
    }
    // A wave walks its rows with the NEXT row's operands (x, dy, the residual gradient, mean / rstd) requested before the current row is
    // reduced and stored: one row per wave in flight left the kernel at 4.2 TB/s (two workgroups per CU, ~6 KB per wave and round trip, nothing
    // in flight during the two wave reductions and the stores); vmcnt retires in issue order, so the wait for a prefetched row never waits
    // for the stores issued after it.
    struct Row { f32x4 x[LN_MAXV], d[LN_MAXV], a[LN_MAXV]; float mean, rstd; };
    auto load_row = [&](long row, Row& r) __attribute__((always_inline)) {
        const TX* xr = reinterpret_cast<const TX*>(p.x) + row * p.ld_x;
        const TDY* dyr = reinterpret_cast<const TDY*>(p.dy) + row * p.ld_dy;
        r.mean = p.mean[row]; r.rstd = p.rstd[row];
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane + 64 * j;
            if (c < nv) {
                r.x[j] = ld4(xr + 4 * c);
                r.d[j] = ld4(dyr + 4 * c);
                if (p.dx_add) r.a[j] = ld4(p.dx_add + row * p.ld_dx + 4 * c);
            }
        }
    };
    const long stride = (long)gridDim.x * 4;
    long row = (long)blockIdx.x * 4 + wave;
    Row cur;
    if (row < p.rows) load_row(row, cur);
    for (; row < p.rows; row += stride) {
        Row nxt;
        if (row + stride < p.rows) load_row(row + stride, nxt);
        const float mean = cur.mean, rstd = cur.rstd;
        f32x4 xh[LN_MAXV], gdy[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane + 64 * j;
            if (c < nv) {
                xh[j] = (cur.x[j] - mean) * rstd;
                f32x4 d = cur.d[j];
                if (p.act == 1) {
                    const f32x4 z = xh[j] * gam[j] + ld4(p.beta + 4 * c);
                    d[0] *= gelu_grad_t<TL>(z[0]); d[1] *= gelu_grad_t<TL>(z[1]); d[2] *= gelu_grad_t<TL>(z[2]); d[3] *= gelu_grad_t<TL>(z[3]);
                }
                dg[j] += d * xh[j];
                db[j] += d;
                gdy[j] = d * gam[j];
                s1 += gdy[j][0] + gdy[j][1] + gdy[j][2] + gdy[j][3];
                const f32x4 t = gdy[j] * xh[j];
                s2 += t[0] + t[1] + t[2] + t[3];
            } else { xh[j] = f32x4{0.f, 0.f, 0.f, 0.f}; gdy[j] = xh[j]; }
        }
        const float m1 = wave_sum(s1) * invW, m2 = wave_sum(s2) * invW;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane + 64 * j;
            if (c < nv) {
                f32x4 dx = (gdy[j] - m1 - xh[j] * m2) * rstd;
                if (p.dx_add) dx += cur.a[j];
                if (p.dx_f32) st4(p.dx_f32 + row * p.ld_dx + 4 * c, dx);
                if (p.dx_lp) st4(reinterpret_cast<TL*>(p.dx_lp) + row * p.ld_dx + 4 * c, dx);
            }
        }
        cur = nxt;
    }
    if (!p.partials) return;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[wave][0][4 * c + e] = dg[j][e]; red[wave][1][4 * c + e] = db[j][e]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.W; c += 256) {
        p.partials[((long)blockIdx.x * 2 + 0) * p.W + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
        p.partials[((long)blockIdx.x * 2 + 1) * p.W + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
    }
}

// one workgroup per 16 columns: 4 column quads (16-B loads) x 64 row groups, fixed summation order (deterministic).  (12 workgroups of 64
// columns with 4-byte loads took 10 us for 512 partial rows -- 106 launches per step.)
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ partials, float* dgamma, float* dbeta, int nblocks, int W,
                                                              int accumulate) {
    __shared__ float red[64][16][2];
    const int cq = threadIdx.x & 3, rg = threadIdx.x >> 2, c = blockIdx.x * 16 + 4 * cq;
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, b = g;
    if (c < W) {
#pragma unroll 4
        for (int k = rg; k < nblocks; k += 64) { g += ld4(partials + ((long)k * 2 + 0) * W + c); b += ld4(partials + ((long)k * 2 + 1) * W + c); }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[rg][4 * cq + e][0] = g[e]; red[rg][4 * cq + e][1] = b[e]; }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int l = threadIdx.x, cc = blockIdx.x * 16 + l;
        if (cc < W) {
            float sg = 0.f, sb = 0.f;
#pragma unroll 8
            for (int k = 0; k < 64; ++k) { sg += red[k][l][0]; sb += red[k][l][1]; }
            if (dgamma) dgamma[cc] = accumulate ? dgamma[cc] + sg : sg;
            if (dbeta) dbeta[cc] = accumulate ? dbeta[cc] + sb : sb;
        }
    }
}

// Deferred form (ABI v5): a step has ~210 LayerNorm backwards, each followed by its own 48-workgroup reduce launch (4-5 us + the gap around
// it: 2.5 ms of serial time per step for 3 MB each).  Nobody reads dgamma / dbeta before the optimizer, so the host may keep every
// LayerNorm's partial slab and reduce up to 64 of them in ONE launch at the end of the backward (ops.ln_flush): grid.y = item.
struct LnReduceBatch { tav_ln_reduce_item it[TAV_LN_REDUCE_MAX]; };
__global__ __launch_bounds__(256) void ln_param_reduce_multi_kernel(const LnReduceBatch b) {
    __shared__ float red[64][16][2];
    const tav_ln_reduce_item& it = b.it[blockIdx.y];
    const int W = it.W, nblocks = it.nblocks;
    if ((int)blockIdx.x * 16 >= W) return;                       // (grid.x covers the widest item; uniform per workgroup)
    const float* __restrict__ partials = it.partials;
    const int cq = threadIdx.x & 3, rg = threadIdx.x >> 2, c = blockIdx.x * 16 + 4 * cq;
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, bsum = g;
    if (c < W) {
#pragma unroll 4
        for (int k = rg; k < nblocks; k += 64) { g += ld4(partials + ((long)k * 2 + 0) * W + c); bsum += ld4(partials + ((long)k * 2 + 1) * W + c); }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[rg][4 * cq + e][0] = g[e]; red[rg][4 * cq + e][1] = bsum[e]; }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int l = threadIdx.x, cc = blockIdx.x * 16 + l;
        if (cc < W) {
            float sg = 0.f, sb = 0.f;
#pragma unroll 8
            for (int k = 0; k < 64; ++k) { sg += red[k][l][0]; sb += red[k][l][1]; }      // same summation order as ln_param_reduce_kernel: bitwise equal results
            if (it.dgamma) it.dgamma[cc] = it.accumulate ? it.dgamma[cc] + sg : sg;
            if (it.dbeta) it.dbeta[cc] = it.accumulate ? it.dbeta[cc] + sb : sb;
        }
    }
}

// cap on workgroups of the LN backward (= rows of the dgamma/dbeta partial buffer).  Measured at 11712 x 768: 256 -> 35.5 us,
// 512 -> 33.9 us, 1024 -> 38.0 us (more partial rows for ln_param_reduce); at 46848 x 768 (batch 32): 256 -> 157 us, 512 -> 106.5 us
// (4.7 TB/s), 1024 -> 131 us, 2048 -> 136 us: one workgroup per CU keeps too few bytes in flight, four pay for their partial rows.
#ifndef LN_MAX_BLOCKS
#define LN_MAX_BLOCKS 512
#endif
static inline int ln_blocks(long rows) {
    long b = (rows + 15) / 16;      // >= 4 rows per wave so the per-lane dgamma/dbeta partials amortise
    return (int)(b < 1 ? 1 : (b > LN_MAX_BLOCKS ? LN_MAX_BLOCKS : b));
}

// ------------------------------------------------------------------------------------------------ group norm over time
// x [B][T][C] channels-last.  stats pass: partial sums over a T-slice for 64 channels; apply pass element-wise.
constexpr int GN_SPLIT = 32;
constexpr int GN_CB = 256;       // channels per workgroup of the statistics pass

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ stats, float* part, int Tn, int C) {
    // thread = 4 channels (one 8-B / 16-B load) x one of 4 row groups; a workgroup covers GN_CB = 256 channels, so a wave instruction reads
    // 512 contiguous bytes of one row.  (One channel per thread and 64 channels per workgroup -- 2-byte loads in 128-B pieces at a 1-KB
    // stride -- ran at 2.1-2.6 TB/s on the 524 MB activation of the wav2vec2 front-end.)
    __shared__ float red[4][GN_CB][2];
    const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6, b = blockIdx.z, sp = blockIdx.y;
    const int c = blockIdx.x * GN_CB + 4 * cq;
    const bool live = c < C;                                 // (C % 64 == 0: a partly filled last block has whole quads)
    const int per = (Tn + GN_SPLIT - 1) / GN_SPLIT, t0 = sp * per;
    int t1 = t0 + per; t1 = t1 < Tn ? t1 : Tn;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    f32x4 mean = a0, rstd = a0, gam = a0, bet = a0;
    if (BWD && live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { mean[k] = stats[((long)b * C + c + k) * 2]; rstd[k] = stats[((long)b * C + c + k) * 2 + 1]; gam[k] = gamma[c + k]; bet[k] = beta[c + k]; }
    }
    auto one = [&](f32x4 v, f32x4 dyv) __attribute__((always_inline)) {
        if (!BWD) { a0 += v; a1 += v * v; }
        else {
            const f32x4 xh = (v - mean) * rstd;
            f32x4 dz;
#pragma unroll
            for (int k = 0; k < 4; ++k) dz[k] = dyv[k] * gelu_grad_t<T>(xh[k] * gam[k] + bet[k]);
            a0 += dz; a1 += dz * xh;
        }
    };
    if (live) {
        constexpr int UB = BWD ? 4 : 8;                      // rows requested before the first is used
        const long rs = 4l * C;
        long e = ((long)b * Tn + t0 + rg) * C + c;
        int t = t0 + rg;
        for (; t + 4 * (UB - 1) < t1; t += 4 * UB, e += UB * rs) {
            f32x4 v[UB], d[UB];
#pragma unroll
            for (int q = 0; q < UB; ++q) { v[q] = ld4(x + e + q * rs); if (BWD) d[q] = ld4(dy + e + q * rs); else d[q] = v[q]; }
#pragma unroll
            for (int q = 0; q < UB; ++q) one(v[q], d[q]);
        }
        for (; t < t1; t += 4, e += rs) one(ld4(x + e), BWD ? ld4(dy + e) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[rg][4 * cq + k][0] = a0[k]; red[rg][4 * cq + k][1] = a1[k]; }
    __syncthreads();
    {
        const int l = threadIdx.x, cc = blockIdx.x * GN_CB + l;
        if (cc < C) {
            part[(((long)b * GN_SPLIT + sp) * C + cc) * 2 + 0] = red[0][l][0] + red[1][l][0] + red[2][l][0] + red[3][l][0];
            part[(((long)b * GN_SPLIT + sp) * C + cc) * 2 + 1] = red[0][l][1] + red[1][l][1] + red[2][l][1] + red[3][l][1];
        }
    }
}
// fwd: stats[b][c] = (mean, rstd);   bwd: sums[b][c] = (sum dz, sum dz*xhat)
__global__ void gn_finalize_kernel(const float* __restrict__ part, float* out, int Tn, int C, int B, float eps, int fwd) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * C) return;
    const int b = (int)(idx / C), c = (int)(idx - (long)b * C);
    float s0 = 0.f, s1 = 0.f;
    for (int sp = 0; sp < GN_SPLIT; ++sp) { s0 += part[(((long)b * GN_SPLIT + sp) * C + c) * 2]; s1 += part[(((long)b * GN_SPLIT + sp) * C + c) * 2 + 1]; }
    if (fwd) {
        const float mean = s0 / Tn;
        float var = s1 / Tn - mean * mean; var = var > 0.f ? var : 0.f;
        out[idx * 2] = mean; out[idx * 2 + 1] = rsqrtf(var + eps);
    } else { out[idx * 2] = s0; out[idx * 2 + 1] = s1; }
}
// Apply passes: thread = 4 channels (one 8-B / 16-B access per row) x GN_AR rows at a stride of 4, so the per-(batch, channel) statistics
// and the affine parameters are loaded ONCE per thread as float4s and four rows' loads are in flight per thread.  (One element quad per
// thread with 12-24 scalar parameter loads beside its one useful load ran at 3.8 TB/s on the 524 MB activation.)
constexpr int GN_AR = 16;        // rows per thread of the apply passes (a workgroup covers 4 * GN_AR rows x 256 channels)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ stats, int Tn, int C) {
    const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6, b = blockIdx.z;
    const int c = blockIdx.x * GN_CB + 4 * cq;
    if (c >= C) return;
    const f32x4 s0 = ld4(stats + ((long)b * C + c) * 2), s1 = ld4(stats + ((long)b * C + c) * 2 + 4);   // (mean, rstd) x 4 channels
    const f32x4 mean = {s0[0], s0[2], s1[0], s1[2]}, gam = ld4(gamma + c), bet = ld4(beta + c);
    const f32x4 sc = f32x4{s0[1], s0[3], s1[1], s1[3]} * gam;
    const int t0 = blockIdx.y * (4 * GN_AR) + rg;
    const long e0 = ((long)b * Tn + t0) * C + c, rs = 4l * C;                 // this thread's first element, elements per row step
    if (t0 + 4 * (GN_AR - 1) < Tn) {                                          // whole block of rows: four rows' loads issued before the first use
#pragma unroll
        for (int r0 = 0; r0 < GN_AR; r0 += 4) {
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ld4(x + e0 + (r0 + q) * rs);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 u = (v[q] - mean) * sc + bet;
                u[0] = gelu_t<T>(u[0]); u[1] = gelu_t<T>(u[1]); u[2] = gelu_t<T>(u[2]); u[3] = gelu_t<T>(u[3]);
                st4(y + e0 + (r0 + q) * rs, u);
            }
        }
    } else {
        for (int r = 0; r < GN_AR && t0 + 4 * r < Tn; ++r) {
            f32x4 u = (ld4(x + e0 + r * rs) - mean) * sc + bet;
            u[0] = gelu_t<T>(u[0]); u[1] = gelu_t<T>(u[1]); u[2] = gelu_t<T>(u[2]); u[3] = gelu_t<T>(u[3]);
            st4(y + e0 + r * rs, u);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ stats, const float* __restrict__ sums, int Tn, int C) {
    const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6, b = blockIdx.z;
    const int c = blockIdx.x * GN_CB + 4 * cq;
    if (c >= C) return;
    const f32x4 s0 = ld4(stats + ((long)b * C + c) * 2), s1 = ld4(stats + ((long)b * C + c) * 2 + 4);
    const f32x4 m0 = ld4(sums + ((long)b * C + c) * 2), m1 = ld4(sums + ((long)b * C + c) * 2 + 4);     // (sum dz, sum dz * xhat) x 4 channels
    const float invT = 1.f / Tn;
    const f32x4 mean = {s0[0], s0[2], s1[0], s1[2]}, rstd = {s0[1], s0[3], s1[1], s1[3]}, gam = ld4(gamma + c), bet = ld4(beta + c);
    const f32x4 k0 = f32x4{m0[0], m0[2], m1[0], m1[2]} * invT, k1 = f32x4{m0[1], m0[3], m1[1], m1[3]} * invT, og = rstd * gam;
    const int t0 = blockIdx.y * (4 * GN_AR) + rg;
    const long e0 = ((long)b * Tn + t0) * C + c, rs = 4l * C;
    auto one = [&](f32x4 xv, f32x4 dyv, long e) __attribute__((always_inline)) {
        const f32x4 xh = (xv - mean) * rstd, z = xh * gam + bet;
        f32x4 dz;
#pragma unroll
        for (int k = 0; k < 4; ++k) dz[k] = dyv[k] * gelu_grad_t<T>(z[k]);
        st4(dx + e, og * (dz - k0 - xh * k1));
    };
    if (t0 + 4 * (GN_AR - 1) < Tn) {
#pragma unroll
        for (int r0 = 0; r0 < GN_AR; r0 += 4) {
            f32x4 xv[4], dv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { xv[q] = ld4(x + e0 + (r0 + q) * rs); dv[q] = ld4(dy + e0 + (r0 + q) * rs); }
#pragma unroll
            for (int q = 0; q < 4; ++q) one(xv[q], dv[q], e0 + (r0 + q) * rs);
        }
    } else {
        for (int r = 0; r < GN_AR && t0 + 4 * r < Tn; ++r) one(ld4(x + e0 + r * rs), ld4(dy + e0 + r * rs), e0 + r * rs);
    }
}
__global__ void gn_param_grad_kernel(const float* __restrict__ sums, float* dgamma, float* dbeta, int B, int C, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float g = 0.f, bsum = 0.f;
    for (int b = 0; b < B; ++b) { bsum += sums[((long)b * C + c) * 2]; g += sums[((long)b * C + c) * 2 + 1]; }
    dgamma[c] = accumulate ? dgamma[c] + g : g;
    dbeta[c] = accumulate ? dbeta[c] + bsum : bsum;
}

template <typename T> __global__ void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = ld4(x + 4 * i);
    v[0] = gelu_t<T>(v[0]); v[1] = gelu_t<T>(v[1]); v[2] = gelu_t<T>(v[2]); v[3] = gelu_t<T>(v[3]);
    st4(y + 4 * i, v);
}
template <typename T> __global__ void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = ld4(x + 4 * i);
    f32x4 d = ld4(dy + 4 * i);
    d[0] *= gelu_grad_t<T>(v[0]); d[1] *= gelu_grad_t<T>(v[1]); d[2] *= gelu_grad_t<T>(v[2]); d[3] *= gelu_grad_t<T>(v[3]);
    st4(dx + 4 * i, d);
}

static LnP pack(const tav_ln_args* a) {
    LnP p;
    p.x = a->x; p.gamma = a->gamma; p.beta = a->beta; p.y_f32 = a->y_f32; p.y_lp = a->y_lp; p.mean = a->mean; p.rstd = a->rstd;
    p.dy = a->dy; p.dx_add = a->dx_add; p.dx_f32 = a->dx_f32; p.dx_lp = a->dx_lp; p.partials = a->partials;
    p.rows = a->rows; p.W = (int)a->W; p.ld_x = a->ld_x; p.ld_y = a->ld_y; p.ld_dy = a->ld_dy; p.ld_dx = a->ld_dx; p.eps = a->eps; p.act = a->act;
    return p;
}

}  // namespace tav
using namespace tav;

extern "C" int tav_ln_bwd_partials(int64_t rows) { return ln_blocks(rows); }

extern "C" int tav_ln_fwd(const tav_ln_args* a, void* stream) {
    if (!a || !a->x || !a->gamma || !a->beta || (!a->y_f32 && !a->y_lp)) return TAV_ERR_NULL;
    if (a->rows <= 0 || a->W <= 0 || a->W > 1024 || a->W % 4) return TAV_ERR_SHAPE;
    if (a->ld_x % 4 || a->ld_y % 4) return TAV_ERR_ALIGN;
    const LnP p = pack(a);
    hipStream_t st = (hipStream_t)stream;
    const long want = (a->rows + 3) / 4;
    dim3 grid((unsigned)(want < 2048 ? want : 2048)), block(256);
    const bool lp_bf16 = a->y_lp && a->lp_dtype == TAV_BF16;
    if (a->x_dtype == TAV_F32) {
        if (lp_bf16) hipLaunchKernelGGL((ln_fwd_kernel<float, bf16>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((ln_fwd_kernel<float, float>), grid, block, 0, st, p);
    } else if (a->x_dtype == TAV_BF16) {
        if (lp_bf16) hipLaunchKernelGGL((ln_fwd_kernel<bf16, bf16>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((ln_fwd_kernel<bf16, float>), grid, block, 0, st, p);
    } else return TAV_ERR_DTYPE;
    return (int)hipGetLastError();
}

extern "C" int tav_ln_bwd(const tav_ln_args* a, void* stream) {
    if (!a || !a->x || !a->gamma || !a->dy || !a->mean || !a->rstd || (!a->dx_f32 && !a->dx_lp)) return TAV_ERR_NULL;
    if (a->act == 1 && !a->beta) return TAV_ERR_NULL;
    if ((a->dgamma || a->dbeta) && !a->partials) return TAV_ERR_NULL;
    if (a->rows <= 0 || a->W <= 0 || a->W > 1024 || a->W % 4) return TAV_ERR_SHAPE;
    if (a->ld_x % 4 || a->ld_dy % 4 || a->ld_dx % 4) return TAV_ERR_ALIGN;
    LnP p = pack(a);
    if (!a->dgamma && !a->dbeta && !a->defer_param_reduce) p.partials = nullptr;
    if (a->defer_param_reduce && !a->partials) return TAV_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const int nb = ln_blocks(a->rows);
    dim3 grid(nb), block(256);
    const bool lp_bf16 = a->dx_lp && a->lp_dtype == TAV_BF16;
#define TAV_LN_BWD(TX, TDY)                                                                              \
    do {                                                                                                 \
        if (lp_bf16) hipLaunchKernelGGL((ln_bwd_kernel<TX, TDY, bf16>), grid, block, 0, st, p);           \
        else hipLaunchKernelGGL((ln_bwd_kernel<TX, TDY, float>), grid, block, 0, st, p);                  \
    } while (0)
    if (a->x_dtype == TAV_F32 && a->dy_dtype == TAV_F32) TAV_LN_BWD(float, float);
    else if (a->x_dtype == TAV_F32 && a->dy_dtype == TAV_BF16) TAV_LN_BWD(float, bf16);
    else if (a->x_dtype == TAV_BF16 && a->dy_dtype == TAV_F32) TAV_LN_BWD(bf16, float);
    else if (a->x_dtype == TAV_BF16 && a->dy_dtype == TAV_BF16) TAV_LN_BWD(bf16, bf16);
    else return TAV_ERR_DTYPE;
#undef TAV_LN_BWD
    int e = (int)hipGetLastError();
    if (e) return e;
    if (p.partials && !a->defer_param_reduce) {
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(tav_cdiv(a->W, 16)), dim3(256), 0, st, a->partials, a->dgamma, a->dbeta, nb, (int)a->W,
                           a->accumulate_params);
        e = (int)hipGetLastError();
    }
    return e;
}

extern "C" int tav_ln_param_reduce_multi(const tav_ln_reduce_item* items, int32_t n, void* stream) {
    if (!items) return TAV_ERR_NULL;
    if (n <= 0 || n > TAV_LN_REDUCE_MAX) return TAV_ERR_SHAPE;
    LnReduceBatch b;
    int wmax = 0;
    for (int i = 0; i < n; ++i) {
        const tav_ln_reduce_item& it = items[i];
        if (!it.partials || (!it.dgamma && !it.dbeta)) return TAV_ERR_NULL;
        if (it.W <= 0 || it.W > 1024 || it.W % 4 || it.nblocks <= 0 || it.nblocks > LN_MAX_BLOCKS) return TAV_ERR_SHAPE;
        b.it[i] = it;
        wmax = it.W > wmax ? it.W : wmax;
    }
    for (int i = n; i < TAV_LN_REDUCE_MAX; ++i) b.it[i] = tav_ln_reduce_item{nullptr, nullptr, nullptr, 0, 0, 0, 0};
    hipLaunchKernelGGL(ln_param_reduce_multi_kernel, dim3(tav_cdiv(wmax, 16), n), dim3(256), 0, (hipStream_t)stream, b);
    return (int)hipGetLastError();
}

extern "C" int tav_gn_workspace_floats(int64_t B, int64_t C) { return (int)(B * GN_SPLIT * C * 2 + B * C * 2); }

extern "C" int tav_gn_gelu_fwd(const void* x, void* y, int32_t dtype, const float* gamma, const float* beta, float* stats, float* workspace,
                               int64_t B, int64_t T, int64_t C, float eps, void* stream) {
    if (!x || !y || !gamma || !beta || !stats || !workspace) return TAV_ERR_NULL;
    if (B <= 0 || T <= 0 || C <= 0 || C % 64) return TAV_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)tav_cdiv(C, GN_CB), GN_SPLIT, (unsigned)B);
    const dim3 agrid((unsigned)tav_cdiv(C, GN_CB), (unsigned)tav_cdiv(T, 4 * GN_AR), (unsigned)B);
    if (dtype == TAV_BF16) hipLaunchKernelGGL((gn_stats_kernel<bf16, false>), grid, dim3(256), 0, st, (const bf16*)x, (const bf16*)nullptr, gamma, beta, stats, workspace, (int)T, (int)C);
    else if (dtype == TAV_F32) hipLaunchKernelGGL((gn_stats_kernel<float, false>), grid, dim3(256), 0, st, (const float*)x, (const float*)nullptr, gamma, beta, stats, workspace, (int)T, (int)C);
    else return TAV_ERR_DTYPE;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(tav_cdiv(B * C, 256)), dim3(256), 0, st, workspace, stats, (int)T, (int)C, (int)B, eps, 1);
    if (dtype == TAV_BF16) hipLaunchKernelGGL((gn_apply_fwd_kernel<bf16>), agrid, dim3(256), 0, st, (const bf16*)x, (bf16*)y, gamma, beta, stats, (int)T, (int)C);
    else hipLaunchKernelGGL((gn_apply_fwd_kernel<float>), agrid, dim3(256), 0, st, (const float*)x, (float*)y, gamma, beta, stats, (int)T, (int)C);
    return (int)hipGetLastError();
}

extern "C" int tav_gn_gelu_bwd(const void* x, const void* dy, void* dx, int32_t dtype, const float* gamma, const float* beta, const float* stats,
                               float* workspace, float* dgamma, float* dbeta, int64_t B, int64_t T, int64_t C, int32_t accumulate, void* stream) {
    if (!x || !dy || !dx || !gamma || !beta || !stats || !workspace || !dgamma || !dbeta) return TAV_ERR_NULL;
    if (B <= 0 || T <= 0 || C <= 0 || C % 64) return TAV_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)tav_cdiv(C, GN_CB), GN_SPLIT, (unsigned)B);
    const dim3 agrid((unsigned)tav_cdiv(C, GN_CB), (unsigned)tav_cdiv(T, 4 * GN_AR), (unsigned)B);
    float* sums = workspace + B * GN_SPLIT * C * 2;
    if (dtype == TAV_BF16) hipLaunchKernelGGL((gn_stats_kernel<bf16, true>), grid, dim3(256), 0, st, (const bf16*)x, (const bf16*)dy, gamma, beta, stats, workspace, (int)T, (int)C);
    else if (dtype == TAV_F32) hipLaunchKernelGGL((gn_stats_kernel<float, true>), grid, dim3(256), 0, st, (const float*)x, (const float*)dy, gamma, beta, stats, workspace, (int)T, (int)C);
    else return TAV_ERR_DTYPE;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(tav_cdiv(B * C, 256)), dim3(256), 0, st, workspace, sums, (int)T, (int)C, (int)B, 0.f, 0);
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3(tav_cdiv(C, 256)), dim3(256), 0, st, sums, dgamma, dbeta, (int)B, (int)C, accumulate);
    if (dtype == TAV_BF16) hipLaunchKernelGGL((gn_apply_bwd_kernel<bf16>), agrid, dim3(256), 0, st, (const bf16*)x, (const bf16*)dy, (bf16*)dx, gamma, beta, stats, sums, (int)T, (int)C);
    else hipLaunchKernelGGL((gn_apply_bwd_kernel<float>), agrid, dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, gamma, beta, stats, sums, (int)T, (int)C);
    return (int)hipGetLastError();
}

extern "C" int tav_gelu_fwd(const void* x, void* y, int32_t dtype, int64_t n, void* stream) {
    if (!x || !y) return TAV_ERR_NULL;
    if (n <= 0 || n % 4) return TAV_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TAV_BF16) hipLaunchKernelGGL((gelu_fwd_kernel<bf16>), dim3(tav_cdiv(n / 4, 256)), dim3(256), 0, st, (const bf16*)x, (bf16*)y, n / 4);
    else if (dtype == TAV_F32) hipLaunchKernelGGL((gelu_fwd_kernel<float>), dim3(tav_cdiv(n / 4, 256)), dim3(256), 0, st, (const float*)x, (float*)y, n / 4);
    else return TAV_ERR_DTYPE;
    return (int)hipGetLastError();
}
extern "C" int tav_gelu_bwd(const void* x, const void* dy, void* dx, int32_t dtype, int64_t n, void* stream) {
    if (!x || !dy || !dx) return TAV_ERR_NULL;
    if (n <= 0 || n % 4) return TAV_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TAV_BF16) hipLaunchKernelGGL((gelu_bwd_kernel<bf16>), dim3(tav_cdiv(n / 4, 256)), dim3(256), 0, st, (const bf16*)x, (const bf16*)dy, (bf16*)dx, n / 4);
    else if (dtype == TAV_F32) hipLaunchKernelGGL((gelu_bwd_kernel<float>), dim3(tav_cdiv(n / 4, 256)), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, n / 4);
    else return TAV_ERR_DTYPE;
    return (int)hipGetLastError();
}
