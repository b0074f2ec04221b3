// HBM-bound support kernels of the TAV path: weight casts (+ transposed copies for dgrad), casts/adds, modality and
// text embeddings, video patch gather, pooling, classifier head, cross-entropy, dropout, grad-norm and AdamW.
// Every kernel is coalesced along the feature axis and vectorised 4-wide where the shape allows.
#include "common.h"
#include "tavhip_internal.h"

namespace tav {

// ------------------------------------------------------------------------------------------------ weight casts
template <typename TD>
__global__ void cast_weight_kernel(const float* __restrict__ src, TD* __restrict__ dst, long ld_n, TD* __restrict__ dst_t, long ld_t, int R, int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) { v = src[(long)r * C + c]; if (dst) ET<TD>::st(dst + (long)r * ld_n + c, v); }
        tile[k][tx] = v;
    }
    __syncthreads();
    if (dst_t) {
        for (int k = ty; k < 32; k += 8) {
            const int c = c0 + k, r = r0 + tx;
            if (r < R && c < C) ET<TD>::st(dst_t + (long)c * ld_t + r, tile[tx][k]);
        }
    }
}

// Multi-tensor form: blockIdx.y selects a descriptor (one launch refreshes every operand copy of a transformer layer).
struct CastDesc { const float* src; void* dst; void* dst_t; long ld_n, ld_t; int R, C, dtype; float scale_n; };   // scale_n: factor on the values written to dst (not dst_t)
// bf16 matrices with R, C multiples of 64 and both copies wanted (every GEMM weight): 64x64 tiles, 16-B loads, 8-B stores (128-B row
// segments) -- the 32x32 / 2-B-store form below ran at 2.1 TB/s
__device__ void cast_weight_tile64(const CastDesc& d, float (*tile)[65], int t, int tiles_c) {
    const int c0 = (t % tiles_c) * 64, r0 = (t / tiles_c) * 64;
    const int q = threadIdx.x & 15, h = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = h + 16 * i;
        const f32x4 v = ld4(d.src + (long)(r0 + row) * d.C + c0 + 4 * q);
        st4((bf16*)d.dst + (long)(r0 + row) * d.ld_n + c0 + 4 * q, v * d.scale_n);
        tile[row][4 * q] = v[0]; tile[row][4 * q + 1] = v[1]; tile[row][4 * q + 2] = v[2]; tile[row][4 * q + 3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = h + 16 * i;
        const f32x4 v = {tile[4 * q][c], tile[4 * q + 1][c], tile[4 * q + 2][c], tile[4 * q + 3][c]};
        st4((bf16*)d.dst_t + (long)(c0 + c) * d.ld_t + r0 + 4 * q, v);
    }
    __syncthreads();
}
__global__ void cast_weight_multi_kernel(const CastDesc* __restrict__ descs) {
    __shared__ float tile[64][65];
    const CastDesc d = descs[blockIdx.y];
    if (d.dtype == TAV_BF16 && d.dst && d.dst_t && (d.R & 63) == 0 && (d.C & 63) == 0 && (d.ld_n & 3) == 0 && (d.ld_t & 3) == 0) {
        const int tc = d.C >> 6, nt = tc * (d.R >> 6);
        for (int t = blockIdx.x; t < nt; t += gridDim.x) cast_weight_tile64(d, tile, t, tc);      // block-uniform loop
        return;
    }
    const int tiles_c = (d.C + 31) >> 5, tiles_r = (d.R + 31) >> 5;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {      // block-uniform loop
        const int c0 = (t % tiles_c) * 32, r0 = (t / tiles_c) * 32;
        for (int k = ty; k < 32; k += 8) {
            const int r = r0 + k, c = c0 + tx;
            float v = 0.f;
            if (r < d.R && c < d.C) {
                v = d.src[(long)r * d.C + c];
                if (d.dst) { if (d.dtype == TAV_BF16) ET<bf16>::st((bf16*)d.dst + (long)r * d.ld_n + c, v * d.scale_n); else ((float*)d.dst)[(long)r * d.ld_n + c] = v * d.scale_n; }
            }
            tile[k][tx] = v;
        }
        __syncthreads();
        if (d.dst_t) {
            for (int k = ty; k < 32; k += 8) {
                const int c = c0 + k, r = r0 + tx;
                if (r < d.R && c < d.C) {
                    if (d.dtype == TAV_BF16) ET<bf16>::st((bf16*)d.dst_t + (long)c * d.ld_t + r, tile[tx][k]); else ((float*)d.dst_t)[(long)c * d.ld_t + r] = tile[tx][k];
                }
            }
        }
        __syncthreads();
    }
}

// nn.Conv1d weight [co][ci][k] -> dst[co][k*CI + ci] (forward/wgrad operand) and dst_t[k*CI + ci][co] (dgrad operand)
// ... and (v5) dst_ph: the input-gradient operands BY OUTPUT PHASE.  dx[s*m + r] = sum over the taps j = r + q*s of dy[m - q] . W[:, :, j]: for every
// residue r of the input position, one NT GEMM whose A row is the Q_r = ceil((K - r) / s) CONSECUTIVE dy rows m - (Q_r - 1) .. m (overlapping
// rows, lda = CO) and whose B operand is  ph_r[ci][q' * CO + co] = W[co][ci][r + (Q_r - 1 - q') * s]  -- the blocks ph_0 | ph_1 | ... stand
// one after the other (CI * Q_r * CO elements each, CO * CI * K in all).  No [T_out, K * CI] column buffer and no col2im pass.
template <typename TD>
__global__ void cast_conv_weight_kernel(const float* __restrict__ src, TD* __restrict__ dst, TD* __restrict__ dst_t, TD* __restrict__ dst_ph, int CO, int CI,
                                        int K, int stride) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)CO * CI * K;
    if (idx >= n) return;
    const int co = (int)(idx / ((long)CI * K));
    const int rem = (int)(idx - (long)co * CI * K);
    const int kk = rem / CI, ci = rem - kk * CI;          // idx enumerates dst order [co][kk][ci]
    const float v = src[((long)co * CI + ci) * K + kk];
    if (dst) ET<TD>::st(dst + idx, v);
    if (dst_t) ET<TD>::st(dst_t + ((long)kk * CI + ci) * CO + co, v);
    if (dst_ph) {
        const int r = kk % stride, q = kk / stride;
        long off = 0;                                       // elements of the phases before r
        for (int rp = 0; rp < r; ++rp) off += (long)CI * CO * ((K - rp + stride - 1) / stride);
        const int Q = (K - r + stride - 1) / stride;
        ET<TD>::st(dst_ph + off + ((long)ci * Q + (Q - 1 - q)) * CO + co, v);
    }
}

// dst[z][c][r] = src[z][r][c]  (batched 2-D transpose through a padded LDS tile; both sides coalesced)
template <typename T>
__global__ void transpose2d_kernel(const T* __restrict__ src, T* __restrict__ dst, int R, int C) {
    __shared__ float tile[32][33];
    const long zoff = (long)blockIdx.z * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? ET<T>::ld(src + zoff + (long)r * C + c) : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < R && c < C) ET<T>::st(dst + zoff + (long)c * R + r, tile[tx][k]);
    }
}

template <typename TS, typename TD>
__global__ void cast2d_kernel(const TS* __restrict__ src, long ld_src, TD* __restrict__ dst, long ld_dst, long R, int C4) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * C4) return;
    const long r = idx / C4; const int c = (int)(idx - r * C4) * 4;
    st4(dst + r * ld_dst + c, ld4(src + r * ld_src + c));
}

template <typename TL>
__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, TL* __restrict__ y_lp, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = ld4(a + 4 * i) + ld4(b + 4 * i);
    if (y) st4(y + 4 * i, v);
    if (y_lp) st4(y_lp + 4 * i, v);
}

__global__ void fill_f32_kernel(float* p, float v, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------------------------------------ modality embedding add
__global__ void embed_add_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ ids, const float* __restrict__ table,
                                     float* __restrict__ out, long rows, int W4) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * W4) return;
    const long r = idx / W4; const int c = (int)(idx - r * W4) * 4;
    const long W = (long)W4 * 4;
    st4(out + r * W + c, ld4(x + r * W + c) + ld4(table + ids[r] * W + c));
}
// partial[block][t][c] = sum over this block's rows with ids == t
__global__ void embed_add_bwd_partial_kernel(const float* __restrict__ dy, const int64_t* __restrict__ ids, float* __restrict__ partial, long rows,
                                             int W, int ntable, int rows_per_block) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block; r1 = r1 < rows ? r1 : rows;
    float acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.f;
    for (long r = r0; r < r1; ++r) {
        const int t = (int)ids[r];
        const float v = dy[r * W + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += (k == t) ? v : 0.f;
    }
    for (int t = 0; t < ntable; ++t) partial[((long)blockIdx.y * ntable + t) * W + c] = acc[t];
}
__global__ void embed_add_bwd_final_kernel(const float* __restrict__ partial, float* dtable, int nparts, int ntable, int W, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ntable * W) return;
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += partial[(long)k * ntable * W + idx];
    dtable[idx] = accumulate ? dtable[idx] + s : s;
}

// ------------------------------------------------------------------------------------------------ text embeddings
__global__ void text_pos_ids_kernel(const int64_t* __restrict__ ids, int64_t* __restrict__ pos_ids, int S, int pad_id) {
    __shared__ int scan[1024];
    const int b = blockIdx.x, s = threadIdx.x;
    const bool in = s < S;
    const int id = in ? (int)ids[(long)b * S + s] : pad_id;
    const int nonpad = (pad_id >= 0) ? (id != pad_id ? 1 : 0) : 1;
    scan[s] = in ? nonpad : 0;
    __syncthreads();
    for (int off = 1; off < (int)blockDim.x; off <<= 1) {
        const int v = (s >= off) ? scan[s - off] : 0;
        __syncthreads();
        scan[s] += v;
        __syncthreads();
    }
    if (in) pos_ids[(long)b * S + s] = (pad_id >= 0) ? (long)(scan[s] * nonpad + pad_id) : (long)s;
}
__global__ void text_embed_sum_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_ids, const float* __restrict__ word,
                                      const float* __restrict__ pos, const float* __restrict__ type, float* __restrict__ pre, long rows, int W4) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * W4) return;
    const long r = idx / W4; const int c = (int)(idx - r * W4) * 4;
    const long W = (long)W4 * 4;
    st4(pre + r * W + c, ld4(word + ids[r] * W + c) + ld4(pos + pos_ids[r] * W + c) + ld4(type + c));
}
// dtable[idx[r]][:] += d[r][:] without atomics (bitwise reproducible): one workgroup per token row r.  Every workgroup lists, in
// ascending order, the rows that carry the same index (rows <= a few thousand: rows/1024 compares per thread); only the workgroup
// of the FIRST such row goes on and adds their sum to the table row.  Wave w sums matches w, w+16, ... (fixed order), the 16
// partial rows meet in LDS in wave order.  A padding token shared by a quarter of the batch costs rows/64 serial row reads.
constexpr int SCAT_T = 1024, SCAT_MAXW = 1024;
__global__ __launch_bounds__(SCAT_T) void scatter_add_rows_kernel(const float* __restrict__ d, const int64_t* __restrict__ idx, float* __restrict__ dtable,
                                                                  int rows, int W, long ntable) {
    extern __shared__ __attribute__((aligned(16))) char scat_smem[];
    // all LDS in the dynamic region (a static __shared__ object in front would move its base off 16-B alignment)
    int* wcount = reinterpret_cast<int*>(scat_smem);                   // [16]
    int* wbase = wcount + 16;                                          // [16]
    int& total = wcount[32];
    int& dup_before = wcount[33];
    int* match = reinterpret_cast<int*>(scat_smem + 256);              // [rows] worst case
    const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t id = idx[r];
    if (t == 0) dup_before = 0;
    __syncthreads();
    const int per = (rows + SCAT_T - 1) / SCAT_T, i0 = t * per;
    int i1 = i0 + per; i1 = i1 < rows ? i1 : rows;
    int cnt = 0;
    for (int i = i0; i < i1; ++i) if (idx[i] == id) { ++cnt; if (i < r) dup_before = 1; }      // benign race: every writer stores 1
    __syncthreads();
    if (dup_before || id < 0 || id >= ntable) return;                  // not the first row with this index (or an index outside the table)
    // ordered compaction of the matching rows: exclusive scan of cnt over the threads (wave scan + 16 wave totals)
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) wcount[wave] = incl;
    __syncthreads();
    if (t == 0) { int a = 0; for (int w = 0; w < 16; ++w) { wbase[w] = a; a += wcount[w]; } total = a; }
    __syncthreads();
    int pos = wbase[wave] + incl - cnt;
    for (int i = i0; i < i1; ++i) if (idx[i] == id) match[pos++] = i;
    __syncthreads();
    const int n = total;
    float* part = reinterpret_cast<float*>(scat_smem + 256 + (((size_t)rows * 4 + 15) & ~(size_t)15));   // [16][W]
    // lane owns float4 columns lane, lane+64, ... (W <= 1024: at most 4)
    f32x4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int W4 = W / 4;
    for (int m = wave; m < n; m += 16) {
        const float* src = d + (long)match[m] * W;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int c = lane + 64 * k; if (c < W4) acc[k] += ld4(src + 4 * c); }
    }
    const int nw = n < 16 ? n : 16;                                      // waves that hold a partial sum
    if (nw > 1) {
        if (wave < nw) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int c = lane + 64 * k; if (c < W4) st4(part + (long)wave * W + 4 * c, acc[k]); }
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* dst = dtable + id * W;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + 64 * k;
            if (c < W4) {
                f32x4 v = acc[k];
                for (int w = 1; w < nw; ++w) v += ld4(part + (long)w * W + 4 * c);
                st4(dst + 4 * c, ld4(dst + 4 * c) + v);
            }
        }
    }
}
// indices are clamped into the table: a caller's bad index must read a wrong row, never fault the device
__global__ void gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ idx, float* __restrict__ out, long rows, int W4, int ntable) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * W4) return;
    const long r = i / W4; const int c = (int)(i - r * W4) * 4;
    const long W = (long)W4 * 4;
    int t = idx[r]; t = t < 0 ? 0 : (t >= ntable ? ntable - 1 : t);
    st4(out + r * W + c, ld4(table + (long)t * W + c));
}

// ------------------------------------------------------------------------------------------------ video patches
// one workgroup per kept token; patch element order = Conv3d weight order (c, dt, dh, dw)
template <typename TD>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ video, const int32_t* __restrict__ keep_idx, TD* __restrict__ out,
                                                       int F, int Hh, int Ww, int nkeep) {
    const int j = blockIdx.x, b = blockIdx.y;
    const int gw = Ww / 16, gh = Hh / 16, ntok = (F / 2) * gh * gw;
    int tok = keep_idx[(long)b * nkeep + j];
    tok = tok < 0 ? 0 : (tok >= ntok ? ntok - 1 : tok);       // never index outside the clip (a short row of the mask repeats its last token)
    const int tt = tok / (gh * gw), rem = tok - tt * gh * gw, hh = rem / gw, ww = rem - hh * gw;
    TD* orow = out + ((long)b * nkeep + j) * 1536;
    for (int v = threadIdx.x; v < 384; v += 256) {           // 384 float4 = 3*2*16 rows of 16 pixels
        const int dw4 = v & 3, rowi = v >> 2;                // rowi = (c*2 + dt)*16 + dh
        const int dh = rowi & 15, cd = rowi >> 4, dt = cd & 1, c = cd >> 1;
        const float* srcp = video + ((((long)b * F + (2 * tt + dt)) * 3 + c) * Hh + (16 * hh + dh)) * Ww + 16 * ww + 4 * dw4;
        st4(orow + rowi * 16 + 4 * dw4, ld4(srcp));
    }
}
__global__ __launch_bounds__(256) void mask_to_index_kernel(const uint8_t* __restrict__ mask, int keep_value, int32_t* __restrict__ keep_idx,
                                                            int32_t* __restrict__ counts, int n, int nkeep) {
    __shared__ int scan[256];
    const int b = blockIdx.x, t = threadIdx.x;
    const int per = (n + 255) / 256, i0 = t * per;
    int i1 = i0 + per; i1 = i1 < n ? i1 : n;
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += ((mask[(long)b * n + i] != 0) == (keep_value != 0)) ? 1 : 0;
    scan[t] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = (t >= off) ? scan[t - off] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    int pos = scan[t] - cnt;
    for (int i = i0; i < i1; ++i)
        if (((mask[(long)b * n + i] != 0) == (keep_value != 0))) { if (pos < nkeep) keep_idx[(long)b * nkeep + pos] = i; ++pos; }
    if (t == 255 && counts) counts[b] = scan[255];
    // a row with fewer than nkeep kept tokens: the unwritten slots repeat the row's last kept index (0 if it has none), so every
    // consumer reads valid memory; the caller decides from counts[] whether that is an error
    const int found = scan[255];
    if (found < nkeep) {
        __shared__ int last_kept;
        if (t == 0) last_kept = 0;
        __syncthreads();
        if (cnt > 0 && scan[t] == found) {                       // the thread that owns the last kept element
            int last = 0;
            for (int i = i0; i < i1; ++i) if (((mask[(long)b * n + i] != 0) == (keep_value != 0))) last = i;
            last_kept = last;
        }
        __syncthreads();
        for (int k = found + t; k < nkeep; k += 256) keep_idx[(long)b * nkeep + k] = last_kept;
    }
}

// ------------------------------------------------------------------------------------------------ pooling / head / loss
// block = 16 float4 columns x 16 row groups: every thread sums rows rg, rg+16, ... of its 4 columns (16-B loads, S/16 iterations)
__global__ __launch_bounds__(256) void mean_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int S, int W) {
    __shared__ f32x4 red[16][16];
    const int b = blockIdx.y, cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = (blockIdx.x * 16 + cq) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < W) for (int s = rg; s < S; s += 16) a += ld4(x + ((long)b * S + s) * W + c);
    red[rg][cq] = a;
    __syncthreads();
    if (rg == 0 && c < W) {
#pragma unroll
        for (int k = 1; k < 16; ++k) a += red[k][cq];
        st4(y + (long)b * W + c, a * (1.f / (float)S));
    }
}
template <typename TL>
__global__ void mean_pool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, TL* __restrict__ dx_lp, long n4, int S, int W4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const long row = i / W4; const int c = (int)(i - row * W4) * 4;
    const long b = row / S;
    const f32x4 v = ld4(dy + b * W4 * 4 + c) * (1.f / (float)S);
    if (dx) st4(dx + 4 * i, v);
    if (dx_lp) st4(dx_lp + 4 * i, v);
}

__global__ void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ Wt, const float* __restrict__ bias, float* __restrict__ y,
                                int Bn, int K, int N) {
    const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= Bn * N) return;
    const int b = o / N, n = o - b * N;
    float a = 0.f;
    for (int k = lane; k < K; k += 64) a += x[(long)b * K + k] * Wt[(long)n * K + k];
    a = wave_sum(a);
    if (lane == 0) y[o] = a + (bias ? bias[n] : 0.f);
}
__global__ void head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ Wt, const float* __restrict__ dy, float* __restrict__ dx,
                                float* __restrict__ dW, float* __restrict__ db, int Bn, int K, int N, int accumulate) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) {
        if (dx)
            for (int b = 0; b < Bn; ++b) {
                float a = 0.f;
                for (int n = 0; n < N; ++n) a += dy[b * N + n] * Wt[(long)n * K + k];
                dx[(long)b * K + k] = a;
            }
        for (int n = 0; n < N; ++n) {
            float a = 0.f;
            for (int b = 0; b < Bn; ++b) a += dy[b * N + n] * x[(long)b * K + k];
            dW[(long)n * K + k] = accumulate ? dW[(long)n * K + k] + a : a;
        }
    }
    if (db && blockIdx.x == 0 && threadIdx.x < N) {
        float a = 0.f;
        for (int b = 0; b < Bn; ++b) a += dy[b * N + threadIdx.x];
        db[threadIdx.x] = accumulate ? db[threadIdx.x] + a : a;
    }
}
__global__ void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = tanhf(x[i]);
}
__global__ void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = dy[i] * (1.f - y[i] * y[i]);
}

// single workgroup: B samples, C classes (C <= 32)
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                            const float* __restrict__ cw, float* __restrict__ loss, float* __restrict__ dlogits,
                                                            int Bn, int C, float grad_scale) {
    __shared__ float s_num[256], s_den[256];
    float num = 0.f, den = 0.f;
    for (int i = threadIdx.x; i < Bn; i += 256) {
        const float* z = logits + (long)i * C;
        float m = z[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c] - m);
        const int t = (int)target[i];
        const float w = cw ? cw[t] : 1.f;
        num += w * (m + logf(se) - z[t]);
        den += w;
    }
    s_num[threadIdx.x] = num; s_den[threadIdx.x] = den;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { s_num[threadIdx.x] += s_num[threadIdx.x + off]; s_den[threadIdx.x] += s_den[threadIdx.x + off]; }
        __syncthreads();
    }
    const float wsum = s_den[0];
    if (threadIdx.x == 0 && loss) loss[0] = s_num[0] / wsum;
    if (dlogits)
        for (int i = threadIdx.x; i < Bn; i += 256) {
            const float* z = logits + (long)i * C;
            float m = z[0];
            for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += expf(z[c] - m);
            const int t = (int)target[i];
            const float w = (cw ? cw[t] : 1.f) * grad_scale / wsum;
            for (int c = 0; c < C; ++c) dlogits[(long)i * C + c] = w * (expf(z[c] - m) / se - (c == t ? 1.f : 0.f));
        }
}

TAV_DEV uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask, long n, float p, uint64_t seed,
                                   uint64_t offset) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = mix64(seed ^ mix64(offset + (uint64_t)i));
    const float u = (float)(r >> 40) * (1.f / 16777216.f);
    const bool keep = u >= p;
    mask[i] = keep ? 1 : 0;
    y[i] = keep ? x[i] / (1.f - p) : 0.f;
}
__global__ void dropout_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask, float* __restrict__ dx, long n, float p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = mask[i] ? dy[i] / (1.f - p) : 0.f;
}

// ------------------------------------------------------------------------------------------------ optimiser
constexpr int SUMSQ_BLOCKS = 96;
__global__ __launch_bounds__(256) void sumsq_multi_kernel(const float* const* __restrict__ ptrs, const int64_t* __restrict__ sizes, float* __restrict__ partials) {
    __shared__ float red[256];
    const int t = blockIdx.y;
    const float* p = ptrs[t];
    const long n = sizes[t];
    float a = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)SUMSQ_BLOCKS * 256) { const float v = p[i]; a += v * v; }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) partials[(long)t * SUMSQ_BLOCKS + blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sum_final_kernel(const float* __restrict__ partials, long n, float* out) {
    __shared__ float red[256];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;            // ~19 k partials in one workgroup: four loads in flight per thread
    long i = threadIdx.x;
    for (; i + 768 < n; i += 1024) { a0 += partials[i]; a1 += partials[i + 256]; a2 += partials[i + 512]; a3 += partials[i + 768]; }
    for (; i < n; i += 256) a0 += partials[i];
    const float a = (a0 + a1) + (a2 + a3);
    red[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float* coef, float* norm_out) {
    const float norm = sqrtf(sumsq[0]);
    if (norm_out) norm_out[0] = norm;
    float c = max_norm / (norm + 1e-6f);            // torch.nn.utils.clip_grad_norm_
    coef[0] = c < 1.f ? c : 1.f;
}
// step counter lives on the device so that a captured hipGraph replays with the right bias correction
__global__ void adam_tick_kernel(int* step, float b1, float b2, float* bc) {
    const int s = step[0] + 1;
    step[0] = s;
    bc[0] = 1.f - powf(b1, (float)s);
    bc[1] = 1.f - powf(b2, (float)s);
}
__global__ __launch_bounds__(256) void adamw_multi_kernel(float* const* __restrict__ params, const float* const* __restrict__ grads,
                                                          float* const* __restrict__ m1, float* const* __restrict__ m2, const int64_t* __restrict__ sizes,
                                                          const float* __restrict__ clip_coef, const float* __restrict__ lr_dev, float b1, float b2, float eps,
                                                          float wd, const float* __restrict__ bc) {
    const float lr = lr_dev[0], bc1 = bc[0], bc2 = bc[1];
    const int t = blockIdx.y;
    float* p = params[t]; const float* g = grads[t]; float* ea = m1[t]; float* es = m2[t];
    const long n = sizes[t];
    const float cc = clip_coef ? clip_coef[0] : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gr = g[i] * cc;
        float w = p[i] * (1.f - lr * wd);
        const float a = ea[i] * b1 + (1.f - b1) * gr;
        const float s = es[i] * b2 + (1.f - b2) * gr * gr;
        ea[i] = a; es[i] = s;
        const float denom = sqrtf(s) / sqrtf(bc2) + eps;
        w -= (lr / bc1) * a / denom;
        p[i] = w;
    }
}

// ---- chunked forms: block c handles chunk c of the concatenation of all tensors (chunk_prefix[t] = first chunk of tensor t, built once
// on the host since the sizes are static).  Every block has OPT_CHUNK elements of work (except a tensor's last chunk): no empty blocks
// for the ~400 small tensors, enough parallelism for the large ones, float4 accesses.
constexpr int OPT_CHUNK = 16384;
TAV_DEV int chunk_owner(const int32_t* __restrict__ prefix, int ntensors, int c) {
    int lo = 0, hi = ntensors - 1;
    while (lo < hi) {                               // last t with prefix[t] <= c
        const int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= c) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(256) void sumsq_chunk_kernel(const float* const* __restrict__ ptrs, const int64_t* __restrict__ sizes,
                                                          const int32_t* __restrict__ prefix, int ntensors, float* __restrict__ partials) {
    __shared__ float red[4];
    const int c = blockIdx.x, t = chunk_owner(prefix, ntensors, c);
    const long off = (long)(c - prefix[t]) * OPT_CHUNK;
    const float* p = ptrs[t] + off;
    long n = sizes[t] - off; n = n < OPT_CHUNK ? n : OPT_CHUNK;
    float a = 0.f;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const long n4 = n >> 2;
        for (long i = threadIdx.x; i < n4; i += 256) { const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i); a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
        for (long i = 4 * n4 + threadIdx.x; i < n; i += 256) a += p[i] * p[i];
    } else {
        for (long i = threadIdx.x; i < n; i += 256) a += p[i] * p[i];
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) partials[c] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void adamw_chunk_kernel(float* const* __restrict__ params, const float* const* __restrict__ grads,
                                                          float* const* __restrict__ m1, float* const* __restrict__ m2, const int64_t* __restrict__ sizes,
                                                          const int32_t* __restrict__ prefix, int ntensors, const float* __restrict__ clip_coef,
                                                          const float* __restrict__ lr_dev, float b1, float b2, float eps, float wd, const float* __restrict__ bc) {
    const float lr = lr_dev[0], bc1 = bc[0], bc2 = bc[1];
    const int c = blockIdx.x, t = chunk_owner(prefix, ntensors, c);
    const long off = (long)(c - prefix[t]) * OPT_CHUNK;
    float* p = params[t] + off; const float* g = grads[t] + off; float* ea = m1[t] + off; float* es = m2[t] + off;
    long n = sizes[t] - off; n = n < OPT_CHUNK ? n : OPT_CHUNK;
    const float cc = clip_coef ? clip_coef[0] : 1.f;
    const float decay = 1.f - lr * wd, step = lr / bc1, rs2 = 1.f / sqrtf(bc2), omb1 = 1.f - b1, omb2 = 1.f - b2;
    // One arithmetic for every element, whichever of the loops below it falls into: contraction is OFF and the three multiply-adds are written
    // out, so the 16-byte and the scalar form agree bit for bit.  (With the compiler free to contract, the two forms differed in the last bit of
    // some weights -- harmless alone, but an optimizer sharded over ranks cuts the tensors elsewhere than the replicated one, lands elements in the
    // other form, and Adam's m / sqrt(v) turns last-bit differences of near-zero gradients into lr-sized ones within a few steps.)
    auto upd = [&](float& w, float gr, float& a, float& s2) {
#pragma clang fp contract(off)
        gr = gr * cc;
        w = w * decay;
        a = __builtin_fmaf(a, b1, omb1 * gr);
        s2 = __builtin_fmaf(s2, b2, (omb2 * gr) * gr);
        const float den = __builtin_fmaf(sqrtf(s2), rs2, eps);
        const float q = (step * a) / den;
        w = w - q;
    };
    const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(ea) | reinterpret_cast<uintptr_t>(es)) & 15) == 0;
    long done = 0;
    if (al) {
        const long n4 = n >> 2;
        constexpr int U = 4;                        // 16 independent 16-B loads in flight per thread before the first use
        long i = threadIdx.x;
        for (; i + (U - 1) * 256 < n4; i += U * 256) {
            f32x4 w[U], a[U], s2[U], gr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                // every byte is touched exactly once per step: non-temporal both ways
                const long j = 4 * (i + u * 256);
                w[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p + j)); gr[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + j));
                a[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(ea + j)); s2[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(es + j));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float wk = w[u][k], ak = a[u][k], sk = s2[u][k];
                    upd(wk, gr[u][k], ak, sk);
                    w[u][k] = wk; a[u][k] = ak; s2[u][k] = sk;
                }
                const long j = 4 * (i + u * 256);
                *reinterpret_cast<f32x4*>(p + j) = w[u];         // the updated weights are read again right away (operand casts): default policy
                __builtin_nontemporal_store(a[u], reinterpret_cast<f32x4*>(ea + j)); __builtin_nontemporal_store(s2[u], reinterpret_cast<f32x4*>(es + j));
            }
        }
        for (; i < n4; i += 256) {
            f32x4 w = *reinterpret_cast<f32x4*>(p + 4 * i), a = *reinterpret_cast<f32x4*>(ea + 4 * i), s2 = *reinterpret_cast<f32x4*>(es + 4 * i);
            const f32x4 gr = *reinterpret_cast<const f32x4*>(g + 4 * i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float wk = w[k], ak = a[k], sk = s2[k];
                upd(wk, gr[k], ak, sk);
                w[k] = wk; a[k] = ak; s2[k] = sk;
            }
            *reinterpret_cast<f32x4*>(p + 4 * i) = w; *reinterpret_cast<f32x4*>(ea + 4 * i) = a; *reinterpret_cast<f32x4*>(es + 4 * i) = s2;
        }
        done = 4 * n4;
    }
    for (long i = done + threadIdx.x; i < n; i += 256) upd(p[i], g[i], ea[i], es[i]);
}

}  // namespace tav
using namespace tav;

#define ST ((hipStream_t)stream)
#define G1(n) dim3(tav_cdiv((n), 256)), dim3(256), 0, ST

extern "C" int tav_version(void) { return TAV_ABI_VERSION; }
extern "C" const char* tav_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case TAV_ERR_NULL: return "required pointer is NULL";
        case TAV_ERR_SHAPE: return "unsupported shape";
        case TAV_ERR_DTYPE: return "unsupported dtype";
        case TAV_ERR_ALIGN: return "stride/offset not 16-byte aligned";
        case TAV_ERR_NO_RCCL: return "RCCL not available (librccl could not be loaded)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown";
    }
}

extern "C" int tav_cast_weight(const float* src, int64_t R, int64_t C, void* dst, int64_t ld_dst, void* dst_t, int64_t ld_dst_t, int32_t dt, void* stream) {
    if (!src || (!dst && !dst_t)) return TAV_ERR_NULL;
    if (R <= 0 || C <= 0) return TAV_ERR_SHAPE;
    const long ld_n = ld_dst > 0 ? ld_dst : C, ld_t = ld_dst_t > 0 ? ld_dst_t : R;
    dim3 grid(tav_cdiv(C, 32), tav_cdiv(R, 32));
    if (dt == TAV_BF16) hipLaunchKernelGGL((cast_weight_kernel<bf16>), grid, dim3(256), 0, ST, src, (bf16*)dst, ld_n, (bf16*)dst_t, ld_t, (int)R, (int)C);
    else if (dt == TAV_F32) hipLaunchKernelGGL((cast_weight_kernel<float>), grid, dim3(256), 0, ST, src, (float*)dst, ld_n, (float*)dst_t, ld_t, (int)R, (int)C);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
extern "C" int tav_cast_weights_multi(const void* descs_dev, int32_t n, int32_t blocks_per_tensor, void* stream) {
    if (!descs_dev) return TAV_ERR_NULL;
    if (n <= 0 || n > 65535 || blocks_per_tensor <= 0) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(cast_weight_multi_kernel, dim3(blocks_per_tensor, n), dim3(256), 0, ST, (const CastDesc*)descs_dev);
    return tav_last_error();
}
extern "C" int tav_cast_conv_weight(const float* src, int64_t co, int64_t ci, int64_t k, void* dst, void* dst_t, void* dst_phase, int64_t conv_stride,
                                    int32_t dt, void* stream) {
    if (!src || (!dst && !dst_t && !dst_phase)) return TAV_ERR_NULL;
    if (co <= 0 || ci <= 0 || k <= 0) return TAV_ERR_SHAPE;
    if (dst_phase && (conv_stride <= 0 || conv_stride > k)) return TAV_ERR_SHAPE;     // (stride > k would leave phases without a tap)
    const long n = co * ci * k;
    const int cs = dst_phase ? (int)conv_stride : 1;
    if (dt == TAV_BF16) hipLaunchKernelGGL((cast_conv_weight_kernel<bf16>), G1(n), src, (bf16*)dst, (bf16*)dst_t, (bf16*)dst_phase, (int)co, (int)ci, (int)k, cs);
    else if (dt == TAV_F32) hipLaunchKernelGGL((cast_conv_weight_kernel<float>), G1(n), src, (float*)dst, (float*)dst_t, (float*)dst_phase, (int)co, (int)ci, (int)k, cs);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
// v5: zero rows [0, pad) and [pad + T, T + 2 pad) of every [T + 2 pad][C] batch entry (the zero frame around a padded gradient buffer)
template <typename T> __global__ void zero_pad_rows_kernel(T* __restrict__ buf, long n4, int Tn, int C, int pad) {
    const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= n4) return;
    const int c4 = C / 4;
    const long row = i4 / c4;                                // (b, pad row 0 .. 2 pad - 1)
    const int col = (int)(i4 - row * c4) * 4;
    const long b = row / (2 * pad);
    const int pr = (int)(row - b * 2 * pad);
    const long trow = b * ((long)Tn + 2 * pad) + (pr < pad ? pr : Tn + pr);
    st4(buf + trow * C + col, f32x4{0.f, 0.f, 0.f, 0.f});
}
extern "C" int tav_zero_pad_rows(void* buf, int32_t dt, int64_t B, int64_t T, int64_t C, int64_t pad, void* stream) {
    if (!buf) return TAV_ERR_NULL;
    if (B <= 0 || T <= 0 || C <= 0 || C % 4 || pad <= 0) return TAV_ERR_SHAPE;
    const long n4 = B * 2 * pad * (C / 4);
    if (dt == TAV_BF16) hipLaunchKernelGGL((zero_pad_rows_kernel<bf16>), G1(n4), (bf16*)buf, n4, (int)T, (int)C, (int)pad);
    else if (dt == TAV_F32) hipLaunchKernelGGL((zero_pad_rows_kernel<float>), G1(n4), (float*)buf, n4, (int)T, (int)C, (int)pad);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
extern "C" int tav_cast2d(const void* src, int32_t sdt, int64_t ld_src, void* dst, int32_t ddt, int64_t ld_dst, int64_t R, int64_t C, void* stream) {
    if (!src || !dst) return TAV_ERR_NULL;
    if (R <= 0 || C <= 0 || C % 4) return TAV_ERR_SHAPE;
    if (ld_src % 4 || ld_dst % 4) return TAV_ERR_ALIGN;
    const long n = R * (C / 4);
    if (sdt == TAV_F32 && ddt == TAV_BF16) hipLaunchKernelGGL((cast2d_kernel<float, bf16>), G1(n), (const float*)src, (long)ld_src, (bf16*)dst, (long)ld_dst, (long)R, (int)(C / 4));
    else if (sdt == TAV_BF16 && ddt == TAV_F32) hipLaunchKernelGGL((cast2d_kernel<bf16, float>), G1(n), (const bf16*)src, (long)ld_src, (float*)dst, (long)ld_dst, (long)R, (int)(C / 4));
    else if (sdt == TAV_F32 && ddt == TAV_F32) hipLaunchKernelGGL((cast2d_kernel<float, float>), G1(n), (const float*)src, (long)ld_src, (float*)dst, (long)ld_dst, (long)R, (int)(C / 4));
    else if (sdt == TAV_BF16 && ddt == TAV_BF16) hipLaunchKernelGGL((cast2d_kernel<bf16, bf16>), G1(n), (const bf16*)src, (long)ld_src, (bf16*)dst, (long)ld_dst, (long)R, (int)(C / 4));
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
extern "C" int tav_transpose2d(const void* src, void* dst, int32_t dt, int64_t R, int64_t C, int64_t nbatch, void* stream) {
    if (!src || !dst) return TAV_ERR_NULL;
    if (R <= 0 || C <= 0 || nbatch <= 0 || nbatch > 65535) return TAV_ERR_SHAPE;
    dim3 grid(tav_cdiv(C, 32), tav_cdiv(R, 32), (unsigned)nbatch);
    if (dt == TAV_BF16) hipLaunchKernelGGL((transpose2d_kernel<bf16>), grid, dim3(256), 0, ST, (const bf16*)src, (bf16*)dst, (int)R, (int)C);
    else if (dt == TAV_F32) hipLaunchKernelGGL((transpose2d_kernel<float>), grid, dim3(256), 0, ST, (const float*)src, (float*)dst, (int)R, (int)C);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
extern "C" int tav_add_f32(const float* a, const float* b, float* y, void* y_lp, int32_t lp, int64_t n, void* stream) {
    if (!a || !b || (!y && !y_lp)) return TAV_ERR_NULL;
    if (n <= 0 || n % 4) return TAV_ERR_SHAPE;
    if (y_lp && lp == TAV_BF16) hipLaunchKernelGGL((add_f32_kernel<bf16>), G1(n / 4), a, b, y, (bf16*)y_lp, n / 4);
    else hipLaunchKernelGGL((add_f32_kernel<float>), G1(n / 4), a, b, y, (float*)y_lp, n / 4);
    return tav_last_error();
}
extern "C" int tav_fill_f32(float* p, float v, int64_t n, void* stream) {
    if (!p) return TAV_ERR_NULL;
    if (n <= 0) return TAV_ERR_SHAPE;
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, ST, p, v, (long)n);
    return tav_last_error();
}
extern "C" int tav_embed_add_fwd(const float* x, const int64_t* ids, const float* table, float* out, int64_t rows, int64_t W, int64_t ntable, void* stream) {
    if (!x || !ids || !table || !out) return TAV_ERR_NULL;
    if (rows <= 0 || W <= 0 || W % 4 || ntable <= 0) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(embed_add_fwd_kernel, G1(rows * (W / 4)), x, ids, table, out, (long)rows, (int)(W / 4));
    return tav_last_error();
}
extern "C" int tav_embed_add_bwd_parts(int64_t rows) { long p = (rows + 255) / 256; return (int)(p > 256 ? 256 : p); }
extern "C" int tav_embed_add_bwd(const float* dy, const int64_t* ids, float* dtable, float* partials, int64_t rows, int64_t W, int64_t ntable,
                                 int32_t accumulate, void* stream) {
    if (!dy || !ids || !dtable || !partials) return TAV_ERR_NULL;
    if (rows <= 0 || W <= 0 || ntable <= 0 || ntable > 8) return TAV_ERR_SHAPE;
    const int nparts = tav_embed_add_bwd_parts(rows);
    const int rpb = (int)((rows + nparts - 1) / nparts);
    hipLaunchKernelGGL(embed_add_bwd_partial_kernel, dim3(tav_cdiv(W, 256), nparts), dim3(256), 0, ST, dy, ids, partials, (long)rows, (int)W, (int)ntable, rpb);
    hipLaunchKernelGGL(embed_add_bwd_final_kernel, G1(ntable * W), partials, dtable, nparts, (int)ntable, (int)W, accumulate);
    return tav_last_error();
}
extern "C" int tav_ln_fwd(const tav_ln_args* a, void* stream);
extern "C" int tav_text_embed_fwd(const tav_text_embed_args* a, void* stream) {
    if (!a || !a->ids || !a->word || !a->pos || !a->type || !a->gamma || !a->beta || !a->pre || !a->pos_ids) return TAV_ERR_NULL;
    if (a->B <= 0 || a->S <= 0 || a->S > 1024 || a->W <= 0 || a->W % 4) return TAV_ERR_SHAPE;
    int threads = 64; while (threads < a->S) threads <<= 1;
    hipLaunchKernelGGL(text_pos_ids_kernel, dim3((unsigned)a->B), dim3(threads), 0, ST, a->ids, a->pos_ids, (int)a->S, a->pad_id);
    const long rows = a->B * a->S;
    hipLaunchKernelGGL(text_embed_sum_kernel, G1(rows * (a->W / 4)), a->ids, a->pos_ids, a->word, a->pos, a->type, a->pre, rows, (int)(a->W / 4));
    int e = tav_last_error();
    if (e) return e;
    tav_ln_args ln = {};
    ln.x = a->pre; ln.x_dtype = TAV_F32; ln.gamma = a->gamma; ln.beta = a->beta; ln.y_f32 = a->y_f32; ln.y_lp = a->y_lp; ln.lp_dtype = a->lp_dtype;
    ln.mean = a->mean; ln.rstd = a->rstd; ln.rows = rows; ln.W = a->W; ln.ld_x = a->W; ln.ld_y = a->W; ln.eps = a->eps;
    return tav_ln_fwd(&ln, stream);
}
extern "C" int tav_scatter_add_rows(const float* d, const int64_t* idx, float* dtable, int64_t rows, int64_t W, int64_t ntable, void* stream) {
    if (!d || !idx || !dtable) return TAV_ERR_NULL;
    if (rows <= 0 || W <= 0 || ntable <= 0 || W % 4 || W > SCAT_MAXW) return TAV_ERR_SHAPE;
    const size_t lds = 256 + (((size_t)rows * 4 + 15) & ~(size_t)15) + (size_t)16 * W * 4;  // header + match list + 16 partial rows
    if (lds > 150 * 1024) return TAV_ERR_SHAPE;                                           // rows <= ~22 k tokens per call at W = 768
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)rows), dim3(SCAT_T), lds, ST, d, idx, dtable, (int)rows, (int)W, (long)ntable);
    return tav_last_error();
}
extern "C" int tav_gather_rows(const float* table, const int32_t* idx, float* out, int64_t rows, int64_t W, int64_t ntable, void* stream) {
    if (!table || !idx || !out) return TAV_ERR_NULL;
    if (rows <= 0 || W <= 0 || W % 4 || ntable <= 0) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(gather_rows_kernel, G1(rows * (W / 4)), table, idx, out, (long)rows, (int)(W / 4), (int)ntable);
    return tav_last_error();
}
extern "C" int tav_patchify(const float* video, const int32_t* keep_idx, void* patches, int32_t dt, int64_t B, int64_t F, int64_t H, int64_t W, int64_t nkeep,
                            void* stream) {
    if (!video || !keep_idx || !patches) return TAV_ERR_NULL;
    if (B <= 0 || F <= 0 || F % 2 || H % 16 || W % 16 || nkeep <= 0) return TAV_ERR_SHAPE;
    dim3 grid((unsigned)nkeep, (unsigned)B);
    if (dt == TAV_BF16) hipLaunchKernelGGL((patchify_kernel<bf16>), grid, dim3(256), 0, ST, video, keep_idx, (bf16*)patches, (int)F, (int)H, (int)W, (int)nkeep);
    else if (dt == TAV_F32) hipLaunchKernelGGL((patchify_kernel<float>), grid, dim3(256), 0, ST, video, keep_idx, (float*)patches, (int)F, (int)H, (int)W, (int)nkeep);
    else return TAV_ERR_DTYPE;
    return tav_last_error();
}
extern "C" int tav_mask_to_index(const uint8_t* mask, int32_t keep_value, int32_t* keep_idx, int32_t* counts, int64_t B, int64_t n, int64_t nkeep, void* stream) {
    if (!mask || !keep_idx) return TAV_ERR_NULL;
    if (B <= 0 || n <= 0 || nkeep <= 0 || nkeep > n) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(mask_to_index_kernel, dim3((unsigned)B), dim3(256), 0, ST, mask, keep_value, keep_idx, counts, (int)n, (int)nkeep);
    return tav_last_error();
}
extern "C" int tav_mean_pool_fwd(const float* x, float* y, int64_t B, int64_t S, int64_t W, void* stream) {
    if (!x || !y) return TAV_ERR_NULL;
    if (B <= 0 || S <= 0 || W <= 0) return TAV_ERR_SHAPE;
    if (W % 4) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(mean_pool_fwd_kernel, dim3(tav_cdiv(W, 64), (unsigned)B), dim3(256), 0, ST, x, y, (int)S, (int)W);
    return tav_last_error();
}
extern "C" int tav_mean_pool_bwd(const float* dy, float* dx, void* dx_lp, int32_t lp, int64_t B, int64_t S, int64_t W, void* stream) {
    if (!dy || (!dx && !dx_lp)) return TAV_ERR_NULL;
    if (B <= 0 || S <= 0 || W <= 0 || W % 4) return TAV_ERR_SHAPE;
    const long n4 = B * S * W / 4;
    if (dx_lp && lp == TAV_BF16) hipLaunchKernelGGL((mean_pool_bwd_kernel<bf16>), G1(n4), dy, dx, (bf16*)dx_lp, n4, (int)S, (int)(W / 4));
    else hipLaunchKernelGGL((mean_pool_bwd_kernel<float>), G1(n4), dy, dx, (float*)dx_lp, n4, (int)S, (int)(W / 4));
    return tav_last_error();
}
extern "C" int tav_head_fwd(const float* x, const float* W, const float* b, float* y, int64_t B, int64_t K, int64_t N, void* stream) {
    if (!x || !W || !y) return TAV_ERR_NULL;
    if (B <= 0 || K <= 0 || N <= 0) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(tav_cdiv(B * N, 4)), dim3(256), 0, ST, x, W, b, y, (int)B, (int)K, (int)N);
    return tav_last_error();
}
extern "C" int tav_head_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db, int64_t B, int64_t K, int64_t N,
                            int32_t accumulate, void* stream) {
    if (!x || !W || !dy || !dW) return TAV_ERR_NULL;
    if (B <= 0 || K <= 0 || N <= 0 || N > 256) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(head_bwd_kernel, G1(K), x, W, dy, dx, dW, db, (int)B, (int)K, (int)N, accumulate);
    return tav_last_error();
}
extern "C" int tav_tanh_fwd(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y) return TAV_ERR_NULL;
    if (n <= 0) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(tanh_fwd_kernel, G1(n), x, y, (long)n);
    return tav_last_error();
}
extern "C" int tav_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) {
    if (!y || !dy || !dx) return TAV_ERR_NULL;
    if (n <= 0) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(tanh_bwd_kernel, G1(n), y, dy, dx, (long)n);
    return tav_last_error();
}
extern "C" int tav_cross_entropy(const float* logits, const int64_t* target, const float* cw, float* loss, float* dlogits, int64_t B, int64_t C,
                                 float grad_scale, void* stream) {
    if (!logits || !target || (!loss && !dlogits)) return TAV_ERR_NULL;
    if (B <= 0 || C <= 0 || C > 64) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, ST, logits, target, cw, loss, dlogits, (int)B, (int)C, grad_scale);
    return tav_last_error();
}
extern "C" int tav_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
    if (!x || !y || !mask) return TAV_ERR_NULL;
    if (n <= 0 || p < 0.f || p >= 1.f) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(dropout_fwd_kernel, G1(n), x, y, mask, (long)n, p, seed, offset);
    return tav_last_error();
}
extern "C" int tav_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, void* stream) {
    if (!dy || !mask || !dx) return TAV_ERR_NULL;
    if (n <= 0 || p < 0.f || p >= 1.f) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(dropout_bwd_kernel, G1(n), dy, mask, dx, (long)n, p);
    return tav_last_error();
}
extern "C" int tav_sumsq_partials(int32_t ntensors) { return ntensors * SUMSQ_BLOCKS; }
extern "C" int tav_sumsq_multi(const float* const* ptrs, const int64_t* sizes, int32_t ntensors, float* partials, float* out_sumsq, void* stream) {
    if (!ptrs || !sizes || !partials || !out_sumsq) return TAV_ERR_NULL;
    if (ntensors <= 0 || ntensors > 65535) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(SUMSQ_BLOCKS, ntensors), dim3(256), 0, ST, ptrs, sizes, partials);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, ST, partials, (long)ntensors * SUMSQ_BLOCKS, out_sumsq);
    return tav_last_error();
}
extern "C" int tav_optim_chunk_elems(void) { return OPT_CHUNK; }
extern "C" int tav_sumsq_chunked(const float* const* ptrs, const int64_t* sizes, const int32_t* chunk_prefix, int32_t ntensors, int32_t nchunks,
                                 float* partials, float* out_sumsq, void* stream) {
    if (!ptrs || !sizes || !chunk_prefix || !partials || !out_sumsq) return TAV_ERR_NULL;
    if (ntensors <= 0 || nchunks <= 0) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(sumsq_chunk_kernel, dim3(nchunks), dim3(256), 0, ST, ptrs, sizes, chunk_prefix, ntensors, partials);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, ST, partials, (long)nchunks, out_sumsq);
    return tav_last_error();
}
// second stage of tav_sumsq_chunked on its own: a sharded optimizer computes the partials of the chunks it owns, the ranks exchange them, and every
// rank adds the complete array here -- same kernel, same order, hence the same norm bit for bit as the replicated optimizer's
extern "C" int tav_sum_partials(const float* partials, int64_t n, float* out, void* stream) {
    if (!partials || !out) return TAV_ERR_NULL;
    if (n <= 0) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, ST, partials, (long)n, out);
    return tav_last_error();
}
extern "C" int tav_adamw_chunked(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes,
                                 const int32_t* chunk_prefix, int32_t ntensors, int32_t nchunks, const float* clip_coef, const float* lr, float beta1,
                                 float beta2, float eps, float weight_decay, int32_t* step, float* bias_corr, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !sizes || !chunk_prefix || !lr || !step || !bias_corr) return TAV_ERR_NULL;
    if (ntensors <= 0 || nchunks <= 0) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, ST, step, beta1, beta2, bias_corr);
    hipLaunchKernelGGL(adamw_chunk_kernel, dim3(nchunks), dim3(256), 0, ST, params, grads, exp_avg, exp_avg_sq, sizes, chunk_prefix, ntensors, clip_coef, lr,
                       beta1, beta2, eps, weight_decay, (const float*)bias_corr);
    return tav_last_error();
}
extern "C" int tav_clip_coef(const float* sumsq, float max_norm, float* coef_out, float* norm_out, void* stream) {
    if (!sumsq || !coef_out) return TAV_ERR_NULL;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, ST, sumsq, max_norm, coef_out, norm_out);
    return tav_last_error();
}
extern "C" int tav_adamw_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes,
                               int32_t ntensors, const float* clip_coef, const float* lr, float beta1, float beta2, float eps, float weight_decay,
                               int32_t* step, float* bias_corr, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !sizes || !lr || !step || !bias_corr) return TAV_ERR_NULL;
    if (ntensors <= 0 || ntensors > 65535) return TAV_ERR_SHAPE;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, ST, step, beta1, beta2, bias_corr);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(192, ntensors), dim3(256), 0, ST, params, grads, exp_avg, exp_avg_sq, sizes, clip_coef, lr, beta1, beta2, eps,
                       weight_decay, (const float*)bias_corr);
    return tav_last_error();
}
