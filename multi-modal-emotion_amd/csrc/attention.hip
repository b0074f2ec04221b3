// Fused multi-head self-attention (head_dim 64), forward and backward, flash-style: the S x S score matrix never
// exists in memory.  Replaces reference utils/TAVFormer.py:357-383 (fusion encoder, mask added AFTER softmax),
// utils/TAVFormer.py:57-81 (TransformerEncoder, mask before softmax) and HF eager_attention_forward
// (roberta/bert with additive key mask, wav2vec2 / videomae without mask).
//
// Orientation.  Scores are produced TRANSPOSED, S^T[key][query] = K . Q^T, with the key index in the accumulator
// registers (rows 4g+r of a 16x16 tile) and the query on the lane.  The softmax reduction over keys is then in-lane
// plus two cross-lane steps (xor 16, 32), and P^T is already laid out as the k-strided operand of
// O^T[d][query] += V^T[d][key] . P^T[key][query]  (common.h: acc_to_kfrag), whose other operand is read from the
// natural [key][d] LDS image of V with ds_read_b64_tr_b16 (bf16) / ds_read_b32 (f32).  No P round-trip through LDS.
//
// Backward is two deterministic kernels (no atomics):
//   dkdv: one workgroup per 128 keys, S[query][key] with the key on the lane -> P and dS feed dV^T += dO^T P and
//         dK^T += Q^T dS directly;   dq: one workgroup per 128 queries, S^T/dP^T as in the forward -> dQ^T += K^T dS^T.
//
// Mask modes: 0 none; 1 additive key mask before softmax; 2 additive key mask AFTER softmax (the reference quirk):
//   o_q = softmax(s_q) V + c,  c = sum_key mask[key] v[key]   (rank-1, same for every query of a (batch, head)),
//   backward: dV[key] += mask[key] * sum_q dO_q  and  delta_q = dO_q . (o_q - c).
#include "common.h"
#include <type_traits>
#include "tavhip_internal.h"

// A/B switches (tools/ab_build.sh): issue all LDS fragment reads of a tile ahead of its first MFMA
#ifndef TAV_HOIST_FWD
#define TAV_HOIST_FWD 0      // 1 needs 204 VGPRs (two waves per SIMD); without it the kernel fits three (165) and is 4-9 % faster
#endif
// (the same hoisting in the two backward kernels measured -1 %: not built)
// forward kernel, bf16: K / V tiles staged by LDS-DMA (global_load_lds_dwordx4) into swizzled ROW images instead of registers + ds_write
#ifndef TAV_ATT_DMA
#define TAV_ATT_DMA 1      // 0: register staging (322 us at S = 1464, batch 32), 1: DMA one tile ahead (300-312), 2: two tiles ahead, unmasked mode (315)
#endif
#ifndef TAV_DKDV_BQ
#define TAV_DKDV_BQ 64       // bf16 query-tile height of the dK/dV kernel (32 or 64)
#endif
#ifndef TAV_DKDV_FAST32
#define TAV_DKDV_FAST32 1    // the unmasked pre-scaled bf16 dK/dV kernel (video / audio stacks) on 32-query tiles at THREE waves per SIMD (168 VGPRs, no scratch)
#endif

namespace tav {

struct AttnP {
    const char* q; const char* k; const char* v; char* o; char* o_soft;
    const float* mask; float* lse; float* corr;
    const char* dout; char* dq; char* dk; char* dv; float* delta;
    int B, S, nh;
    long ld_q, ld_k, ld_v, ld_o, ld_do, ld_dq, ld_dk, ld_dv;
    float scale;
    int pre;       // q holds q * scale * log2(e) already (the QKV projection's q rows were scaled in the weight copy)
    int tiles;     // workgroups per (batch, head) slice of THIS launch (query tiles: forward, dQ; key tiles: dK/dV); grid = tiles * nh * B, 1-D
};

// Which tile of which (batch, head) slice a workgroup owns.  The grid is one-dimensional and walked through xcd_remap with the tile index
// fastest: every XCD gets a contiguous band of whole (batch, head) slices, so all the tiles that stream one slice's K / V (forward, dQ)
// or Q / dO (dK/dV) run on the SAME XCD, back to back, and share them through that XCD's L2.  With the plain (tiles, nh, B) grid the 12
// tiles of a slice were dealt round-robin over the 8 XCDs and every L2 fetched every slice (round-3 PMC at the video shape: 1310 MB moved
// per forward launch against 288 MB algorithmic).  Resident set per XCD at S = 1464: 32 CUs x 3 workgroups / 12 tiles = 8 slices x 375 KB
// of K + V = 3 MB of the 4 MiB L2.  Placement is a speed matter only (the dispatcher's round-robin is observed, not promised).
#ifndef TAV_ATT_XCD
#define TAV_ATT_XCD 1      // 0: tiles dealt in plain id order (rounds 1-3), kept for A/B builds
#endif
struct AttnTile { int x, head, b; };
TAV_DEV AttnTile attn_tile(const AttnP& p) {
    const int id = TAV_ATT_XCD ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int hb = id / p.tiles, b = hb / p.nh;
    AttnTile w;                    // (pinned to SGPRs: the slice bases feed the scalar operand of the LDS-DMA asm)
    w.x = to_sgpr(id - hb * p.tiles); w.head = to_sgpr(hb - b * p.nh); w.b = to_sgpr(b);
    return w;
}

TAV_DEV float vmax3(float a, float b, float c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }
// v_exp_f32 directly (exp2f() adds denormal-range fix-up code; scores are <= 0 after the max subtraction, flush is fine)
TAV_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <typename T> struct HD {
    static constexpr int ES = ET<T>::ES;
    static constexpr int ROWB = 64 * ES;            // bytes per head row
    static constexpr int ROWCH = ROWB / 16;         // 16-byte chunks per row (8 / 16)
    static constexpr int NSD = ROWCH / 4;           // mma16 steps over d (2 / 4)
    static constexpr int PITCH_N = ROWB + (ES == 2 ? 32 : 16);   // natural image pitch (k-strided reads conflict-free)
    // row image: 16-B chunk index XOR-swizzled with the row.
    //  bf16 (128-B rows): chunk ^ (((row >> 1) & 3) << 1).  ONE image serves both kinds of read, conflict free:
    //    * ds_read_b128 of a 16x16x32 operand (lane group = 8 rows at chunk c + 8 rows at chunk c+1): the 16 slots
    //      8*(row&1) + (chunk ^ sw) are all distinct;
    //    * ds_read_b64_tr_b16 (a 32-lane half = 8 consecutive rows x 32 B): the XOR moves the row's chunk PAIR (bits 1-2),
    //      so the 8 rows fall into 8 distinct 32-B bank windows and the two chunks of a pair stay in order.
    //  f32 (256-B rows): chunk ^ (row & 15) for the row reads; k-strided reads use a second, padded natural image.
    static constexpr bool DUAL = (ES == 2);
    static TAV_DEV int row_off(int row, int chunk) {
        return row * ROWB + ((chunk ^ (ES == 2 ? (((row >> 1) & 3) << 1) : (row & 15))) << 4);
    }
};

// k-strided (transposed) operand fragment straight from the swizzled bf16 ROW image: same k-set as frag_kstrided<bf16>
// (rows krow0 + 4g + q and +16), columns 16*dt + 4p .. +3.  EXEC must be all ones.
TAV_DEV uint4 frag_tr_rowimg(const char* img, int krow0, int dt, int lane) {
    using H = HD<bf16>;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r0 = krow0 + 4 * g + q, r1 = r0 + 16;
    const uint2 lo = lds_read_tr16(img + H::row_off(r0, 2 * dt + (p >> 1)) + 8 * (p & 1));
    const uint2 hi = lds_read_tr16(img + H::row_off(r1, 2 * dt + (p >> 1)) + 8 * (p & 1));
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
// one element of a staged head row (row image if DUAL, natural image otherwise)
template <typename T> TAV_DEV float tile_elem(const char* rowimg, const char* natimg, int row, int d) {
    if constexpr (HD<T>::DUAL) return ET<T>::ld(reinterpret_cast<const T*>(rowimg + HD<T>::row_off(row, d >> 3)) + (d & 7));
    else return ET<T>::ld(reinterpret_cast<const T*>(natimg + row * HD<T>::PITCH_N) + d);
}
template <typename T> TAV_DEV uint4 tile_kfrag(const char* rowimg, const char* natimg, int krow0, int dt, int lane) {
    if constexpr (HD<T>::DUAL) return frag_tr_rowimg(rowimg, krow0, dt, lane);
    else return frag_kstrided<T>(natimg, HD<T>::PITCH_N, krow0, 16 * dt, lane);
}

// Per-thread addressing of the streamed ROWS x 64 tiles of one (batch, head) slice: the byte offset of each of the thread's
// 16-B chunks in tile 0 and its clamp (the same chunk of row S-1) are computed once; a tile costs one add and one min per
// chunk (instead of a 64-bit multiply-add chain), and the load takes the slice base as a scalar operand.
template <typename T, int ROWS, int N>
TAV_DEV void tile_addr_init(unsigned (&off0)[N], unsigned (&offmax)[N], long ld_bytes, int S, int tid) {
    constexpr int ROWCH = HD<T>::ROWCH;
    static_assert(N == ROWS * ROWCH / 256, "chunks per thread");
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const int idx = tid + 256 * e, row = idx / ROWCH, ch = idx - row * ROWCH;
        off0[e] = (unsigned)(row * ld_bytes + ch * 16);
        offmax[e] = (unsigned)((S - 1) * ld_bytes + ch * 16);
    }
}
template <int N>
TAV_DEV void tile_gload(uint4 (&regs)[N], const char* base, const unsigned (&off0)[N], const unsigned (&offmax)[N], unsigned tile_off) {
#pragma unroll
    for (int e = 0; e < N; ++e) {                       // tile_off = t * ROWS * ld_bytes (wave-uniform)
        unsigned o = off0[e] + tile_off;
        o = o < offmax[e] ? o : offmax[e];
        // loaded as a native vector: a plain uint4 struct copy becomes a global->private memcpy that keeps `regs` in scratch
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + o);
        regs[e] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

// zero the staged rows past S -- applied when the tile is written to LDS, NOT at the load (a select on a just-loaded value
// would make the wave wait for its own prefetch)
template <typename T, int ROWS>
TAV_DEV void tile_zero_pad(uint4* regs, int row0, int S, int tid) {
    constexpr int ROWCH = HD<T>::ROWCH, N = ROWS * ROWCH / 256;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const int idx = tid + 256 * e, row = idx / ROWCH;
        if (row0 + row >= S) regs[e] = make_uint4(0, 0, 0, 0);
    }
}
template <typename T, int ROWS>
TAV_DEV void tile_lstore_row(const uint4* regs, char* img, int tid) {
    constexpr int ROWCH = HD<T>::ROWCH, N = ROWS * ROWCH / 256;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const int idx = tid + 256 * e, row = idx / ROWCH, ch = idx - row * ROWCH;
        *reinterpret_cast<uint4*>(img + HD<T>::row_off(row, ch)) = regs[e];
    }
}
template <typename T, int ROWS>
TAV_DEV void tile_lstore_nat(const uint4* regs, char* img, int tid) {
    constexpr int ROWCH = HD<T>::ROWCH, N = ROWS * ROWCH / 256;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const int idx = tid + 256 * e, row = idx / ROWCH, ch = idx - row * ROWCH;
        *reinterpret_cast<uint4*>(img + row * HD<T>::PITCH_N + ch * 16) = regs[e];
    }
}

// lane (g,i) fragment of a row held in global memory: chunk 4s+g of row `r` (clamped)
template <typename T>
TAV_DEV void row_frags_gload(uint4* f, const char* base, long ld_bytes, int r, int S, int g) {
    if (r >= S) r = S - 1;
#pragma unroll
    for (int s = 0; s < HD<T>::NSD; ++s) f[s] = *reinterpret_cast<const uint4*>(base + (long)r * ld_bytes + (4 * s + g) * 16);
}

// LDS-DMA staging of two 64-row x 64-element bf16 tiles (A, B) of one (batch, head) slice into swizzled row images (HD<bf16>::row_off:
// linear per 8 rows, so one DMA instruction fills 8 rows x 8 slots with the XOR applied to the SOURCE chunk a lane fetches).  Wave w fills
// rows [16w, 16w + 16) of both images.  The lane offsets are constants of the kernel and the tile walks on the SCALAR bases: a regular
// tile costs no vector instruction for its addresses.  RAGGED (the last tile when S % 64 != 0): rows past S-1 step back to row S-1 --
// finite data; the consumer gives those rows / keys weight zero (-inf logits).
template <int ROWS = 64> struct PairDmaT {
    static constexpr int NP = ROWS / 32;       // 1-KiB pieces (8 rows) per wave and operand
    const char* a; const char* b;
    unsigned lda_b, ldb_b;             // bytes per row
    unsigned offa[NP], offb[NP];
    unsigned lds_a, lds_b;             // LDS byte addresses of this wave's rows in slot 0 (wave-uniform)
    int S, wave;
    TAV_DEV void init(const char* a_, const char* b_, long lda_bytes, long ldb_bytes, int S_, const void* lds_img_a, const void* lds_img_b, int tid) {
        a = a_; b = b_; lda_b = (unsigned)lda_bytes; ldb_b = (unsigned)ldb_bytes; S = S_;
        const int lane = tid & 63;
        wave = tid >> 6;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int row = wave * 8 * NP + j * 8 + (lane >> 3), slot = lane & 7;
            const int ch = slot ^ (((row >> 1) & 3) << 1);
            offa[j] = (unsigned)row * lda_b + (unsigned)(ch * 16);      // (32-bit: the host checks that a slice fits)
            offb[j] = (unsigned)row * ldb_b + (unsigned)(ch * 16);
        }
        lds_a = __builtin_amdgcn_readfirstlane(lds_addr(lds_img_a) + wave * 8 * NP * 128);
        lds_b = __builtin_amdgcn_readfirstlane(lds_addr(lds_img_b) + wave * 8 * NP * 128);
    }
    template <bool RAGGED> TAV_DEV void issue(int t, unsigned slot_bytes) const {
        const char* ab = a + (size_t)t * (unsigned)ROWS * lda_b;
        const char* bb = b + (size_t)t * (unsigned)ROWS * ldb_b;
        unsigned oa[NP], ob[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) { oa[j] = offa[j]; ob[j] = offb[j]; }
        if constexpr (RAGGED) {
            int ln = threadIdx.x & 63;
            asm volatile("" : "+v"(ln));                 // (re-materialised here: nothing of this is hoisted above the loop and kept live)
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int over = t * ROWS + wave * 8 * NP + j * 8 + (ln >> 3) - (S - 1);
                if (over > 0) { oa[j] -= (unsigned)over * lda_b; ob[j] -= (unsigned)over * ldb_b; }
            }
        }
        if constexpr (NP == 2) glds16_x4(ab, bb, oa[0], oa[1], ob[0], ob[1], lds_a + slot_bytes, lds_b + slot_bytes);
        else glds16_x2(ab, bb, oa[0], ob[0], lds_a + slot_bytes, lds_b + slot_bytes);
    }
};
using PairDma = PairDmaT<64>;

// ================================================================================================= forward
// ablation switches for the forward kernel (tools/ab_build.sh; timing experiments only, results are wrong with any of them set)
#ifdef TAV_ABL_ATT_NOMFMA
#define ATT_FWD_MMA(a, b, c) do { (c)[0] += __uint_as_float((a).x ^ (b).x); } while (0)
#else
#define ATT_FWD_MMA(a, b, c) mma16<T>(a, b, c)
#endif
#ifdef TAV_ABL_ATT_NOEXP
#define ATT_FWD_EXP2(x) ((x) * 0.5f)
#else
#define ATT_FWD_EXP2(x) fast_exp2(x)
#endif
#ifdef TAV_ABL_ATT_NOGLOAD
constexpr bool ATT_ABL_NOGLOAD = true;
#else
constexpr bool ATT_ABL_NOGLOAD = false;
#endif
#ifdef TAV_ABL_ATT_NOBAR
constexpr bool ATT_ABL_NOBAR = true;
#else
constexpr bool ATT_ABL_NOBAR = false;
#endif
#ifndef TAV_ATT_FWD_OCC
#define TAV_ATT_FWD_OCC 3      // waves per SIMD the forward kernel is compiled for (register budget 512 / OCC; 4 spills and halves the speed)
#endif
// Lazy running maximum (PRE kernels): the accumulated O / l are rescaled only when a tile's scores exceed the reference exponent by more
// than this many binary orders of magnitude; until then p = exp2(s - m_ref) may grow to 2^THR, which f32 accumulators and bf16 operands
// (8-bit exponent) carry without loss.
#ifndef TAV_ATT_LAZY_THR
#define TAV_ATT_LAZY_THR 12.0f
#endif
// PRE: q is pre-multiplied by scale * log2(e) (the engine folds the factor into the q rows of the QKV weight copy: no extra rounding), so
// S^T = K q~^T is already the exp2-domain logit.  The unmasked, full tiles then take a fast path built for the VALU issue port, which is
// what bounds this kernel at head dim 64 (rocprof r03: the port is 81 % busy, the matrix pipe 39 %):
//   * the running reference exponent enters as the C operand of the first QK^T MFMA (-m on every row of the query's column), so the
//     accumulators come out as s - m and p = exp2(acc): no fma per score;
//   * O / l are rescaled only when the tile maximum passes the reference by TAV_ATT_LAZY_THR (rare after the first tile);
//   * ring slot, LDS addresses and the DMA source walk are compile-time / scalar: the loop body is instantiated once per slot.
#ifndef TAV_ATT_FWD_NQ
#define TAV_ATT_FWD_NQ 2     // 16-query tiles per wave of the unmasked pre-scaled bf16 forward (3: K/V fragment reads, DMA and barriers amortised over 1.5x the MFMAs; two waves per SIMD)
#endif
template <typename T, int MODE, bool PRE> constexpr int fwd_nq() { return (sizeof(T) == 2 && MODE == 0 && PRE) ? TAV_ATT_FWD_NQ : 2; }
template <typename T, int MODE, bool PRE>
__global__ __launch_bounds__(256, (fwd_nq<T, MODE, PRE>() > 2 ? 2 : TAV_ATT_FWD_OCC)) void attn_fwd_kernel(const AttnP p) {
    constexpr int NQ = fwd_nq<T, MODE, PRE>();             // 16-query tiles per wave (2; 3 = 48 queries per wave, 192 per workgroup)
    using H = HD<T>;
    constexpr int ES = H::ES, NSD = H::NSD, KSTEP = ET<T>::KSTEP, BKV = 64;
    constexpr int NCH = BKV * H::ROWCH / 256;
    // bf16: both tiles live in the swizzled row image (H::row_off), which is linear per 8 rows -- one DMA instruction fills 8 rows x 8
    // slots, the XOR applied to the SOURCE chunk a lane fetches -- and serves the row reads (K) as well as the transposed reads (V,
    // frag_tr_rowimg).  f32 keeps the register path and the padded natural V image.
    constexpr bool DMA = (ES == 2) && TAV_ATT_DMA;
    constexpr int KROW_B = BKV * H::ROWB, VNAT_B = DMA ? BKV * H::ROWB : BKV * H::PITCH_N;
    constexpr int BUF_B = KROW_B + VNAT_B + 2 * BKV * 4;
    constexpr int NBUF = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + NBUF * BUF_B);   // [4][64] + [64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const AttnTile wg = attn_tile(p);
    const int head = wg.head, b = wg.b, S = p.S;
    const int q0 = wg.x * (64 * NQ) + wave * (16 * NQ);
    const long hoff = (long)head * 64 * ES;
    const char* Qb = p.q + (long)b * S * p.ld_q * ES + hoff;
    const char* Kb = p.k + (long)b * S * p.ld_k * ES + hoff;
    const char* Vb = p.v + (long)b * S * p.ld_v * ES + hoff;
    const float* maskb = p.mask ? p.mask + (long)b * S : nullptr;

    uint4 qf[NQ][NSD];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) row_frags_gload<T>(qf[qt], Qb, p.ld_q * ES, q0 + 16 * qt + i, S, g);

    // softmax runs in the exp2 domain: t = s*scale*log2(e) + mask*log2(e); p = exp2(t - m).  The row sums l come out of the MFMA
    // pipe (a ones-row operand times P^T) instead of 32 VALU adds + shuffles per tile: the kernel is VALU-issue bound.
    constexpr float LOG2E = 1.4426950408889634f;
    const float c2 = PRE ? 1.0f : p.scale * LOG2E;
    const uint4 ones = (ES == 2) ? make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u)
                                 : make_uint4(0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u);
    uint4 ones_v = ones;
    asm volatile("" : "+v"(ones_v.x), "+v"(ones_v.y), "+v"(ones_v.z), "+v"(ones_v.w));   // held in VGPRs (else re-moved from SGPRs every tile)
    // lane constants of the LDS fragment reads (bf16 row images): K row read of k-step s, V transposed read of d-tile dt; the tile's key
    // rows and the ring slot are immediates / one scalar add away (row_off's XOR only looks at row bits 1-2, which 16-row steps leave alone)
    unsigned koff[NSD], vtoff[4];
    if constexpr (ES == 2) {
#pragma unroll
        for (int s = 0; s < NSD; ++s) { koff[s] = (unsigned)H::row_off(i, 4 * s + g); asm volatile("" : "+v"(koff[s])); }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            vtoff[dt] = (unsigned)(H::row_off(4 * g + (i >> 2), 2 * dt + ((i & 3) >> 1)) + 8 * (i & 1));
            asm volatile("" : "+v"(vtoff[dt]));
        }
    }
    float m_run[NQ];                                          // reference exponent of O / l per query (lane i, both q tiles)
    f32x4 mneg[NQ];   // PRE fast path: -m_run on all four rows = the C operand of QK^T
    f32x4 lacc[NQ];   // every register = sum_key P[key][q] (rows of the ones operand)
    f32x4 oacc[4][NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
        m_run[qt] = -1e30f;
        mneg[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        lacc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) oacc[a][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float corr_part = 0.f;

    const int nkt = (S + BKV - 1) / BKV;
    uint4 rk[NCH], rv[NCH];
    float r_mask = 0.f;
    // gload only ISSUES loads (no arithmetic on a loaded value: that would put a vmcnt wait -- a drain of the whole prefetch --
    // right behind the issue); lstore, one tile of compute later, turns the raw mask value into the per-key additive terms.
    unsigned k_off0[NCH], k_max[NCH], v_off0[NCH], v_max[NCH];
    if constexpr (!DMA) {
        tile_addr_init<T, BKV>(k_off0, k_max, p.ld_k * ES, S, tid);
        tile_addr_init<T, BKV>(v_off0, v_max, p.ld_v * ES, S, tid);
    }
    const unsigned kstep_b = (unsigned)(BKV * p.ld_k * ES), vstep_b = (unsigned)(BKV * p.ld_v * ES);
    // DMA geometry: wave w fills rows [16w, 16w + 16) of both images, 8 rows per instruction; lane -> (row, slot), source chunk = slot ^ swizzle.
    // The lane offsets are constants of the kernel; the tile walks on the SCALAR base (one s_add per operand), so a regular tile costs no
    // vector instruction for its addresses.  Only the ragged last tile clamps rows past S (to row S-1: finite data, their scores are -inf).
    PairDma kv;
    if constexpr (DMA) kv.init(Kb, Vb, p.ld_k * ES, p.ld_v * ES, S, smem, smem + KROW_B, tid);
    auto dma = [&](int t, int buf, auto ragged_tag) __attribute__((always_inline)) {
        kv.template issue<decltype(ragged_tag)::value != 0>(t, (unsigned)(buf * BUF_B));
    };
    auto gload = [&](int t) {
        if constexpr (!DMA) {
            tile_gload(rk, Kb, k_off0, k_max, t * kstep_b);
            tile_gload(rv, Vb, v_off0, v_max, t * vstep_b);
        }
        if (MODE != 0 && tid < BKV) {
            int key = t * BKV + tid;
            key = key < S ? key : S - 1;
            r_mask = maskb[key];
        }
    };
    // per-key additive terms of tile t -> slot buf.  Without a mask only the ragged last tile reads them.
    auto lstore = [&](int t, int buf) {
        char* base = smem + buf * BUF_B;
        if constexpr (!DMA) {
            tile_lstore_row<T, BKV>(rk, base, tid);
            tile_lstore_nat<T, BKV>(rv, base + KROW_B, tid);
        }
        if ((MODE != 0 || t == nkt - 1) && tid < BKV) {
            const bool ok = t * BKV + tid < S;
            float* f = reinterpret_cast<float*>(base + KROW_B + VNAT_B);
            f[tid] = ok ? (MODE == 1 ? r_mask * 1.4426950408889634f : 0.f) : -INFINITY;   // already in the exp2 domain
            f[BKV + tid] = (MODE == 2 && ok) ? r_mask : 0.f;
        }
    };
    using TagNo = std::integral_constant<int, 0>;
    using TagYes = std::integral_constant<int, 1>;
    if constexpr (DMA) { if (nkt == 1) dma(0, 0, TagYes{}); else dma(0, 0, TagNo{}); }
    gload(0); lstore(0, 0);
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int s = 0; s < NSD; ++s) settle(qf[qt][s]);
    if constexpr (DMA) wait_vmcnt0();
    __syncthreads();

    // One K/V tile.  KADD (compile time): the tile adds per-key terms (pre-softmax mask, -inf past S): every tile under MODE 1, else only the
    // last one, which is peeled off the loop -- the loop body itself has a single softmax path.  NEXT: 0 no prefetch, 1 regular, 2 ragged tile.
    auto tile_body = [&](const int t, auto kadd_tag, auto next_tag) __attribute__((always_inline)) {
        constexpr bool KADD = decltype(kadd_tag)::value != 0;
        constexpr int NEXT = decltype(next_tag)::value;
        const int cur = t & 1, nxt = cur ^ 1;
        if constexpr (NEXT != 0 && !ATT_ABL_NOGLOAD) {           // (slot nxt was last read in iteration t-1, behind a barrier)
            if constexpr (DMA) dma(t + 1, nxt, std::integral_constant<int, NEXT == 2>{});
            gload(t + 1);
        }
        const char* Krow = smem + cur * BUF_B;
        const char* Vnat = Krow + KROW_B;
        const float* kadd = reinterpret_cast<const float*>(Vnat + VNAT_B);
        const float* cm = kadd + BKV;

        f32x4 sacc[4][NQ];
        // S^T = K q^T: the accumulators start at cinit (fast path: -m_ref on every row, so s - m_ref comes out of the MFMA chain; else zero)
        auto qk = [&](const f32x4 (&cinit)[NQ]) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < NSD; ++s)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    uint4 a;
                    if constexpr (ES == 2) a = *reinterpret_cast<const uint4*>(Krow + koff[s] + kt * 16 * H::ROWB);
                    else a = *reinterpret_cast<const uint4*>(Krow + H::row_off(16 * kt + i, 4 * s + g));
#pragma unroll
                    for (int qt = 0; qt < NQ; ++qt) {
                        if (s == 0) sacc[kt][qt] = cinit[qt];
                        ATT_FWD_MMA(a, qf[qt][s], sacc[kt][qt]);
                    }
                }
        };
        if constexpr (PRE && !KADD) {
            qk(mneg);
            // sacc = s - m_ref.  Tile maximum per query; rescale only when it passes the reference by more than THR (or on the first tile,
            // where the reference is still undefined: mneg = 0, so sacc are the raw logits).
            const bool first = (t == 0);
            float mx[NQ];
            bool grow = first;
#pragma unroll
            for (int qt = 0; qt < NQ; ++qt) {
                // v_maximum3_f32 (IEEE maximum: no canonicalising v_max in front of MFMA results, as fmaxf needs): 8 instructions for 16 values
                float m0 = vmax3(sacc[0][qt][0], sacc[0][qt][1], sacc[0][qt][2]);
                m0 = vmax3(m0, sacc[0][qt][3], sacc[1][qt][0]);
#pragma unroll
                for (int kt = 1; kt < 4; ++kt) {
                    if (kt > 1) m0 = vmax3(m0, sacc[kt - 1][qt][3], sacc[kt][qt][0]);
                    m0 = vmax3(m0, sacc[kt][qt][1], sacc[kt][qt][2]);
                }
                m0 = __builtin_elementwise_maximum(m0, sacc[3][qt][3]);
                mx[qt] = max_over_row_groups(m0);
                grow |= mx[qt] > TAV_ATT_LAZY_THR;
            }
            if (__any(grow)) {                                     // rare after the first tile: move the reference, THEN take the common path
#pragma unroll
                for (int qt = 0; qt < NQ; ++qt) {
                    const float shift = first ? mx[qt] : fmaxf(mx[qt], 0.f);         // new reference = old + shift (never lowered after tile 0)
                    m_run[qt] = first ? shift : m_run[qt] + shift;
                    mneg[qt] = f32x4{-m_run[qt], -m_run[qt], -m_run[qt], -m_run[qt]};
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) sacc[kt][qt] -= shift;
                    if (!first) {                                   // (O = l = 0 on the first tile; exp2(-shift) could overflow there)
                        const float al = ATT_FWD_EXP2(-shift);
                        lacc[qt] *= al;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) oacc[dt][qt] *= al;
                    }
                }
            }
#pragma unroll
            for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[kt][qt][r] = ATT_FWD_EXP2(sacc[kt][qt][r]);
        } else {
            f32x4 zero2[NQ];
#pragma unroll
            for (int qt = 0; qt < NQ; ++qt) zero2[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            qk(zero2);
            // running max per query (= per lane i, both q tiles), then p = exp2(s*c2 + kadd - m).  The per-key additive term is zero
            // except under a pre-softmax mask (MODE 1) and on the ragged last tile (-inf past S): only those tiles pay for it;
            // the others take the max over the raw scores and fold scale and max into one fma per element.
            float alpha[NQ];
            bool moved = false;
            if constexpr (KADD) {
                float mx[NQ];
#pragma unroll
                for (int qt = 0; qt < NQ; ++qt) mx[qt] = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const f32x4 ka = *reinterpret_cast<const f32x4*>(kadd + 16 * kt + 4 * g);
#pragma unroll
                    for (int qt = 0; qt < NQ; ++qt) {
                        sacc[kt][qt] = PRE ? sacc[kt][qt] + ka : sacc[kt][qt] * c2 + ka;
                        mx[qt] = fmaxf(fmaxf(mx[qt], sacc[kt][qt][0]), fmaxf(sacc[kt][qt][1], fmaxf(sacc[kt][qt][2], sacc[kt][qt][3])));
                    }
                }
#pragma unroll
                for (int qt = 0; qt < NQ; ++qt) {
                    const float m_new = fmaxf(m_run[qt], max_over_row_groups(mx[qt]));
                    alpha[qt] = ATT_FWD_EXP2(m_run[qt] - m_new);
                    moved |= m_new > m_run[qt];
                    m_run[qt] = m_new;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc[kt][qt][r] = ATT_FWD_EXP2(sacc[kt][qt][r] - m_new);
                }
            } else {
#pragma unroll
                for (int qt = 0; qt < NQ; ++qt) {
                    float mx = fmaxf(fmaxf(sacc[0][qt][0], sacc[0][qt][1]), fmaxf(sacc[0][qt][2], sacc[0][qt][3]));
#pragma unroll
                    for (int kt = 1; kt < 4; ++kt)
                        mx = fmaxf(fmaxf(mx, sacc[kt][qt][0]), fmaxf(sacc[kt][qt][1], fmaxf(sacc[kt][qt][2], sacc[kt][qt][3])));
                    const float m_new = fmaxf(m_run[qt], max_over_row_groups(mx) * c2);
                    alpha[qt] = ATT_FWD_EXP2(m_run[qt] - m_new);
                    moved |= m_new > m_run[qt];
                    m_run[qt] = m_new;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc[kt][qt][r] = ATT_FWD_EXP2(__builtin_fmaf(sacc[kt][qt][r], c2, -m_new));
                }
            }
            if (__any(moved)) {        // wave-uniform: after the first tiles the running max rarely moves, skip 34 multiplies
#pragma unroll
                for (int qt = 0; qt < NQ; ++qt) {
                    lacc[qt] *= alpha[qt];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) oacc[dt][qt] *= alpha[qt];
                    if constexpr (PRE && MODE != 1) mneg[qt] = f32x4{-m_run[qt], -m_run[qt], -m_run[qt], -m_run[qt]};
                }
            }
        }
        // O^T += V^T P^T ;  l += 1^T P^T
        constexpr int NKS = BKV / KSTEP;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            uint4 pb[NQ];
#pragma unroll
            for (int qt = 0; qt < NQ; ++qt) {
                f32x4 tl[2];
                tl[0] = sacc[ks * ET<T>::ACC_TILES][qt];
                tl[1] = sacc[ks * ET<T>::ACC_TILES + ET<T>::ACC_TILES - 1][qt];
                pb[qt] = acc_to_kfrag<T>(tl);
            }
#pragma unroll
            for (int qt = 0; qt < NQ; ++qt) ATT_FWD_MMA(ones_v, pb[qt], lacc[qt]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint4 a;
                if constexpr (DMA) {                          // = frag_tr_rowimg(Vnat, ks * KSTEP, dt, lane) with the lane part precomputed
                    const char* vt = Vnat + vtoff[dt] + ks * KSTEP * H::ROWB;
                    const uint2 lo = lds_read_tr16(vt), hi = lds_read_tr16(vt + 16 * H::ROWB);
                    a = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
                else a = frag_kstrided<T>(Vnat, H::PITCH_N, ks * KSTEP, 16 * dt, lane);
#pragma unroll
                for (int qt = 0; qt < NQ; ++qt) ATT_FWD_MMA(a, pb[qt], oacc[dt][qt]);
            }
        }
        if (MODE == 2) {   // c[d] += sum_key mask[key] * V[key][d]; thread -> (d = tid & 63, 16 keys of this tile)
            const int d = tid & 63, kq = tid >> 6;
#pragma unroll 4
            for (int kk = 0; kk < 16; ++kk) {
                const int key = kq * 16 + kk;
                if constexpr (DMA) corr_part += cm[key] * ET<T>::ld(reinterpret_cast<const T*>(Vnat + H::row_off(key, d >> 3)) + (d & 7));
                else corr_part += cm[key] * ET<T>::ld(reinterpret_cast<const T*>(Vnat + key * H::PITCH_N) + d);
            }
        }
        if constexpr (NEXT != 0 && !ATT_ABL_NOGLOAD && (MODE != 0 || NEXT == 2 || !DMA)) lstore(t + 1, nxt);   // (unmasked: only the last tile has per-key terms)
        if constexpr (DMA) wait_vmcnt0();                     // tile t+1 has landed
        if (!ATT_ABL_NOBAR) __syncthreads();
    };
    using K0 = std::integral_constant<int, (MODE == 1) ? 1 : 0>;      // middle tiles carry per-key terms only under the pre-softmax mask
    int t = 0;
    for (; t + 2 < nkt; ++t) tile_body(t, K0{}, std::integral_constant<int, 1>{});
    if (t + 1 < nkt) { tile_body(t, K0{}, std::integral_constant<int, 2>{}); ++t; }
    tile_body(t, TagYes{}, std::integral_constant<int, 0>{});          // the last tile: -inf past S

    if (MODE == 2) {
        red[wave * 64 + lane] = corr_part;
        __syncthreads();
        if (tid < 64) {
            const float c = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
            red[256 + tid] = c;
            if (wg.x == 0) p.corr[((long)b * p.nh + head) * 64 + tid] = c;
        }
        __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
        const float l = lacc[qt][0];          // the MFMA already summed over all keys (all lane groups)
        const int q = q0 + 16 * qt + i;
        if (q < S) {
            const float inv = 1.f / l;
            T* orow = reinterpret_cast<T*>(p.o) + ((long)b * S + q) * p.ld_o + head * 64;
            T* srow = reinterpret_cast<T*>(p.o_soft) + ((long)b * S + q) * p.ld_o + head * 64;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 v = oacc[dt][qt] * inv;
                if (MODE == 2) {   // keep softmax(s)·v on its own: o - c cancels catastrophically once |mask| ~ 6.5e4
                    st4(srow + 16 * dt + 4 * g, v);
                    v += *reinterpret_cast<const f32x4*>(red + 256 + 16 * dt + 4 * g);
                }
                st4(orow + 16 * dt + 4 * g, v);
            }
            if (g == 0) p.lse[((long)b * p.nh + head) * S + q] = (m_run[qt] + log2f(l)) * 0.6931471805599453f;   // natural-log LSE
        }
    }
}

// ================================================================================================= forward, 32x32x16 MFMA shape
// The same algorithm as attn_fwd_kernel<bf16, 0, true> (no mask, pre-scaled q, lazy reference exponent, LDS-DMA ring) on
// v_mfma_f32_32x32x16_bf16: a wave owns 32 queries x 64 keys per tile in 16 MFMA issues instead of 36, every lane's 32 scores of a tile belong to
// ONE query (the running maximum needs one cross-lane step instead of two, the row sum none inside the loop), and the reference exponent is
// one 16-register C operand.  The kernel is bound by instructions issued per SIMD (profiles/r03_experiments.md): fewer, longer MFMAs.
//   S^T tile  (keys x queries) = K[32 keys][16 d] . Q^T[16 d][32 q]   lane (q = l % 32, h = l / 32): acc r <-> key 8 (r / 4) + 4 h + r % 4
//   O^T tile  (d x queries)   += V^T[32 d][16 keys] . P^T[16 keys][32 q]; the k-slots of a 16-key chunk C are the keys the lane's accumulators
//             8 c' .. 8 c' + 7 hold (16 C + 4 h + {0..3}, 16 C + 8 + 4 h + {0..3}) -- any assignment works as long as V^T's fragments use the same;
//             M-tile T covers d in {16 T .. 16 T + 15} u {32 + 16 T .. 47 + 16 T}: two chunk pairs whose XOR-swizzled positions never share a
//             32-byte bank window inside the 4-row block a transposed read fetches (HD<bf16>::row_off).
// MEASURED SLOWER than the 16x16x32 kernel (video shape, batch 32: 254.6-263 us against 246.5-249 us; profiles/r03_experiments.md section 14)
// and therefore not compiled by default: -DTAV_ATT_FWD32=1 builds it and routes the unmasked pre-scaled bf16 forward through it (all kernel
// checks pass on it).
#ifndef TAV_ATT_FWD32
#define TAV_ATT_FWD32 0
#endif
#if TAV_ATT_FWD32
typedef __attribute__((ext_vector_type(16))) float f32x16;
TAV_DEV void mma32(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
#ifndef TAV_ATT_FWD32_OCC
#define TAV_ATT_FWD32_OCC 3
#endif
__global__ __launch_bounds__(256, TAV_ATT_FWD32_OCC) void attn_fwd32_kernel(const AttnP p) {
    using T = bf16;
    using H = HD<T>;
    constexpr int BKV = 64;
    constexpr int KROW_B = BKV * H::ROWB, VNAT_B = BKV * H::ROWB;
    constexpr int BUF_B = KROW_B + VNAT_B + 2 * BKV * 4;       // (the layout of attn_fwd_kernel: K image, V image, per-key terms)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 31, h = lane >> 5;
    const AttnTile wg = attn_tile(p);
    const int head = wg.head, b = wg.b, S = p.S;
    const int q0 = wg.x * 128 + wave * 32;
    const long hoff = (long)head * 64 * 2;
    const char* Qb = p.q + (long)b * S * p.ld_q * 2 + hoff;
    const char* Kb = p.k + (long)b * S * p.ld_k * 2 + hoff;
    const char* Vb = p.v + (long)b * S * p.ld_v * 2 + hoff;

    uint4 qf[4];                                              // Q[q][16 s + 8 h .. + 7]
    {
        int r = q0 + lq; r = r < S ? r : S - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const uint4*>(Qb + (long)r * p.ld_q * 2 + (2 * s + h) * 16);
    }
    // lane constants of the LDS reads; key rows / chunks of 16 keys / the ring slot are immediates or one scalar add away
    unsigned koff[4], vtoff[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) { koff[s] = (unsigned)H::row_off(lq, 2 * s + h); asm volatile("" : "+v"(koff[s])); }
    {
        const int g = lane >> 4, i = lane & 15, q4 = i >> 2, pp = i & 3;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int d0 = ((g & 1) ? 32 : 0) + 16 * tt;
            vtoff[tt] = (unsigned)(H::row_off(4 * h + q4, d0 / 8 + (pp >> 1)) + 8 * (pp & 1));
            asm volatile("" : "+v"(vtoff[tt]));
        }
    }
    float m_run = -1e30f;
    f32x16 mneg, oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { mneg[r] = 0.f; oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    f32x2_t l2 = {0.f, 0.f};                                  // this lane's share of sum_key p (its 32 keys per tile); halves joined at the end

    const int nkt = (S + BKV - 1) / BKV;
    PairDma kv;
    kv.init(Kb, Vb, p.ld_k * 2, p.ld_v * 2, S, smem, smem + KROW_B, tid);
    auto dma = [&](int t, int buf, auto ragged_tag) __attribute__((always_inline)) {
        kv.template issue<decltype(ragged_tag)::value != 0>(t, (unsigned)(buf * BUF_B));
    };
    auto store_kadd = [&](int t, int buf) {                   // -inf for the keys past S of the (ragged) last tile
        if (tid < BKV) reinterpret_cast<float*>(smem + buf * BUF_B + KROW_B + VNAT_B)[tid] = (t * BKV + tid < S) ? 0.f : -INFINITY;
    };
    using TagNo = std::integral_constant<int, 0>;
    using TagYes = std::integral_constant<int, 1>;
    if (nkt == 1) { dma(0, 0, TagYes{}); store_kadd(0, 0); } else dma(0, 0, TagNo{});
#pragma unroll
    for (int s = 0; s < 4; ++s) settle(qf[s]);
    wait_vmcnt0();
    __syncthreads();

    // One K/V tile.  LAST (compile time): the tile adds the per-key terms (-inf past S).  NEXT: 0 no prefetch, 1 regular, 2 ragged tile.
    auto tile_body = [&](const int t, auto last_tag, auto next_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value != 0;
        constexpr int NEXT = decltype(next_tag)::value;
        const int cur = t & 1, nxt = cur ^ 1;
        if constexpr (NEXT != 0) dma(t + 1, nxt, std::integral_constant<int, NEXT == 2>{});
        const char* Krow = smem + cur * BUF_B;
        const char* Vimg = Krow + KROW_B;
        f32x16 sacc[2];
        uint4 kf[2][4], vf[4][2];
        // all eight K fragments are requested before the first MFMA, all sixteen V^T fragments behind the QK^T MFMAs: they land under the
        // softmax arithmetic (a read issued right in front of the MFMA that needs it exposes the LDS latency sixteen times per tile)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) kf[kt][s] = *reinterpret_cast<const uint4*>(Krow + koff[s] + kt * 32 * H::ROWB);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                if (s == 0) sacc[kt] = mneg;
                mma32(kf[kt][s], qf[s], sacc[kt]);
            }
#pragma unroll
        for (int C = 0; C < 4; ++C)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const char* vt = Vimg + vtoff[tt] + C * 16 * H::ROWB;
                const uint2 lo = lds_read_tr16(vt), hi = lds_read_tr16(vt + 8 * H::ROWB);
                vf[C][tt] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LAST) {
            const float* kadd = reinterpret_cast<const float*>(Vimg + VNAT_B);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 ka = *reinterpret_cast<const f32x4*>(kadd + 32 * kt + 8 * j + 4 * h);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[kt][4 * j + r] += ka[r];
                }
        }
        // sacc = s - m_ref.  Tile maximum of the lane's query over its 32 keys, then the other half's; the reference moves only when the
        // maximum passes it by more than THR (or on the first tile, where it is still undefined: mneg = 0, sacc are the raw logits).
        const bool first = (t == 0);
        float m0 = vmax3(sacc[0][0], sacc[0][1], sacc[0][2]);
#pragma unroll
        for (int r = 3; r + 1 < 16; r += 2) m0 = vmax3(m0, sacc[0][r], sacc[0][r + 1]);
        m0 = vmax3(m0, sacc[0][15], sacc[1][0]);
#pragma unroll
        for (int r = 1; r + 1 < 16; r += 2) m0 = vmax3(m0, sacc[1][r], sacc[1][r + 1]);
        m0 = __builtin_elementwise_maximum(m0, sacc[1][15]);
        {
            const unsigned u = __float_as_uint(m0);
            auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            m0 = vmax_raw(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
        }
        if (__any(first || m0 > TAV_ATT_LAZY_THR)) {            // rare after the first tile: move the reference, THEN take the common path
            const float shift = first ? m0 : fmaxf(m0, 0.f);
            m_run = first ? shift : m_run + shift;
#pragma unroll
            for (int r = 0; r < 16; ++r) { mneg[r] = -m_run; sacc[0][r] -= shift; sacc[1][r] -= shift; }
            if (!first) {
                const float al = fast_exp2(-shift);
                l2 *= al;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[0][r] *= al; oacc[1][r] *= al; }
            }
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] = fast_exp2(sacc[kt][r]);
        {
            f32x2_t la = {0.f, 0.f}, lb = {0.f, 0.f};            // (two chains: sixteen dependent packed adds are a latency chain of their own)
#pragma unroll
            for (int r = 0; r < 16; r += 2) { la += f32x2_t{sacc[0][r], sacc[0][r + 1]}; lb += f32x2_t{sacc[1][r], sacc[1][r + 1]}; }
            l2 += la + lb;
        }
        // O^T += V^T P^T, 16 keys per MFMA: chunk C = 2 kt + c' takes the accumulators 8 c' .. 8 c' + 7 of key tile kt
#pragma unroll
        for (int C = 0; C < 4; ++C) {
            const int kt = C >> 1, c8 = 8 * (C & 1);
            const uint4 pb = make_uint4(pack_bf16x2(sacc[kt][c8 + 0], sacc[kt][c8 + 1]), pack_bf16x2(sacc[kt][c8 + 2], sacc[kt][c8 + 3]),
                                        pack_bf16x2(sacc[kt][c8 + 4], sacc[kt][c8 + 5]), pack_bf16x2(sacc[kt][c8 + 6], sacc[kt][c8 + 7]));
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) mma32(vf[C][tt], pb, oacc[tt]);
        }
        if constexpr (NEXT == 2) store_kadd(t + 1, nxt);
        wait_vmcnt0();                                        // tile t+1 has landed
        __syncthreads();
    };
    int t = 0;
    for (; t + 2 < nkt; ++t) tile_body(t, TagNo{}, std::integral_constant<int, 1>{});
    if (t + 1 < nkt) { tile_body(t, TagNo{}, std::integral_constant<int, 2>{}); ++t; }
    tile_body(t, TagYes{}, std::integral_constant<int, 0>{});

    float l = l2[0] + l2[1];
    {
        const unsigned u = __float_as_uint(l);
        auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        l = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
    }
    const int q = q0 + lq;
    if (q < S) {
        const float inv = 1.f / l;
        T* orow = reinterpret_cast<T*>(p.o) + ((long)b * S + q) * p.ld_o + head * 64;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {                     // accumulator rows 8 j + 4 h + {0..3} of M-tile tt
                const int d = (j < 2 ? 8 * j : 32 + 8 * (j - 2)) + 4 * h + 16 * tt;
                st4(orow + d, f32x4{oacc[tt][4 * j], oacc[tt][4 * j + 1], oacc[tt][4 * j + 2], oacc[tt][4 * j + 3]} * inv);
            }
        if (h == 0) p.lse[((long)b * p.nh + head) * S + q] = (m_run + log2f(l)) * 0.6931471805599453f;
    }
}

#endif   // TAV_ATT_FWD32

// ================================================================================================= backward: dK, dV
// query-tile height of the dK/dV kernel: 64 for bf16 (half as many barriers and staging round trips per MFMA as 32), 32 for
// f32 (register budget)
template <typename T, int MODE, bool PRE> constexpr bool dkdv_fast32() { return TAV_DKDV_FAST32 && sizeof(T) == 2 && MODE == 0 && PRE; }
template <typename T, int MODE, bool PRE> constexpr int dkdv_bq() { return sizeof(T) == 2 ? (dkdv_fast32<T, MODE, PRE>() ? 32 : TAV_DKDV_BQ) : 32; }

// (waves per SIMD the backward kernels are compiled for: at 3 both spill -- 48..256 B of scratch -- which halved the forward's speed when tried there)
#ifndef TAV_ATT_DKDV_OCC
#define TAV_ATT_DKDV_OCC 2
#endif
#ifndef TAV_ATT_DQ_OCC
#define TAV_ATT_DQ_OCC 2
#endif
template <typename T, int MODE, bool PRE>
__global__ __launch_bounds__(256, (dkdv_fast32<T, MODE, PRE>() ? 3 : TAV_ATT_DKDV_OCC)) void attn_bwd_dkdv_kernel(const AttnP p) {
    using H = HD<T>;
    constexpr int ES = H::ES, NSD = H::NSD, KSTEP = ET<T>::KSTEP, BQ = dkdv_bq<T, MODE, PRE>(), NQT = BQ / 16;
    constexpr int NCH = BQ * H::ROWCH / 256;
    constexpr bool DMA = (ES == 2) && TAV_ATT_DMA;              // bf16: Q / dO tiles by LDS-DMA (PairDmaT<BQ>); f32: register staging
    constexpr int ROW_B = BQ * H::ROWB, NAT_B = H::DUAL ? 0 : BQ * H::PITCH_N;
    constexpr int BUF_B = 2 * ROW_B + 2 * NAT_B + 2 * BQ * 4;   // Qrow, dOrow, [Qnat, dOnat: f32 only], lse, delta
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + 2 * BUF_B);   // [4][64] + [64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const AttnTile wg = attn_tile(p);
    const int head = wg.head, b = wg.b, S = p.S;
    const int k0 = wg.x * 128 + wave * 32;
    const long hoff = (long)head * 64 * ES;
    const char* Qb = p.q + (long)b * S * p.ld_q * ES + hoff;
    const char* Kb = p.k + (long)b * S * p.ld_k * ES + hoff;
    const char* Vb = p.v + (long)b * S * p.ld_v * ES + hoff;
    const char* dOb = p.dout + (long)b * S * p.ld_do * ES + hoff;
    const float* lseb = p.lse + ((long)b * p.nh + head) * S;
    const float* deltab = p.delta + ((long)b * p.nh + head) * S;

    uint4 kf[2][NSD], vf[2][NSD];
    float kadd[2], cmk[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = k0 + 16 * kt + i;
        row_frags_gload<T>(kf[kt], Kb, p.ld_k * ES, key, S, g);
        row_frags_gload<T>(vf[kt], Vb, p.ld_v * ES, key, S, g);
        const bool ok = key < S;
        const float mv = (MODE != 0 && ok) ? p.mask[(long)b * S + key] : 0.f;
        kadd[kt] = ok ? (MODE == 1 ? mv * 1.4426950408889634f : 0.f) : -INFINITY;    // exp2 domain
        cmk[kt] = (MODE == 2) ? mv : 0.f;
    }
    // PRE: q holds q * scale * log2(e), so S = q~ k^T is the exp2-domain logit: the accumulators start at -lse * log2(e) and p = exp2(acc)
    // without a multiply; dK = sum dS q comes out in units of q~ and is brought back with ln 2 instead of the softmax scale.
    const float c2 = PRE ? 1.0f : p.scale * 1.4426950408889634f, lse_mul = PRE ? 1.4426950408889634f : 1.0f / p.scale;
    const float dk_mul = PRE ? 0.6931471805599453f : p.scale;
    f32x4 dVt[4][2], dKt[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) { dVt[a][c] = f32x4{0.f, 0.f, 0.f, 0.f}; dKt[a][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float dosum_part = 0.f;

    const bool with_kadd = (MODE == 1) || (k0 + 32 > S);      // wave-uniform
    const int nqt = (S + BQ - 1) / BQ;
    uint4 rq[NCH], rdo[NCH];
    float r_lse = 0.f, r_delta = 0.f;
    unsigned q_off0[NCH], q_max[NCH], do_off0[NCH], do_max[NCH];
    if constexpr (!DMA) {
        tile_addr_init<T, BQ>(q_off0, q_max, p.ld_q * ES, S, tid);
        tile_addr_init<T, BQ>(do_off0, do_max, p.ld_do * ES, S, tid);
    }
    const unsigned qstep_b = (unsigned)(BQ * p.ld_q * ES), dostep_b = (unsigned)(BQ * p.ld_do * ES);
    PairDmaT<(ES == 2 ? BQ : 64)> qd;
    if constexpr (DMA) qd.init(Qb, dOb, p.ld_q * ES, p.ld_do * ES, S, smem, smem + ROW_B, tid);
    // lane constants of the fragment reads (bf16 row images; see attn_fwd_kernel): the row reads of Q and dO share one per k-step, the
    // transposed reads one per d-tile
    unsigned roff[NSD], toff[4];
    if constexpr (ES == 2) {
#pragma unroll
        for (int s = 0; s < NSD; ++s) { roff[s] = (unsigned)H::row_off(i, 4 * s + g); asm volatile("" : "+v"(roff[s])); }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            toff[dt] = (unsigned)(H::row_off(4 * g + (i >> 2), 2 * dt + ((i & 3) >> 1)) + 8 * (i & 1));
            asm volatile("" : "+v"(toff[dt]));
        }
    }
    auto gload = [&](int t) {
        if constexpr (!DMA) {
            tile_gload(rq, Qb, q_off0, q_max, t * qstep_b);
            tile_gload(rdo, dOb, do_off0, do_max, t * dostep_b);
        }
        if (tid < BQ) {                                   // raw loads only (see attn_fwd_kernel::gload)
            int q = t * BQ + tid;
            q = q < S ? q : S - 1;
            r_lse = lseb[q];
            r_delta = deltab[q];
        }
    };
    auto lstore = [&](int t, int buf) {
        char* base = smem + buf * BUF_B;
        if constexpr (!DMA) {
            tile_zero_pad<T, BQ>(rq, t * BQ, S, tid);
            tile_zero_pad<T, BQ>(rdo, t * BQ, S, tid);
            tile_lstore_row<T, BQ>(rq, base, tid);
            tile_lstore_row<T, BQ>(rdo, base + ROW_B, tid);
            if constexpr (!H::DUAL) {
                tile_lstore_nat<T, BQ>(rq, base + 2 * ROW_B, tid);
                tile_lstore_nat<T, BQ>(rdo, base + 2 * ROW_B + NAT_B, tid);
            }
        }
        if (tid < BQ) {
            const bool ok = t * BQ + tid < S;
            float* f = reinterpret_cast<float*>(base + 2 * ROW_B + 2 * NAT_B);
            f[tid] = ok ? -r_lse * lse_mul : -INFINITY;     // S accumulators start at -lse/scale (PRE: -lse*log2e); -inf => p = 0 for rows past S
            f[BQ + tid] = ok ? -r_delta : 0.f;                // dP accumulators start at -delta
        }
    };
    if constexpr (DMA) { if (nqt == 1) qd.template issue<true>(0, 0u); else qd.template issue<false>(0, 0u); }
    gload(0); lstore(0, 0);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int s = 0; s < NSD; ++s) { settle(kf[kt][s]); settle(vf[kt][s]); }
        settle(kadd[kt]); settle(cmk[kt]);
    }
    if constexpr (DMA) wait_vmcnt0();
    __syncthreads();

    // NEXT: 0 no prefetch, 1 regular, 2 the ragged last tile (DMA rows past S-1 repeat row S-1: finite, and their p is 0 through lse = -inf)
    auto tile_body = [&](const int t, auto next_tag) __attribute__((always_inline)) {
        constexpr int NEXT = decltype(next_tag)::value;
        const int cur = t & 1;
        if constexpr (NEXT != 0) {
            if constexpr (DMA) qd.template issue<NEXT == 2>(t + 1, (unsigned)((cur ^ 1) * BUF_B));
            gload(t + 1);
        }
        const char* Qrow = smem + cur * BUF_B;
        const char* dOrow = Qrow + ROW_B;
        const char* Qnat = Qrow + 2 * ROW_B;
        const char* dOnat = Qnat + NAT_B;
        const float* lse_s = reinterpret_cast<const float*>(dOnat + NAT_B);
        const float* delta_s = lse_s + BQ;
        const int qbase = t * BQ;

        // Row constants as the initial accumulators: S' = q.k - lse/scale and dP' = dO.v - delta leave the MFMA chains ready, so
        // p = exp2(c2 * S') and dS = p * dP' need no subtraction per element (the per-query constants sit on the register axis).
        f32x4 sacc[NQT][2], dpacc[NQT][2];   // [qt][kt]: rows(regs) = query, cols(lanes) = key
#pragma unroll
        for (int a = 0; a < NQT; ++a) {
            const f32x4 L = *reinterpret_cast<const f32x4*>(lse_s + 16 * a + 4 * g);
            const f32x4 D = *reinterpret_cast<const f32x4*>(delta_s + 16 * a + 4 * g);
#pragma unroll
            for (int c = 0; c < 2; ++c) { sacc[a][c] = L; dpacc[a][c] = D; }
        }
        constexpr int NKS = BQ / KSTEP;
#pragma unroll
        for (int s = 0; s < NSD; ++s)
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                uint4 aq, ad;
                if constexpr (ES == 2) {
                    aq = *reinterpret_cast<const uint4*>(Qrow + roff[s] + qt * 16 * H::ROWB);
                    ad = *reinterpret_cast<const uint4*>(dOrow + roff[s] + qt * 16 * H::ROWB);
                } else {
                    aq = *reinterpret_cast<const uint4*>(Qrow + H::row_off(16 * qt + i, 4 * s + g));
                    ad = *reinterpret_cast<const uint4*>(dOrow + H::row_off(16 * qt + i, 4 * s + g));
                }
                mma16<T>(aq, kf[0][s], sacc[qt][0]);
                mma16<T>(aq, kf[1][s], sacc[qt][1]);
                mma16<T>(ad, vf[0][s], dpacc[qt][0]);
                mma16<T>(ad, vf[1][s], dpacc[qt][1]);
            }
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            // p = exp2(c2 * S' + kadd);  dS (without the softmax scale, applied once to dK at the end) = p * dP'.
            // kadd is zero except under a pre-softmax mask (MODE 1) and for keys past S (-inf): wave-uniform choice.
            if (with_kadd) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    // (without a pre-softmax mask the term is 0 / -inf past S: re-derived here, on the ragged key block only, instead of
                    // living in two registers across the whole loop -- the 32-query form of this kernel sits at the 168-register line)
                    float ka = kadd[kt];
                    if constexpr (MODE != 1) { int ky = k0 + 16 * kt + i; asm volatile("" : "+v"(ky)); ka = ky < S ? 0.f : -INFINITY; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[qt][kt][r] = fast_exp2(PRE ? sacc[qt][kt][r] + ka : __builtin_fmaf(sacc[qt][kt][r], c2, ka));
                }
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[qt][kt][r] = fast_exp2(PRE ? sacc[qt][kt][r] : sacc[qt][kt][r] * c2);
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dpacc[qt][kt][r] *= sacc[qt][kt][r];
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            uint4 pb[2], dsb[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x4 tl[2];
                tl[0] = sacc[ks * ET<T>::ACC_TILES][kt]; tl[1] = sacc[ks * ET<T>::ACC_TILES + ET<T>::ACC_TILES - 1][kt];
                pb[kt] = acc_to_kfrag<T>(tl);
                tl[0] = dpacc[ks * ET<T>::ACC_TILES][kt]; tl[1] = dpacc[ks * ET<T>::ACC_TILES + ET<T>::ACC_TILES - 1][kt];
                dsb[kt] = acc_to_kfrag<T>(tl);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint4 a1, a2;
                if constexpr (ES == 2) {                      // = frag_tr_rowimg(.., ks * KSTEP, dt, lane) with the lane part precomputed
                    const char* t1 = dOrow + toff[dt] + ks * KSTEP * H::ROWB;
                    const char* t2 = Qrow + toff[dt] + ks * KSTEP * H::ROWB;
                    const uint2 lo1 = lds_read_tr16(t1), hi1 = lds_read_tr16(t1 + 16 * H::ROWB);
                    const uint2 lo2 = lds_read_tr16(t2), hi2 = lds_read_tr16(t2 + 16 * H::ROWB);
                    a1 = make_uint4(lo1.x, lo1.y, hi1.x, hi1.y);
                    a2 = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
                } else {
                    a1 = tile_kfrag<T>(dOrow, dOnat, ks * KSTEP, dt, lane);
                    a2 = tile_kfrag<T>(Qrow, Qnat, ks * KSTEP, dt, lane);
                }
                mma16<T>(a1, pb[0], dVt[dt][0]);
                mma16<T>(a1, pb[1], dVt[dt][1]);
                mma16<T>(a2, dsb[0], dKt[dt][0]);
                mma16<T>(a2, dsb[1], dKt[dt][1]);
            }
        }
        if (MODE == 2) {   // sum_q dO[q][d]  (register staging zero-fills rows past S; the DMA repeats row S-1 there: skip them)
            const int d = tid & 63, rq8 = tid >> 6;
#pragma unroll
            for (int rr = 0; rr < BQ / 4; ++rr) {
                const int row = rq8 * (BQ / 4) + rr;
                const float v = tile_elem<T>(dOrow, dOnat, row, d);
                dosum_part += (!DMA || qbase + row < S) ? v : 0.f;
            }
        }
        if constexpr (NEXT != 0) lstore(t + 1, cur ^ 1);
        if constexpr (DMA) wait_vmcnt0();                     // tile t+1 has landed
        __syncthreads();
    };
    int t = 0;
    for (; t + 2 < nqt; ++t) tile_body(t, std::integral_constant<int, 1>{});
    if (t + 1 < nqt) { tile_body(t, std::integral_constant<int, 2>{}); ++t; }
    tile_body(t, std::integral_constant<int, 0>{});

    if (MODE == 2) {
        red[wave * 64 + lane] = dosum_part;
        __syncthreads();
        if (tid < 64) red[256 + tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
        __syncthreads();
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = k0 + 16 * kt + i;
        if (key < S) {
            T* dkrow = reinterpret_cast<T*>(p.dk) + ((long)b * S + key) * p.ld_dk + head * 64;
            T* dvrow = reinterpret_cast<T*>(p.dv) + ((long)b * S + key) * p.ld_dv + head * 64;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 dvv = dVt[dt][kt];
                if (MODE == 2) dvv += *reinterpret_cast<const f32x4*>(red + 256 + 16 * dt + 4 * g) * cmk[kt];
                st4(dvrow + 16 * dt + 4 * g, dvv);
                st4(dkrow + 16 * dt + 4 * g, dKt[dt][kt] * dk_mul);
            }
        }
    }
}

// ================================================================================================= backward: dQ
template <typename T, int MODE, bool PRE>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? TAV_ATT_DQ_OCC : 1)) void attn_bwd_dq_kernel(const AttnP p) {
    using H = HD<T>;
    constexpr int ES = H::ES, NSD = H::NSD, KSTEP = ET<T>::KSTEP, BKV = 64;
    constexpr int NCH = BKV * H::ROWCH / 256;
    constexpr bool DMA = (ES == 2) && TAV_ATT_DMA;          // bf16: K / V tiles by LDS-DMA (PairDma), no staging registers; f32: register staging
    constexpr int ROW_B = BKV * H::ROWB, NAT_B = H::DUAL ? 0 : BKV * H::PITCH_N;
    constexpr int BUF_B = 2 * ROW_B + NAT_B + BKV * 4;   // Krow, Vrow, [Knat: f32 only], kadd
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const AttnTile wg = attn_tile(p);
    const int head = wg.head, b = wg.b, S = p.S;
    const int q0 = wg.x * 128 + wave * 32;
    const long hoff = (long)head * 64 * ES;
    const char* Qb = p.q + (long)b * S * p.ld_q * ES + hoff;
    const char* Kb = p.k + (long)b * S * p.ld_k * ES + hoff;
    const char* Vb = p.v + (long)b * S * p.ld_v * ES + hoff;
    const char* dOb = p.dout + (long)b * S * p.ld_do * ES + hoff;
    const float* maskb = p.mask ? p.mask + (long)b * S : nullptr;

    uint4 qf[2][NSD], dof[2][NSD];
    float lse_q[2], delta_q[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int q = q0 + 16 * qt + i;
        row_frags_gload<T>(qf[qt], Qb, p.ld_q * ES, q, S, g);
        row_frags_gload<T>(dof[qt], dOb, p.ld_do * ES, q, S, g);
        if (q >= S) q = S - 1;
        lse_q[qt] = -p.lse[((long)b * p.nh + head) * S + q] * (PRE ? 1.4426950408889634f : 1.0f / p.scale);   // S accumulators start at -lse/scale (PRE: -lse*log2e), dP at -delta
        // delta[q] = sum_d dO[q][d] * (softmax(s) v)[q][d], formed HERE (this kernel runs first and has the dO rows in registers anyway; the
        // separate delta kernel of rounds 1-2 cost a launch and 33 us at batch 32) and written out for the dK/dV kernel.  Lane (g, i) holds
        // chunks 4s + g of query i's rows: 16 of the 64 products; the four lane groups meet in two permlane swaps.
        const char* Ob = (MODE == 2 ? p.o_soft : p.o) + ((long)b * S * p.ld_o + (long)head * 64) * ES;     // (mode 2: the softmax-only part, never o - corr)
        uint4 of[NSD];
        row_frags_gload<T>(of, Ob, p.ld_o * ES, q, S, g);
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < NSD; ++s) {
            const uint32_t ow[4] = {of[s].x, of[s].y, of[s].z, of[s].w}, dw[4] = {dof[qt][s].x, dof[qt][s].y, dof[qt][s].z, dof[qt][s].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (ES == 2) {
                    part = __builtin_fmaf(bf16_bits_to_f32(ow[e] & 0xffffu), bf16_bits_to_f32(dw[e] & 0xffffu), part);
                    part = __builtin_fmaf(__uint_as_float(ow[e] & 0xffff0000u), __uint_as_float(dw[e] & 0xffff0000u), part);
                } else part = __builtin_fmaf(__uint_as_float(ow[e]), __uint_as_float(dw[e]), part);
            }
        }
        part = sum_over_row_groups(part);
        delta_q[qt] = -part;                                               // (row constants as the initial accumulators)
        if (g == 0 && q0 + 16 * qt + i < S) p.delta[((long)b * p.nh + head) * S + q] = part;
    }
    const float c2 = PRE ? 1.0f : p.scale * 1.4426950408889634f;
    f32x4 dQt[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) { dQt[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; dQt[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nkt = (S + BKV - 1) / BKV;
    uint4 rk[NCH], rv[NCH];
    float r_kadd = 0.f;
    unsigned k_off0[NCH], k_max[NCH], v_off0[NCH], v_max[NCH];
    if constexpr (!DMA) {
        tile_addr_init<T, BKV>(k_off0, k_max, p.ld_k * ES, S, tid);
        tile_addr_init<T, BKV>(v_off0, v_max, p.ld_v * ES, S, tid);
    }
    const unsigned kstep_b = (unsigned)(BKV * p.ld_k * ES), vstep_b = (unsigned)(BKV * p.ld_v * ES);
    PairDma kv;
    if constexpr (DMA) kv.init(Kb, Vb, p.ld_k * ES, p.ld_v * ES, S, smem, smem + ROW_B, tid);
    // lane constants of the fragment reads (bf16 row images; see attn_fwd_kernel): row reads of K and V share one, the transposed K reads four
    unsigned koff[NSD], ktoff[4];
    if constexpr (ES == 2) {
#pragma unroll
        for (int s = 0; s < NSD; ++s) { koff[s] = (unsigned)H::row_off(i, 4 * s + g); asm volatile("" : "+v"(koff[s])); }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            ktoff[dt] = (unsigned)(H::row_off(4 * g + (i >> 2), 2 * dt + ((i & 3) >> 1)) + 8 * (i & 1));
            asm volatile("" : "+v"(ktoff[dt]));
        }
    }
    auto gload = [&](int t) {
        if constexpr (!DMA) {
            tile_gload(rk, Kb, k_off0, k_max, t * kstep_b);
            tile_gload(rv, Vb, v_off0, v_max, t * vstep_b);
        }
        if (MODE == 1 && tid < BKV) {                     // raw load only (see attn_fwd_kernel::gload)
            int key = t * BKV + tid;
            key = key < S ? key : S - 1;
            r_kadd = maskb[key];
        }
    };
    auto lstore = [&](int t, int buf) {
        char* base = smem + buf * BUF_B;
        if constexpr (!DMA) {
            tile_lstore_row<T, BKV>(rk, base, tid);
            tile_lstore_row<T, BKV>(rv, base + ROW_B, tid);
            if constexpr (!H::DUAL) tile_lstore_nat<T, BKV>(rk, base + 2 * ROW_B, tid);
        }
        if ((MODE == 1 || t == nkt - 1) && tid < BKV)     // (only the masked / ragged tiles read the per-key terms)
            reinterpret_cast<float*>(base + 2 * ROW_B + NAT_B)[tid] = (t * BKV + tid < S) ? (MODE == 1 ? r_kadd * 1.4426950408889634f : 0.f) : -INFINITY;
    };
    if constexpr (DMA) { if (nkt == 1) kv.template issue<true>(0, 0u); else kv.template issue<false>(0, 0u); }
    gload(0); lstore(0, 0);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int s = 0; s < NSD; ++s) { settle(qf[qt][s]); settle(dof[qt][s]); }
        settle(lse_q[qt]); settle(delta_q[qt]);
    }
    if constexpr (DMA) wait_vmcnt0();
    __syncthreads();

    // KADD: the tile has per-key additive terms (pre-softmax mask; -inf past S on the last tile).  NEXT: 0 none, 1 regular, 2 ragged prefetch.
    auto tile_body = [&](const int t, auto kadd_tag, auto next_tag) __attribute__((always_inline)) {
        constexpr bool KADD = decltype(kadd_tag)::value != 0;
        constexpr int NEXT = decltype(next_tag)::value;
        const int cur = t & 1;
        if constexpr (NEXT != 0) {
            if constexpr (DMA) kv.template issue<NEXT == 2>(t + 1, (unsigned)((cur ^ 1) * BUF_B));
            gload(t + 1);
        }
        const char* Krow = smem + cur * BUF_B;
        const char* Vrow = Krow + ROW_B;
        const char* Knat = Krow + 2 * ROW_B;
        const float* kadd = reinterpret_cast<const float*>(Knat + NAT_B);

        f32x4 sacc[4][2], dpacc[4][2];   // [kt][qt]: rows(regs) = key, cols(lanes) = query
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                sacc[kt][c] = f32x4{lse_q[c], lse_q[c], lse_q[c], lse_q[c]};
                dpacc[kt][c] = f32x4{delta_q[c], delta_q[c], delta_q[c], delta_q[c]};
            }
        constexpr int NKS = BKV / KSTEP;
#pragma unroll
        for (int s = 0; s < NSD; ++s)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                uint4 ak, av;
                if constexpr (ES == 2) {
                    ak = *reinterpret_cast<const uint4*>(Krow + koff[s] + kt * 16 * H::ROWB);
                    av = *reinterpret_cast<const uint4*>(Vrow + koff[s] + kt * 16 * H::ROWB);
                } else {
                    ak = *reinterpret_cast<const uint4*>(Krow + H::row_off(16 * kt + i, 4 * s + g));
                    av = *reinterpret_cast<const uint4*>(Vrow + H::row_off(16 * kt + i, 4 * s + g));
                }
                mma16<T>(ak, qf[0][s], sacc[kt][0]);
                mma16<T>(ak, qf[1][s], sacc[kt][1]);
                mma16<T>(av, dof[0][s], dpacc[kt][0]);
                mma16<T>(av, dof[1][s], dpacc[kt][1]);
            }
        // p = exp2(c2 * S' + kadd);  dS^T (without the softmax scale, applied at the store) = p * dP'.  kadd is zero except under a
        // pre-softmax mask (MODE 1) and on the ragged last tile (-inf past S).
        if constexpr (KADD) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 ka = *reinterpret_cast<const f32x4*>(kadd + 16 * kt + 4 * g);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[kt][qt][r] = fast_exp2(PRE ? sacc[kt][qt][r] + ka[r] : __builtin_fmaf(sacc[kt][qt][r], c2, ka[r])) * dpacc[kt][qt][r];
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[kt][qt][r] = fast_exp2(PRE ? sacc[kt][qt][r] : sacc[kt][qt][r] * c2) * dpacc[kt][qt][r];
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            uint4 dsb[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 tl[2];
                tl[0] = sacc[ks * ET<T>::ACC_TILES][qt]; tl[1] = sacc[ks * ET<T>::ACC_TILES + ET<T>::ACC_TILES - 1][qt];
                dsb[qt] = acc_to_kfrag<T>(tl);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint4 a;
                if constexpr (ES == 2) {                      // = frag_tr_rowimg(Krow, ks * KSTEP, dt, lane) with the lane part precomputed
                    const char* kt_ = Krow + ktoff[dt] + ks * KSTEP * H::ROWB;
                    const uint2 lo = lds_read_tr16(kt_), hi = lds_read_tr16(kt_ + 16 * H::ROWB);
                    a = make_uint4(lo.x, lo.y, hi.x, hi.y);
                } else a = tile_kfrag<T>(Krow, Knat, ks * KSTEP, dt, lane);
                mma16<T>(a, dsb[0], dQt[dt][0]);
                mma16<T>(a, dsb[1], dQt[dt][1]);
            }
        }
        if constexpr (NEXT != 0) lstore(t + 1, cur ^ 1);
        if constexpr (DMA) wait_vmcnt0();                     // tile t+1 has landed
        __syncthreads();
    };
    using K0 = std::integral_constant<int, (MODE == 1) ? 1 : 0>;
    int t = 0;
    for (; t + 2 < nkt; ++t) tile_body(t, K0{}, std::integral_constant<int, 1>{});
    if (t + 1 < nkt) { tile_body(t, K0{}, std::integral_constant<int, 2>{}); ++t; }
    tile_body(t, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});          // the last tile: -inf past S
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + 16 * qt + i;
        if (q < S) {
            T* dqrow = reinterpret_cast<T*>(p.dq) + ((long)b * S + q) * p.ld_dq + head * 64;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) st4(dqrow + 16 * dt + 4 * g, dQt[dt][qt] * p.scale);
        }
    }
}

template <typename T> constexpr size_t fwd_lds() {
    return 2 * (64 * HD<T>::ROWB + 64 * HD<T>::PITCH_N + 2 * 64 * 4) + (256 + 64) * 4;
}
template <typename T, int MODE, bool PRE> constexpr size_t dkdv_lds() {
    constexpr int BQ = dkdv_bq<T, MODE, PRE>();
    return 2 * (2 * BQ * HD<T>::ROWB + (HD<T>::DUAL ? 0 : 2 * BQ * HD<T>::PITCH_N) + 2 * BQ * 4) + (256 + 64) * 4;
}
template <typename T> constexpr size_t dq_lds() { return 2 * (2 * 64 * HD<T>::ROWB + (HD<T>::DUAL ? 0 : 64 * HD<T>::PITCH_N) + 64 * 4); }

// ------------------------------------------------------------------------------------------------
// Slow-path helpers for two options of the fusion encoder that its training loop never uses (reference utils/TAVFormer.py:190, :368-370,
// :389): the attention probabilities as a tensor (output_attentions) and a per-head factor on the softmax part of the context (head_mask).
// Plain VALU kernels; the flash kernels above stay the product path and supply lse.
//   probs[b][h][i][j] = hs[b][h] * exp(scale * q_i.k_j (+ mask_j, mode 1) - lse_i) (+ mask_j, mode 2)
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_kernel(const AttnP p, float* __restrict__ probs, const float* __restrict__ hs, int hs_bstride, int mode) {
    constexpr int ES = ET<T>::ES;
    constexpr float L2E = 1.4426950408889634f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;             // (b, head, i)
    if (row >= (long)p.B * p.nh * p.S) return;
    const int i = (int)(row % p.S), head = (int)((row / p.S) % p.nh), b = (int)(row / ((long)p.S * p.nh));
    const T* qr = reinterpret_cast<const T*>(p.q + ((long)(b * p.S + i) * p.ld_q + head * 64) * ES);
    f32x4 qv[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) qv[t] = ld4(qr + 4 * t);     // the same address in every lane: one broadcast fetch
    const float c = p.pre ? 1.f : p.scale * L2E;
    const float l2 = p.lse[((long)b * p.nh + head) * p.S + i] * L2E;
    const float f = hs ? hs[(long)b * hs_bstride + head] : 1.f;
    for (int j = lane; j < p.S; j += 64) {
        const T* kr = reinterpret_cast<const T*>(p.k + ((long)(b * p.S + j) * p.ld_k + head * 64) * ES);
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) { const f32x4 kv = ld4(kr + 4 * t); dot += qv[t][0] * kv[0] + qv[t][1] * kv[1] + qv[t][2] * kv[2] + qv[t][3] * kv[3]; }
        float s2 = dot * c;
        if (mode == 1) s2 += p.mask[(long)b * p.S + j] * L2E;
        float pr = exp2f(s2 - l2) * f;
        if (mode == 2) pr += p.mask[(long)b * p.S + j];
        probs[row * p.S + j] = pr;
    }
}
//   out[r][h*64 + d] = (a ? a[r][...] : 0) + (c0 + (hs ? hs[b][h] : 0)) * b[r][...]        r = b*S + s
template <typename T>
__global__ void head_scale_kernel(const T* __restrict__ a, const T* __restrict__ bsrc, T* __restrict__ out, const float* __restrict__ hs, int hs_bstride,
                                  float c0, long n4, int S, int nh, long lda, long ldb, long ldo) {
    const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= n4) return;
    const long row = i4 / (nh * 16);
    const int col4 = (int)(i4 - row * (nh * 16)), head = col4 >> 4;
    const float f = c0 + (hs ? hs[(row / S) * hs_bstride + head] : 0.f);
    f32x4 v = ld4(bsrc + row * ldb + 4 * col4) * f;
    if (a) v += ld4(a + row * lda + 4 * col4);
    st4(out + row * ldo + 4 * col4, v);
}

static int check(const tav_attn_args* a, bool bwd) {
    if (!a || !a->q || !a->k || !a->v || !a->o || !a->lse) return TAV_ERR_NULL;
    if (a->B <= 0 || a->S <= 0 || a->nheads <= 0) return TAV_ERR_SHAPE;
    if (((a->S + 63) / 64) * a->nheads * a->B >= (1ll << 31)) return TAV_ERR_SHAPE;      // the grid is one-dimensional (attn_tile)
    if (a->dtype != TAV_BF16 && a->dtype != TAV_F32) return TAV_ERR_DTYPE;
    if (a->mask_mode < 0 || a->mask_mode > 2) return TAV_ERR_SHAPE;
    if (a->mask_mode != 0 && !a->key_mask) return TAV_ERR_NULL;
    if (a->mask_mode == 2 && (!a->corr || !a->o_soft)) return TAV_ERR_NULL;
    const int pk = a->dtype == TAV_BF16 ? 8 : 4;
    if (a->ld_q % pk || a->ld_k % pk || a->ld_v % pk || a->ld_o % 4) return TAV_ERR_ALIGN;
    if (a->ld_q < a->nheads * 64 || a->ld_k < a->nheads * 64 || a->ld_v < a->nheads * 64 || a->ld_o < a->nheads * 64) return TAV_ERR_SHAPE;
    {   // the streamed tiles are addressed with 32-bit byte offsets inside one batch entry's slice
        const int64_t es = a->dtype == TAV_BF16 ? 2 : 4;
        int64_t ld = a->ld_q > a->ld_k ? a->ld_q : a->ld_k;
        ld = ld > a->ld_v ? ld : a->ld_v;
        if (bwd && a->ld_do > ld) ld = a->ld_do;
        if ((a->S + 64) * ld * es >= (1ll << 32)) return TAV_ERR_SHAPE;
    }
    if (bwd) {
        if (!a->dout || !a->dq || !a->dk || !a->dv || !a->delta) return TAV_ERR_NULL;
        if (a->ld_do % pk || a->ld_dq % 4 || a->ld_dk % 4 || a->ld_dv % 4) return TAV_ERR_ALIGN;
    }
    return 0;
}
static AttnP pack(const tav_attn_args* a) {
    AttnP p;
    p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v; p.o = (char*)a->o; p.o_soft = (char*)a->o_soft;
    p.mask = a->key_mask; p.lse = a->lse; p.corr = a->corr;
    p.dout = (const char*)a->dout; p.dq = (char*)a->dq; p.dk = (char*)a->dk; p.dv = (char*)a->dv; p.delta = a->delta;
    p.B = (int)a->B; p.S = (int)a->S; p.nh = (int)a->nheads;
    p.ld_q = a->ld_q; p.ld_k = a->ld_k; p.ld_v = a->ld_v; p.ld_o = a->ld_o; p.ld_do = a->ld_do;
    p.ld_dq = a->ld_dq; p.ld_dk = a->ld_dk; p.ld_dv = a->ld_dv;
    p.scale = a->scale;
    p.pre = a->q_prescaled != 0;
    p.tiles = 1;
    return p;
}

template <typename T, int MODE, bool PRE> static int launch_fwd(const AttnP& p, hipStream_t st) {
#if TAV_ATT_FWD32
    if constexpr (sizeof(T) == 2 && MODE == 0 && PRE) {
        AttnP pl = p; pl.tiles = (p.S + 127) / 128;
        hipLaunchKernelGGL(attn_fwd32_kernel, dim3((unsigned)pl.tiles * p.nh * p.B), dim3(256), fwd_lds<bf16>(), st, pl);
        return (int)hipGetLastError();
    }
#endif
    constexpr int QW = 64 * fwd_nq<T, MODE, PRE>();          // queries per workgroup
    AttnP pl = p; pl.tiles = (p.S + QW - 1) / QW;
    hipLaunchKernelGGL((attn_fwd_kernel<T, MODE, PRE>), dim3((unsigned)pl.tiles * p.nh * p.B), dim3(256), fwd_lds<T>(), st, pl);
    return (int)hipGetLastError();
}
template <typename T, int MODE, bool PRE> static int launch_bwd(const AttnP& p, hipStream_t st) {
    AttnP pl = p; pl.tiles = (p.S + 127) / 128;
    const dim3 grid((unsigned)pl.tiles * p.nh * p.B);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, MODE, PRE>), grid, dim3(256), dq_lds<T>(), st, pl);        // also writes delta [B][nh][S]
    constexpr size_t lds_dkdv = dkdv_lds<T, MODE, PRE>();
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, MODE, PRE>), grid, dim3(256), lds_dkdv, st, pl);    // reads it
    return (int)hipGetLastError();
}
template <typename T, bool PRE> static int dispatch_fwd(const AttnP& p, int mode, hipStream_t st) {
    switch (mode) { case 0: return launch_fwd<T, 0, PRE>(p, st); case 1: return launch_fwd<T, 1, PRE>(p, st); default: return launch_fwd<T, 2, PRE>(p, st); }
}
template <typename T, bool PRE> static int dispatch_bwd(const AttnP& p, int mode, hipStream_t st) {
    switch (mode) { case 0: return launch_bwd<T, 0, PRE>(p, st); case 1: return launch_bwd<T, 1, PRE>(p, st); default: return launch_bwd<T, 2, PRE>(p, st); }
}

}  // namespace tav

using namespace tav;

extern "C" int tav_attn_fwd(const tav_attn_args* a, void* stream) {
    int e = check(a, false);
    if (e) return e;
    const AttnP p = pack(a);
    hipStream_t st = (hipStream_t)stream;
    if (a->dtype == TAV_BF16) return p.pre ? dispatch_fwd<bf16, true>(p, a->mask_mode, st) : dispatch_fwd<bf16, false>(p, a->mask_mode, st);
    return p.pre ? dispatch_fwd<float, true>(p, a->mask_mode, st) : dispatch_fwd<float, false>(p, a->mask_mode, st);
}

extern "C" int tav_attn_bwd(const tav_attn_args* a, void* stream) {
    int e = check(a, true);
    if (e) return e;
    const AttnP p = pack(a);
    hipStream_t st = (hipStream_t)stream;
    if (a->dtype == TAV_BF16) return p.pre ? dispatch_bwd<bf16, true>(p, a->mask_mode, st) : dispatch_bwd<bf16, false>(p, a->mask_mode, st);
    return p.pre ? dispatch_bwd<float, true>(p, a->mask_mode, st) : dispatch_bwd<float, false>(p, a->mask_mode, st);
}

extern "C" int tav_attn_probs(const tav_attn_args* a, float* probs, const float* head_scale, int64_t hs_bstride, void* stream) {
    if (!a || !a->q || !a->k || !a->lse || !probs) return TAV_ERR_NULL;
    if (a->B <= 0 || a->S <= 0 || a->nheads <= 0 || hs_bstride < 0) return TAV_ERR_SHAPE;
    if (a->dtype != TAV_BF16 && a->dtype != TAV_F32) return TAV_ERR_DTYPE;
    if (a->mask_mode < 0 || a->mask_mode > 2) return TAV_ERR_SHAPE;
    if (a->mask_mode != 0 && !a->key_mask) return TAV_ERR_NULL;
    const int pk = a->dtype == TAV_BF16 ? 8 : 4;
    if (a->ld_q % pk || a->ld_k % pk) return TAV_ERR_ALIGN;
    const AttnP p = pack(a);
    const long rows = (long)p.B * p.nh * p.S;
    hipStream_t st = (hipStream_t)stream;
    if (a->dtype == TAV_BF16) hipLaunchKernelGGL((attn_probs_kernel<bf16>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p, probs, head_scale, (int)hs_bstride, (int)a->mask_mode);
    else hipLaunchKernelGGL((attn_probs_kernel<float>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p, probs, head_scale, (int)hs_bstride, (int)a->mask_mode);
    return (int)hipGetLastError();
}

extern "C" int tav_head_scale(const void* a, const void* b, void* out, int32_t dtype, const float* head_scale, int64_t hs_bstride, float c0, int64_t B,
                              int64_t S, int64_t nheads, int64_t lda, int64_t ldb, int64_t ldo, void* stream) {
    if (!b || !out) return TAV_ERR_NULL;
    if (B <= 0 || S <= 0 || nheads <= 0 || hs_bstride < 0) return TAV_ERR_SHAPE;
    if (dtype != TAV_BF16 && dtype != TAV_F32) return TAV_ERR_DTYPE;
    if (lda % 4 || ldb % 4 || ldo % 4) return TAV_ERR_ALIGN;
    const long n4 = B * S * nheads * 16;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
    if (dtype == TAV_BF16) hipLaunchKernelGGL((head_scale_kernel<bf16>), grid, block, 0, st, (const bf16*)a, (const bf16*)b, (bf16*)out, head_scale, (int)hs_bstride, c0, n4, (int)S, (int)nheads, (long)lda, (long)ldb, (long)ldo);
    else hipLaunchKernelGGL((head_scale_kernel<float>), grid, block, 0, st, (const float*)a, (const float*)b, (float*)out, head_scale, (int)hs_bstride, c0, n4, (int)S, (int)nheads, (long)lda, (long)ldb, (long)ldo);
    return (int)hipGetLastError();
}
