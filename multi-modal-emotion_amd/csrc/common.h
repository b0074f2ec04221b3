// Device-side building blocks shared by every kernel of libtavhip (gfx950 / CDNA4 only).
//
// Conventions used throughout csrc/:
//   * wave = 64 lanes; lane = threadIdx.x & 63; for MFMA work lane = (g, i) with i = lane & 15
//     (the "row/col within a 16x16 tile" index) and g = lane >> 4 (the k-group, 0..3).
//   * one "chunk" = 16 bytes = PK elements (8 bf16 or 4 f32).  Every matrix product in the library
//     is built from mma16<T>(): a 16x16 output tile accumulated over KSTEP = 4*PK values of k
//     (one v_mfma_f32_16x16x32_bf16, or four v_mfma_f32_16x16x4_f32 for exact-f32 parity runs).
//   * operand k-sets.  For one mma16 step starting at k0, lane group g supplies
//         row-chunk operand  (k contiguous in memory):  k = k0 + PK*g + j            j < PK
//         k-strided operand  (k is the slow index)   :  bf16: k = k0 + 4g + j (j<4), k0 + 16 + 4g + (j-4) (j>=4)
//                                                       f32 : k = k0 + 4g + j (j<4)
//     The second map is exactly the order in which a lane already holds a 16x16 accumulator tile
//     (rows 4g..4g+3), so an accumulator can be fed back as an operand with no lane movement;
//     two operands of one product must use the same map.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tav {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

struct bf16 { uint16_t x; };   // storage tag for bfloat16 tensors
struct fp8 { uint8_t x; };     // storage tag for OCP e4m3 tensors (GEMM operands of the fp8 path only)

#define TAV_DEV __device__ __forceinline__

TAV_DEV float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
TAV_DEV uint32_t f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return (uint32_t)__builtin_bit_cast(uint16_t, b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// one v_cvt_pk_bf16_f32 for the pair (converting the halves separately costs 2 cvt + shift + or)
TAV_DEV uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct ET;
template <> struct ET<float> {
    static constexpr int PK = 4, KSTEP = 16, ES = 4, ACC_TILES = 1;
    static TAV_DEV float ld(const float* p) { return *p; }
    static TAV_DEV void st(float* p, float v) { *p = v; }
};
template <> struct ET<bf16> {
    static constexpr int PK = 8, KSTEP = 32, ES = 2, ACC_TILES = 2;
    static TAV_DEV float ld(const bf16* p) { return bf16_bits_to_f32(p->x); }
    static TAV_DEV void st(bf16* p, float v) { p->x = (uint16_t)f32_to_bf16_bits(v); }
};

template <> struct ET<fp8> {
    static constexpr int PK = 16, KSTEP = 128, ES = 1, ACC_TILES = 4;
};

// ---- 4-wide vector load / store of T as floats (addresses 4-element aligned) -------------------
TAV_DEV f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
TAV_DEV f32x4 ld4(const bf16* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    f32x4 r = {bf16_bits_to_f32(u.x & 0xffffu), bf16_bits_to_f32(u.x >> 16), bf16_bits_to_f32(u.y & 0xffffu),
               bf16_bits_to_f32(u.y >> 16)};
    return r;
}
TAV_DEV void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
TAV_DEV void st4(bf16* p, f32x4 v) {
    uint2 u;
    u.x = pack_bf16x2(v[0], v[1]);
    u.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
}

// ---- one 16x16 tile step ------------------------------------------------------------------------
// c[r] (r=0..3) is C[row 4g+r][col i]; "a" supplies C's row index, "b" its column index.
template <typename T> TAV_DEV void mma16(const uint4& a, const uint4& b, f32x4& c);
template <> TAV_DEV void mma16<bf16>(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                                0, 0);
}
template <> TAV_DEV void mma16<float>(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
}

// fp8 (e4m3 x e4m3): one block-scaled MFMA over 128 k-values with every block scale = 2^0 (E8M0 byte 0x7F).  The scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 issues at twice the FLOP rate of the bf16 form (the plain ..._fp8_fp8 form only matches bf16), and
// with unit scales it is an ordinary fp8 product; the per-tensor scales are applied once, in the GEMM epilogue.  A lane supplies 32
// bytes per operand: (lo, hi) = the two 16-byte chunks it read.  Which k-values those are does not matter as long as both operands
// use the same chunks, which they do (same swizzled LDS image, same chunk indices).
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
TAV_DEV void mma16_fp8(const uint4& a_lo, const uint4& a_hi, const uint4& b_lo, const uint4& b_hi, f32x4& c) {
    const i32x8_t a = {(int)a_lo.x, (int)a_lo.y, (int)a_lo.z, (int)a_lo.w, (int)a_hi.x, (int)a_hi.y, (int)a_hi.z, (int)a_hi.w};
    const i32x8_t b = {(int)b_lo.x, (int)b_lo.y, (int)b_lo.z, (int)b_lo.w, (int)b_hi.x, (int)b_hi.y, (int)b_hi.z, (int)b_hi.w};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// ---- LDS helpers --------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

TAV_DEV uint2 lds_read_tr16(const char* p) {
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p));
    return __builtin_bit_cast(uint2, v);
}

// k-strided operand fragment from a "natural" LDS tile: tile[krow][col] with `pitch` bytes per row.
// Returns, for lane (g, i), the PK/… values {tile[k][col0 + i]} over this lane group's k-set (see header).
// EXEC must be all ones (the bf16 form is a cross-lane gather); callers pad instead of masking.
template <typename T> TAV_DEV uint4 frag_kstrided(const char* tile, int pitch, int krow0, int col0, int lane);
template <> TAV_DEV uint4 frag_kstrided<bf16>(const char* tile, int pitch, int krow0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const char* a0 = tile + (krow0 + 4 * g + q) * pitch + (col0 + 4 * p) * 2;
    uint2 lo = lds_read_tr16(a0);
    uint2 hi = lds_read_tr16(a0 + 16 * pitch);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
template <> TAV_DEV uint4 frag_kstrided<float>(const char* tile, int pitch, int krow0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const char* a0 = tile + (krow0 + 4 * g) * pitch + (col0 + i) * 4;
    uint4 r;
    r.x = *reinterpret_cast<const uint32_t*>(a0);
    r.y = *reinterpret_cast<const uint32_t*>(a0 + pitch);
    r.z = *reinterpret_cast<const uint32_t*>(a0 + 2 * pitch);
    r.w = *reinterpret_cast<const uint32_t*>(a0 + 3 * pitch);
    return r;
}

// Accumulator tile(s) -> k-strided operand.  bf16 takes two consecutive 16-row tiles (32 k values),
// f32 one tile (16 k values); element order matches frag_kstrided's k-set.
template <typename T> TAV_DEV uint4 acc_to_kfrag(const f32x4* tiles);
template <> TAV_DEV uint4 acc_to_kfrag<bf16>(const f32x4* t) {
    return make_uint4(pack_bf16x2(t[0][0], t[0][1]), pack_bf16x2(t[0][2], t[0][3]), pack_bf16x2(t[1][0], t[1][1]),
                      pack_bf16x2(t[1][2], t[1][3]));
}
template <> TAV_DEV uint4 acc_to_kfrag<float>(const f32x4* t) {
    return make_uint4(__float_as_uint(t[0][0]), __float_as_uint(t[0][1]), __float_as_uint(t[0][2]),
                      __float_as_uint(t[0][3]));
}

// ---- LDS-DMA (global -> LDS without VGPRs) ----------------------------------------------------------------------
// 32-bit LDS byte address of a __shared__ pointer
TAV_DEV unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
}
// One wave instruction: lane l copies the 16 bytes at (sbase + voff_l) to LDS byte (lds_base + 16*l).  sbase (64-bit) and lds_base
// must be wave-uniform (SGPRs), voff is the lane's 32-bit byte offset: advancing a K-tile costs ONE v_add_u32 per instruction instead
// of a 64-bit multiply-add chain.  Issued from inline asm on purpose: hipcc would otherwise treat the DMA as a pending LDS write and
// put `s_waitcnt vmcnt(0)` in front of every following ds_read, serialising load and compute.  The caller waits with a counted
// vmcnt (wait_vmcnt0() in the simplest case) + a barrier before any wave reads the staged bytes.
// A wave-uniform value the compiler computed on the VALU (integer division, ...) pinned to an SGPR.  __builtin_amdgcn_readfirstlane of a
// value LLVM already knows to be uniform is folded away, and whether the dependent address chain then lives in SGPRs is a heuristic
// (SIFixSGPRCopies) -- glds16_s needs its base there, so this one is opaque.
// The hazard recognizer does not look inside inline asm, so the s_nops cover the gfx90a+ hazards by hand: "VALU writes VGPR -> readlane
// reads it" (1 wait state; the v_mov that feeds %1 is often the instruction right in front) and "VALU writes SGPR -> VALU reads it
// (2) / VMEM reads it (5)" behind.  Without them the lane read returns the register's PREVIOUS value.
TAV_DEV int to_sgpr(int v) { int s; asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(s) : "v"(v)); return s; }
// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so ids congruent mod 8 share an L2.
// Give each XCD a contiguous band of tiles (bijective for any grid size); the caller decodes the band index so that
// neighbouring tiles of one XCD reuse the same operand panel (GEMM: the A rows; attention: the K / V of one (batch, head)).
TAV_DEV int xcd_remap(int id, int total) {
    const int q = total >> 3, r = total & 7, x = id & 7, k = id >> 3;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + k;
}
TAV_DEV void glds16_s(const void* sbase, unsigned voff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_base)
                 : "memory");
}
// The GEMM main loops' form: no save / restore of M0.  That is legal only
// because nothing the compiler generates for these kernels reads or writes M0 (gfx9+ DS instructions do not use it); tools/check_isa.py
// verifies it on the built library (every M0 write is one of these and is followed by its LDS-DMA; no other M0 user exists).  Together with a
// scalar source base that the CALLER advances per K-tile (voff stays a lane constant) a piece costs 4 instructions (one compiler-made scalar add for the LDS address) instead of 7: the GEMM
// main loops are bound by instructions ISSUED per SIMD -- with the MFMAs removed they take the same time, with the DMA sequences removed
// 21 % less (profiles/r02_experiments.md), which is the share of issue slots those sequences held.
TAV_DEV void glds16_m0(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// Four LDS-DMA instructions in ONE statement: two 16-row pieces (1 KiB apart in LDS) of each of two operands.  One M0 save / restore for
// the group and the second piece's LDS address formed by s_add into M0: 15 scalar / vector-memory instructions instead of 4 x 5 + the
// address adds (the attention kernels are bound by instructions ISSUED per SIMD, scalar ones included: profiles/r03_experiments.md).
TAV_DEV void glds16_x4(const void* abase, const void* bbase, unsigned va0, unsigned va1, unsigned vb0, unsigned vb1, unsigned lds_a, unsigned lds_b) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"
                 "s_add_u32 m0, %7, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, %8, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(va0), "v"(va1), "v"(vb0), "v"(vb1), "s"(abase), "s"(bbase), "s"(lds_a), "s"(lds_b)
                 : "memory", "scc");
}
TAV_DEV void glds16_x2(const void* abase, const void* bbase, unsigned va0, unsigned vb0, unsigned lds_a, unsigned lds_b) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(va0), "v"(vb0), "s"(abase), "s"(bbase), "s"(lds_a), "s"(lds_b)
                 : "memory");
}
// Values loaded from global memory BEFORE a loop and consumed inside it: hipcc's waitcnt pass cannot tell how many younger loads
// a conditional in-loop prefetch has put in flight, so it protects every in-loop use with `s_waitcnt vmcnt(0/1)` -- which drains
// the prefetch issued a few instructions earlier and exposes the full memory latency on every tile.  settle() makes the register
// the output of an (empty) asm statement: the wait happens once, here, and the loop body carries none.
TAV_DEV void settle(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
TAV_DEV void settle(float& v) { asm volatile("" : "+v"(v)); }
TAV_DEV void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- wave reductions ----------------------------------------------------------------------------
TAV_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
TAV_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// max over the four lanes {i, i+16, i+32, i+48} (the 4 row groups of a 16x16 accumulator column) without an LDS round trip:
// v_permlane16_swap / v_permlane32_swap exchange rows between two copies of the value (gfx950), one v_max each.
TAV_DEV float vmax_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
TAV_DEV float max_over_row_groups(float v) {
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float a = vmax_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
    u = __float_as_uint(a);
    auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return vmax_raw(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
}

// sum over the same four lanes (two swaps, two adds; every lane ends with the total)
TAV_DEV float sum_over_row_groups(float v) {
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    u = __float_as_uint(a);
    auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

// exact-erf GELU (nn.GELU() default; reference utils/TAVFormer.py:398) and its derivative
TAV_DEV float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
TAV_DEV float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// bf16 operand paths: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16's 2^-9) -- ~12 VALU instructions instead of
// erff()'s ~35, and gelu' reuses the same exponential.  The fp32 policy keeps the exact erff forms above (parity runs).
TAV_DEV void gelu_parts_fast(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);             // exp(-x^2/2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;
    cdf = 0.5f + 0.5f * copysignf(erf_abs, x);
}
template <typename T> TAV_DEV float gelu_t(float x);
template <> TAV_DEV float gelu_t<float>(float x) { return gelu_f(x); }
template <> TAV_DEV float gelu_t<bf16>(float x) { float c, e; gelu_parts_fast(x, c, e); return x * c; }
template <typename T> TAV_DEV float gelu_grad_t(float x);
template <> TAV_DEV float gelu_grad_t<float>(float x) { return gelu_grad_f(x); }
template <> TAV_DEV float gelu_grad_t<bf16>(float x) { float c, e; gelu_parts_fast(x, c, e); return c + x * e * 0.39894228040143267794f; }

// gelu(x) and gelu'(x) together (they share the exponential / the erf): the FFN1 epilogue stores the derivative for the backward
// pass instead of the pre-activation, so the dgrad epilogue multiplies by it without any transcendental
template <typename T> TAV_DEV void gelu_both_t(float x, float& y, float& dy);
template <> TAV_DEV void gelu_both_t<float>(float x, float& y, float& dy) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    y = x * cdf;
    dy = cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
template <> TAV_DEV void gelu_both_t<bf16>(float x, float& y, float& dy) {
    float c, e;
    gelu_parts_fast(x, c, e);
    y = x * c;
    dy = fmaf(x * 0.39894228040143267794f, e, c);
}

// The same pair for four values at once, written on two-wide vectors so the arithmetic compiles to packed FP32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes of work per issue slot); rcp and exp2 stay scalar.  Same formula and constants as
// gelu_parts_fast -- the FFN1 epilogue of a 256 x 256 tile is VALU bound on exactly this.
typedef float f32x2 __attribute__((ext_vector_type(2)));
TAV_DEV void gelu_both2_fast(f32x2 x, f32x2& y, f32x2& dy) {
    const f32x2 z = f32x2{fabsf(x[0]), fabsf(x[1])} * 0.70710678118654752440f;
    const f32x2 den = z * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const f32x2 zz = z * z * -1.4426950408889634f;
    const f32x2 e = {__builtin_amdgcn_exp2f(zz[0]), __builtin_amdgcn_exp2f(zz[1])};                  // exp(-x^2/2)
    const f32x2 poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const f32x2 erf_abs = 1.0f - poly * e;
    const f32x2 cdf = f32x2{copysignf(erf_abs[0], x[0]), copysignf(erf_abs[1], x[1])} * 0.5f + 0.5f;
    y = x * cdf;
    dy = (x * 0.39894228040143267794f) * e + cdf;
}
template <typename T> TAV_DEV void gelu_both4_t(f32x4 x, f32x4& y, f32x4& dy);
template <> TAV_DEV void gelu_both4_t<float>(f32x4 x, f32x4& y, f32x4& dy) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { float a, b; gelu_both_t<float>(x[e], a, b); y[e] = a; dy[e] = b; }
}
template <> TAV_DEV void gelu_both4_t<bf16>(f32x4 x, f32x4& y, f32x4& dy) {
    f32x2 y0, d0, y1, d1;
    gelu_both2_fast(f32x2{x[0], x[1]}, y0, d0);
    gelu_both2_fast(f32x2{x[2], x[3]}, y1, d1);
    y = f32x4{y0[0], y0[1], y1[0], y1[1]};
    dy = f32x4{d0[0], d0[1], d1[0], d1[1]};
}

}  // namespace tav
