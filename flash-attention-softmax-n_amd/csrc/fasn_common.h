// fasn_common.h — device-side building blocks shared by the forward and backward kernels.
// gfx950 (CDNA4) only: wave64, v_mfma_f32_32x32x16_{bf16,f16}, ds_read_b64_tr_b16, v_permlane32_swap.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef FASN_PRIO8
#define FASN_PRIO8 0   // (round 5 A/B: static s_setprio 1 for the second-dispatched half of the 8-wave forward workgroups, cdna_hip_programming.md T5 static form)
#endif
#ifndef FASN_PRIO_WS
#define FASN_PRIO_WS 0   // (round 5 A/B, two-wave backward kernels: static s_setprio 1 for wave B (1) or wave A (2) of every SIMD's pair)
#endif

namespace fasn {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

#define FASN_DEV __device__ __forceinline__
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct bf16_tag {};
struct f16_tag {};

// Element-type traits: the MFMA operand vector (8 x 16-bit = 4 VGPRs) and conversions.
template <typename Tag>
struct ET;

template <>
struct ET<bf16_tag> {
    typedef bf16x8 vec8;
    typedef bf16x4 vec4;
    static FASN_DEV f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static FASN_DEV vec8 cvt8(f32x8 x) { return __builtin_convertvector(x, bf16x8); }
    // acc + lo + hi of one packed pair (v_dot2c_f32_bf16 against {1, 1}): a row sum costs one VALU issue per TWO weights
    static FASN_DEV float pair_sum(uint32_t pk, float acc) {
        typedef __bf16 pr __attribute__((ext_vector_type(2)));
        pr a;
        __builtin_memcpy(&a, &pk, 4);
        return __builtin_amdgcn_fdot2_f32_bf16(a, pr{(__bf16)1.0f, (__bf16)1.0f}, acc, false);
    }
    static FASN_DEV vec4 cvt4(f32x4 x) { return __builtin_convertvector(x, bf16x4); }
    static FASN_DEV float to_f32(uint16_t bits) { return __uint_as_float(((uint32_t)bits) << 16); }
};

template <>
struct ET<f16_tag> {
    typedef f16x8 vec8;
    typedef f16x4 vec4;
    static FASN_DEV f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static FASN_DEV vec8 cvt8(f32x8 x) { return __builtin_convertvector(x, f16x8); }
    static FASN_DEV float pair_sum(uint32_t pk, float acc) {
        typedef _Float16 pr __attribute__((ext_vector_type(2)));
        pr a;
        __builtin_memcpy(&a, &pk, 4);
        return __builtin_amdgcn_fdot2(a, pr{(_Float16)1.0f, (_Float16)1.0f}, acc, false);
    }
    static FASN_DEV vec4 cvt4(f32x4 x) { return __builtin_convertvector(x, f16x4); }
    static FASN_DEV float to_f32(uint16_t bits) {
        _Float16 h;
        __builtin_memcpy(&h, &bits, 2);
        return (float)h;
    }
};

// ---- LDS tile image -------------------------------------------------------------------------
// A tile is [rows][D] 16-bit elements, row-major, with the 16-byte chunks of each row XOR-permuted
// so that BOTH access patterns below are bank-conflict free (bank = (addr/4) mod 64):
//   (1) ds_read_b128 "row = lane&31, same chunk" (MFMA operand with the contraction along D),
//       serviced in 16-lane groups {0-3,12-15,20-27},{4-11,16-19,28-31},...
//   (2) ds_read_b64_tr_b16 "4 consecutive rows x 32 contiguous columns per half-wave"
//       (MFMA operand with the contraction along rows), serviced per 32 lanes.
// chunk' = chunk ^ f(row):
//   D=256 (512-B rows, two bank-rows each): as D=128 - the XOR stays inside a 256-byte half row
//   D=128 (256-B rows, one bank-row each):  f = ((row&3)<<2) | ((row>>2)&3)
//   D=64  (128-B rows, two per bank-row):   f = (bit1(row)<<2) | (bit2(row)<<1) | bit3(row)
//   D=32  (64-B rows, four per bank-row):   f = (row>>2)&3
template <int D>
FASN_DEV int swz_f(int row) {
    if constexpr (D == 128 || D == 256) {
        return ((row & 3) << 2) | ((row >> 2) & 3);
    } else if constexpr (D == 64) {
        return ((row & 2) << 1) | ((row >> 1) & 2) | ((row >> 3) & 1);
    } else {
        static_assert(D == 32, "unsupported head dim");
        return (row >> 2) & 3;
    }
}

// byte offset of (row, 16-byte chunk) inside a tile image
template <int D>
FASN_DEV int tile_off(int row, int chunk) {
    return row * (D * 2) + ((chunk ^ swz_f<D>(row)) << 4);
}

// MFMA operand "row = lane&31, k = 8*(lane>>5)+0..7" for k-step `ks` (16 columns per step):
// one ds_read_b128 of chunk 2*ks + hi from row `row0 + (lane&31)`.
template <typename E, int D>
FASN_DEV typename E::vec8 lds_read_rowfrag(const char* tile, int row, int ks, int hi) {
    const int off = tile_off<D>(row, 2 * ks + hi);
    u32x4 raw = *LDS_PTR(const u32x4, tile + off);
    typename E::vec8 r;
    __builtin_memcpy(&r, &raw, 16);
    return r;
}

// Transposed MFMA operand: "row(i) = column c0 + (lane&31) of the tile, k-slots = 8 tile rows".
// The 8 rows are {rbase + 4*hi + 0..3} and {rbase + 8 + 4*hi + 0..3}: exactly the rows whose values
// a lane with the same `hi` holds in registers 8t..8t+7 of a 32x32 MFMA accumulator (C layout
// row = (r&3) + 8*(r>>2) + 4*hi), so accumulator registers feed the other operand with no shuffle.
// Each ds_read_b64_tr_b16: within a 16-lane group, lane i supplies the address of 4 contiguous
// elements = tile[row0 + (i>>2)][col0 + 4*(i&3) ..]; lane i receives tile[row0 + j][col0 + i], j=0..3.
// (Measured and rejected: keeping the lane part of the address apart from rbase * row bytes, or pinning per-tile base registers in
// the non-unrolled D = 128 mask / bias loops. Both remove VALU instructions - 64 -> 16 per tile there, 256 -> 210 registers in the
// unrolled kernels - and both ran 1 - 1.5 % slower: plain (4,32,8192,128) 3.72 -> 3.78 ms, causal 2.16 -> 2.19, C4 4.42 -> 4.47.)
template <typename E, int D>
FASN_DEV typename E::vec8 lds_read_trfrag(const char* tile, int rbase, int cblk, int lane) {
    const int hi = lane >> 5;
    const int i = lane & 15;
    const int g1 = (lane >> 4) & 1;
    const int col = cblk * 32 + g1 * 16 + 4 * (i & 3);  // element column of this lane's 4-element piece
    const int chunk = col >> 3;
    const int sub = (col & 7) * 2;  // 0 or 8 bytes
    const int r0 = rbase + 4 * hi + (i >> 2);
    const int r1 = r0 + 8;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + tile_off<D>(r0, chunk) + sub));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + tile_off<D>(r1, chunk) + sub));
    s16x8 ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    typename E::vec8 r;
    __builtin_memcpy(&r, &ab, 16);
    return r;
}

// 16 bytes per lane straight from a buffer into LDS: LDS byte address = lds_base + 16*lane (lds_base wave-uniform, goes to
// M0), global address = descriptor base + voff + soff. Written as inline asm on purpose: for the builtin the compiler
// tracks the LDS write and puts `s_waitcnt vmcnt(0)` in front of later LDS reads it cannot prove disjoint, which would
// serialise the prefetch; here the caller owns the `s_waitcnt vmcnt(N)` + barrier that publishes the data. The s_nop is the
// wait state the ISA asks for between an SALU write of M0 and an LDS-DMA instruction (the compiler emits the same).
FASN_DEV void lds_dma16(u32x4 rsrc, uint32_t lds_base, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
}
// 4 bytes per lane into LDS (LDS byte address = lds_base + 4*lane): used as an L2 PREFETCH - the data lands in a junk area
// nobody reads, no register is written, so the request may stay in flight as long as it likes
FASN_DEV void lds_dma4(u32x4 rsrc, uint32_t lds_base, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
}
// raw buffer descriptor (stride 0, range-checked on `bytes`) as four SGPR words
FASN_DEV u32x4 make_rsrc_words(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
// the lane id from the execution mask (v_mbcnt_lo / _hi with all lanes active): a value nothing has to keep live
// (volatile asm: the builtins are pure, so the compiler would fold every call into ONE value computed at kernel entry and keep that live)
FASN_DEV int fresh_lane_id() {
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}
FASN_DEV uint32_t lds_addr(const void* p) {
    return (uint32_t)reinterpret_cast<uintptr_t>(LDS_PTR(const char, p));
}

// exchange a value between lane l and lane l^32 and return max(own, partner's)
FASN_DEV float max_across_halves(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
FASN_DEV float sum_across_halves(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Retire the global loads that filled loop-invariant register operands BEFORE a pipelined loop. Without this, hipcc's
// first in-loop use of such a register carries `s_waitcnt vmcnt(0)`, which on every iteration also drains the K/V
// prefetch issued a few instructions earlier (vmcnt only counts, it cannot tell old loads from new ones).
template <typename V>
FASN_DEV void retire_loads(V& v) {
    asm volatile("" : "+v"(v));
}

// ---- dropout -------------------------------------------------------------------------------------------------------
// Counter-based, layout-independent (stream definition 2, round 6): the keep/drop decision of attention weight (bh, row i, key j) is a
// 16-bit field of a hash of (seed, offset, bh, i, j >> 4) and j & 15; the weight is kept iff field >= thr (drop probability thr / 65536, so a
// requested p is honoured to 1.5e-5). Every kernel (forward and both backward kernels, which hold the score tile in different register
// layouts) recomputes the same bits. Mirror on the host: flash-attention-softmax-n_amd/dropout.py (the tests build the explicit mask for
// the oracle with it).
//   state  y = drop_mix(row_base(seed_lo, bh, i), seed_hi, j >> 4)        one 32-bit state per (row, GROUP OF 16 KEYS)
//   pair   p = (j & 15) >> 1 = 4 q + 2 h + c   (q = octet of the group, h = quad of the octet, c = pair of the quad)
//   word   w = mul24(rotl(y, 16 h + 4 q + 5 c), kDropMul[2 q + c])          one 24-bit multiply per PAIR of keys
//   field  f = (j & 1 ? w >> 16 : w & 0xffff) ^ 0x8000,  kept iff f >= thr   <=>  (int16) half >= (int16)(thr - 32768)
// Round 5 had one state per key quad and one multiply per key (6.4 VALU instructions per weight in the forward, 13 per score with the
// select): the state now serves 16 keys and a product both keys of a pair - its halves ARE the two fields - so the forward tests a
// PACKED pair of weights with three packed 16-bit instructions (saturating subtract, arithmetic shift, and) instead of a compare and a
// select per weight. The window rotation is additive in (h, q, c): a lane of the transposed accumulator layout, which holds the keys
// 8 q + 4 hi + {0..3} of a group, rotates the state ONCE by 16 hi and uses compile-time rotations and multipliers from there on.
// Statistics (tools/dropout_hash_stats.py): keep decisions of any two of a group's 16 keys (they share one 32-bit state: the window offsets
// 16 / 4 / 5 are the ones whose worst pair stays in the 1 / sqrt(N) noise of 4 M states at p = 0.1 .. 0.9 - with 16 / 8 / 4 one pair
// correlated at 6e-3), of adjacent rows / heads / seeds / offsets correlate below the noise of 4 M samples.
// The (seed, offset) pair comes by value in the launch parameters or - when the caller passes a device pointer - from two
// 64-bit words in device memory that a captured graph can advance between replays (fasn_rng_advance).
struct DropSeed {
    uint32_t lo, hi;
};
FASN_DEV DropSeed drop_seed(uint32_t seed_lo, uint32_t seed_hi, const uint64_t* rng) {
    DropSeed d{seed_lo, seed_hi};
    if (rng != nullptr) {   // same folding as the host side (fasn_api.hip: build_fwd)
        const uint64_t s = rng[0], o = rng[1];
        d.lo = (uint32_t)s ^ ((uint32_t)o * 0x9E3779B1u);
        d.hi = (uint32_t)(s >> 32) + (uint32_t)(o >> 32);
    }
    d.lo = __builtin_amdgcn_readfirstlane(d.lo);
    d.hi = __builtin_amdgcn_readfirstlane(d.hi);
    return d;
}
FASN_DEV uint32_t drop_row_base(uint32_t seed_lo, uint32_t bh, uint32_t row) {
    return (seed_lo ^ (bh * 0x9E3779B1u)) + row * 0x85EBCA77u;
}
// The hash uses only full-rate VALU operations: 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24: low 24 bits of both operands, low 32
// bits of the product), rotates (v_alignbit_b32), adds and xors - a 32-bit v_mul_lo_u32 issues at a quarter of that rate.
FASN_DEV uint32_t rotl32(uint32_t x, uint32_t r) { return __builtin_amdgcn_alignbit(x, x, 32u - r); }
FASN_DEV uint32_t drop_mix(uint32_t row_base, uint32_t seed_hi, uint32_t key_group) {   // key_group = key >> 4
    uint32_t x = (row_base + __umul24(key_group, 0x9E3779u)) ^ seed_hi;
    const uint32_t a = __umul24(x, 0xC2B2AFu), b = __umul24(rotl32(x, 20), 0x85EBCBu);
    uint32_t y = a + rotl32(b, 13);
    y ^= y >> 15;
    return y + rotl32(y, 9);
}
// kDropMul[2 q + c]
constexpr uint32_t drop_mul_of(int i) { return i == 0 ? 0x2C1B3Du : i == 1 ? 0x297A2Du : i == 2 ? 0x1B56C5u : 0x7ED55Du; }
// word of pair p = 4 q + 2 h + c (compile-time p): both fields of the keys 2 p and 2 p + 1 of the group
template <int P>
FASN_DEV uint32_t drop_pair_word(uint32_t y) {
    constexpr int Q = P >> 2, H = (P >> 1) & 1, C = P & 1, R = 16 * H + 4 * Q + 5 * C;
    return __umul24(R == 0 ? y : rotl32(y, (uint32_t)R), drop_mul_of(2 * Q + C));
}
// the same from a state that is already rotated by 16 h (yh = rotl(y, 16 hi), once per state, where h is the lane's half-wave)
template <int Q, int C>
FASN_DEV uint32_t drop_pair_word_h(uint32_t yh) {
    constexpr int R = 4 * Q + 5 * C;
    return __umul24(R == 0 ? yh : rotl32(yh, (uint32_t)R), drop_mul_of(2 * Q + C));
}
// thresholds of a launch, derived once: t16 = thr - 32768 as int16 (the fields are compared as signed halves),
//   hi32 = t16 << 16: the HIGH field of a word is kept iff (int32) word >= hi32 (the low half cannot change the answer)
//   pk1  = both halves t16 - 1: sat_sub_i16(pk1, word) is negative in exactly the halves that are kept
struct DropThr {
    int32_t hi32;
    int16_t t16;
    uint32_t pk1;
};
FASN_DEV DropThr drop_thr(uint32_t thr) {   // thr in [1, 65535]
    DropThr t;
    const int32_t s = (int32_t)thr - 32768;
    t.t16 = (int16_t)s;
    t.hi32 = (int32_t)((uint32_t)s << 16);
    t.pk1 = ((uint32_t)(s - 1) & 0xffffu) * 0x10001u;
    return t;
}
FASN_DEV bool drop_keep_hi(uint32_t word, const DropThr& t) { return (int32_t)word >= t.hi32; }          // odd key of the pair
FASN_DEV bool drop_keep_lo(uint32_t word, const DropThr& t) { return (int16_t)(uint16_t)word >= t.t16; }  // even key of the pair
template <int S>
FASN_DEV bool drop_keep_half(uint32_t word, const DropThr& t) { return S ? drop_keep_hi(word, t) : drop_keep_lo(word, t); }
// 0xffff in the halves of `word` that are kept, 0 in the dropped ones: and-ed onto a packed pair of 16-bit weights
FASN_DEV uint32_t drop_keep_mask_pk(uint32_t word, const DropThr& t) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    s16x2 w, th;
    __builtin_memcpy(&w, &word, 4);
    __builtin_memcpy(&th, &t.pk1, 4);
    const s16x2 d = __builtin_elementwise_sub_sat(th, w) >> (short)15;
    uint32_t m;
    __builtin_memcpy(&m, &d, 4);
    return m;
}
// The keep bits of the 16 weights a lane of a row-owning kernel (forward, dQ) holds of one 32-key block, in both register layouts -
//   plain        : register r = key 8 (r >> 2) + 4 hi + (r & 3) of the block: pairs (q = g & 1, h = hi, c) of the groups g >> 1 (g = r >> 2).
//                  Each half-wave computes ONE of the block's two states and both get both through one v_permlane32_swap (all 64 lanes
//                  must be active and construct the block together); the states are rotated once by 16 hi (rh = 32 - 16 hi, the alignbit
//                  amount), the windows from there on are compile-time;
//   key-permuted : register r = key 16 hi + r (vector mask / bias modes): the lane's 16 keys are the whole group 2 kb + hi.
// group0 = first key of the block >> 4. Only the two states stay live: word(i) - the product whose halves are the fields of registers 2 i and
// 2 i + 1 - is formed where it is used (i compile-time after unrolling).
template <bool KPERM>
struct DropBlock {
    uint32_t st[2];
    FASN_DEV DropBlock(uint32_t row_base, uint32_t seed_hi, uint32_t group0, int hi, uint32_t rh) {
        const uint32_t own = drop_mix(row_base, seed_hi, group0 + (uint32_t)hi);
        if constexpr (KPERM) {
            st[0] = st[1] = own;
        } else {
            const auto both = __builtin_amdgcn_permlane32_swap(own, own, false, false);   // [0]: the lower half-wave's state (group0), [1]: the upper one's (group0 + 1)
            st[0] = __builtin_amdgcn_alignbit(both[0], both[0], rh);                      // rotated left by 16 hi
            st[1] = __builtin_amdgcn_alignbit(both[1], both[1], rh);
        }
    }
    FASN_DEV uint32_t word(int i) const {   // i = r >> 1
        if constexpr (KPERM) {
            switch (i & 7) {
                case 0: return drop_pair_word<0>(st[0]);
                case 1: return drop_pair_word<1>(st[0]);
                case 2: return drop_pair_word<2>(st[0]);
                case 3: return drop_pair_word<3>(st[0]);
                case 4: return drop_pair_word<4>(st[0]);
                case 5: return drop_pair_word<5>(st[0]);
                case 6: return drop_pair_word<6>(st[0]);
                default: return drop_pair_word<7>(st[0]);
            }
        } else {
            const uint32_t yh = st[(i >> 2) & 1];
            switch (i & 3) {   // 2 q + c
                case 0: return drop_pair_word_h<0, 0>(yh);
                case 1: return drop_pair_word_h<0, 1>(yh);
                case 2: return drop_pair_word_h<1, 0>(yh);
                default: return drop_pair_word_h<1, 1>(yh);
            }
        }
    }
    FASN_DEV bool keep(int r, const DropThr& t) const { return (r & 1) ? drop_keep_hi(word(r >> 1), t) : drop_keep_lo(word(r >> 1), t); }
    FASN_DEV uint32_t keep_mask_pk(int i, const DropThr& t) const { return drop_keep_mask_pk(word(i), t); }   // registers 2 i, 2 i + 1 as a packed pair
};
FASN_DEV uint32_t drop_rh_of(int hi) { return 32u - 16u * (uint32_t)hi; }

// a lane-dependent key (the dK/dV kernels: a lane owns one key): rotation, multiplier and the shift that brings the key's field into
// the high half of the word, picked per lane, once
struct DropLane {
    uint32_t rot, mul, sh;
};
FASN_DEV DropLane drop_lane(int key) {
    const int p = (key & 15) >> 1, q = p >> 2, h = (p >> 1) & 1, c = p & 1, i = 2 * q + c;
    DropLane d;
    const int r = 16 * h + 4 * q + 5 * c;
    d.rot = r == 0 ? 32u : (uint32_t)r;   // (alignbit by 32 - rot: rotate by 32 = by 0)
    d.mul = i == 0 ? 0x2C1B3Du : i == 1 ? 0x297A2Du : i == 2 ? 0x1B56C5u : 0x7ED55Du;
    d.sh = (key & 1) ? 0u : 16u;
    return d;
}
// the lane's field of state y, moved into the high half: kept iff (int32) result >= DropThr::hi32
FASN_DEV uint32_t drop_word(uint32_t y, DropLane d) { return __umul24(__builtin_amdgcn_alignbit(y, y, 32u - d.rot), d.mul) << d.sh; }
// In the dK/dV kernels a lane owns a KEY and its 16 registers of a block are 16 ROWS - (r & 3) + 8 (r >> 2) + 4 hi - so the state of
// (row, key group) would be needed once per weight. The four lanes of a key quad (same group) hold the same rows: each computes the state of
// ONE row of every 4-row register group (row 8 g + 4 hi + (lane & 3)) and the quad exchanges them with DPP quad_perm broadcasts.
FASN_DEV uint32_t quad_bcast(uint32_t v, int i) {   // the value of lane (quad base + i); i is a compile-time constant at every call site
    switch (i & 3) {
        case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xf, 0xf, false);
        case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xf, 0xf, false);
        case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xf, 0xf, false);
        default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xf, 0xf, false);
    }
}
// own[g] = state of row row0 + 8 g + 4 hi + (lane & 3) and this lane's key group (row0: first row of the 32-row block)
FASN_DEV void drop_quad_states(uint32_t (&own)[4], uint32_t seed_lo, uint32_t seed_hi, uint32_t bh, int row0, int hi, int lane, int key) {
#pragma unroll
    for (int g = 0; g < 4; ++g) own[g] = drop_mix(drop_row_base(seed_lo, bh, (uint32_t)(row0 + 8 * g + 4 * hi + (lane & 3))), seed_hi, (uint32_t)(key >> 4));
}
// kept? - weight (register r of the block, this lane's key)
FASN_DEV bool drop_keep_quad(const uint32_t (&own)[4], int r, const DropLane& dl, const DropThr& t) {
    return (int32_t)drop_word(quad_bcast(own[r >> 2], r & 3), dl) >= t.hi32;
}
// one weight, everything at run time (element-load and fp32 kernels)
FASN_DEV bool drop_keep_at(const DropSeed& dsd, uint32_t bh, uint32_t row, uint32_t key, uint32_t thr) {
    const uint32_t y = drop_mix(drop_row_base(dsd.lo, bh, row), dsd.hi, key >> 4);
    return (int32_t)drop_word(y, drop_lane((int)key)) >= drop_thr(thr).hi32;
}

FASN_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// global 16-byte load / store helpers (pointers are 16-B aligned by the host-side contract)
FASN_DEV u32x4 gload16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
FASN_DEV void gstore16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
FASN_DEV void gstore8(void* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }

// One 32-column block of a row-per-lane result (accumulator registers 4g .. 4g+3 = columns 8g + 4hi .. + 3 of the lane's row; the partner
// lane in the other half-wave holds the other four of every eight): scaled, rounded to the element type and written as TWO 16-byte stores
// per lane instead of four 8-byte ones - groups (g, g+1) are paired through v_permlane32_swap, after which the lower half-wave holds
// columns 8g .. 8g+7 and the upper one 8(g+1) .. 8(g+1)+7. The store tail of a row-per-lane epilogue is bound by instruction issue
// (cdna_hip_programming.md T21). `blk` = the row's address + the block's column offset; both half-waves must be active for the row.
template <typename E>
FASN_DEV void store_block_wide(char* blk, const f32x16& acc, float sc, int hi) {
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
        u32x2 pk[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = acc[4 * (g + j) + e] * sc;
            typename E::vec4 y = E::cvt4(x);
            __builtin_memcpy(&pk[j], &y, 8);
        }
        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        gstore16(blk + (8 * g + 8 * hi) * 2, u32x4{r0[0], r1[0], r0[1], r1[1]});
    }
}

// the same block as four 8-byte stores per lane: for the kernels that sit at their register limit (the paired form holds two groups at a
// time and cost them 1 - 13 spilled registers in the epilogue)
template <typename E>
FASN_DEV void store_block_narrow(char* blk, const f32x16& acc, float sc, int hi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 x;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = acc[4 * g + e] * sc;
        typename E::vec4 y = E::cvt4(x);
        u32x2 raw;
        __builtin_memcpy(&raw, &y, 8);
        gstore8(blk + (8 * g + 4 * hi) * 2, raw);
    }
}

// XCD-aware block -> (batch*head, q-block) map. Blocks are dispatched round-robin over the 8 XCDs
// (block b -> XCD b%8, observed, used for speed only): give every XCD whole heads so one head's
// K/V stays in one XCD's L2.
FASN_DEV void block_to_work(int bid, int nbh, int nblk_per_head, int& bh, int& blk) {
    if ((nbh & 7) == 0) {
        const int xcd = bid & 7;
        const int j = bid >> 3;
        bh = (j / nblk_per_head) * 8 + xcd;
        blk = j % nblk_per_head;
    } else {
        bh = bid / nblk_per_head;
        blk = bid % nblk_per_head;
    }
}

// Causal launches of kernels that do not pair their blocks (the two-wave kernels): the dispatcher hands workgroups out in order, so with whole heads
// one after the other (above) the heavy blocks of the last heads start late and the launch ends on them (+18 % in a list-scheduling model of
// (4,16,2048,128), measured 0.69 of the non-causal time where 0.53 is the work). Heads are therefore taken in GROUPS of G per XCD and the blocks of a
// group handed out block index by block index (heaviest first, see the callers): a decreasing sequence per group, which list scheduling packs tightly.
// G = as many heads as share the XCD's 4 MiB L2 with their K / V (or Q / dO) - at most 4, a power of two that divides the XCD's heads.
#ifndef FASN_CAUSAL_GROUPS
#define FASN_CAUSAL_GROUPS 1
#endif
FASN_DEV int causal_head_group(int nbh, int rows, int D) {
    if ((nbh & 7) != 0) return 1;
    const int hx = nbh >> 3;
    const long per_head = 4L * rows * D;   // two 16-bit matrices
    int g = 1;
    while (g < 4 && (hx % (2 * g)) == 0 && 2L * g * per_head <= (4L << 20)) g *= 2;
    return g;
}
FASN_DEV void block_to_work_grouped(int bid, int nbh, int nblk_per_head, int G, int& bh, int& blk) {
    if ((nbh & 7) == 0 && G > 1) {
        const int xcd = bid & 7, j = bid >> 3;
        const int per = G * nblk_per_head, g = j / per, r = j % per;
        blk = r / G;
        bh = (g * G + r % G) * 8 + xcd;
    } else {
        block_to_work(bid, nbh, nblk_per_head, bh, blk);
    }
}

}  // namespace fasn
