// fasn_fwd_kernel.h — forward: tiled  S^T = K Q^T  ->  online softmax_n  ->  O^T += V^T P^T.
//
// Math (per (b,h), query row i), cf. reference flash_attention_softmax_n/core/functional.py:15-29,32-93:
//   x_ij = scale * q_i.k_j + bias_ij     (masked / non-causal-visible -> -inf)
//   LSE_i = log(n + sum_j exp(x_ij)),  P_ij = exp(x_ij - LSE_i),  O_i = sum_j P_ij v_j
//
// The "+n" is a virtual sink column with logit 0 and value 0 (what flash_attn.py:66-73 builds with
// n zero-padded K/V rows, generalised to real n): the online-softmax state starts at
// (m, l, acc) = (0, n, 0) instead of (-inf, 0, 0); the FA-2 recurrence is otherwise unchanged.
//
// Orientation. Everything is computed transposed so that ONE LANE OWNS ONE QUERY ROW end to end:
//   S^T[key][q] = mfma(A = K rows (ds_read_b128), B = Q^T (registers))  -> lane (q=lane&31, hi) holds
//                 keys (r&3) + 8*(r>>2) + 4*hi, r = 0..15, of each 32-key block
//   row max / row sum: in-lane over its registers + one v_permlane32_swap with lane^32
//   P^T: those same registers, exponentiated and packed to 16 bit, ARE the B operand of the next MFMA
//   O^T[d][q] += mfma(A = V^T (ds_read_b64_tr_b16 with the matching key permutation), B = P^T)
// so there is no LDS round trip and no cross-lane shuffle for P, and alpha/l/m are per-lane scalars.
//
// Work decomposition: workgroup = 4 waves = BM = 4*QB*32 query rows of one (b,h); each wave owns
// QB 32-row blocks and reuses every K / V fragment it reads from LDS for all of them. K/V tiles of 64
// keys are staged global -> registers -> swizzled LDS image, double buffered, one barrier per tile.
#pragma once
#include <type_traits>
#include "fasn_common.h"

#ifndef FASN_UNR3_D128
#define FASN_UNR3_D128 1
#endif
#ifndef FASN_SEED_KEEPALIVE
#define FASN_SEED_KEEPALIVE 1
#endif
#ifndef FASN_FWD_UNR2
#define FASN_FWD_UNR2 1
#endif
#ifndef FASN_VEC_PAIR
#define FASN_VEC_PAIR 1
#endif
#ifndef FASN_DROP_PAIR
#define FASN_DROP_PAIR 1   // paired blocks in the causal dropout forward too (round 6)
#endif
namespace fasn {

// MODE_GENERAL: mask and/or bias through 4-key vector (buffer) loads - needs key stride 1 and aligned rows (bias_vec /
// mask_vec). MODE_GENERAL_SLOW: anything else (fp32 or unaligned bias, strided mask) through per-element loads.
// The vector kernels are specialised at compile time on which operands exist (MODE_GENERAL_B / _M / _BM), so that every
// load in the tile loop is unconditional and hipcc can emit counted s_waitcnt vmcnt(N) instead of draining the queue.
enum { MODE_PLAIN = 0, MODE_CAUSAL = 1, MODE_GENERAL = 2 /* = bias + mask */, MODE_GENERAL_SLOW = 3, MODE_GENERAL_B = 4, MODE_GENERAL_M = 5,
       MODE_KEYPAD = 6 /* boolean mask that depends on (batch, head, key) only - key padding - and no bias */,
       MODE_BIAS_KEYPAD = 7 /* vector bias + key-padding mask (e.g. ALiBi on a padded batch): the bias through the vector path, the mask
                               as the per-tile visibility word of MODE_KEYPAD - no mask image, no per-element byte test, padded tiles skipped */ };
constexpr bool mode_has_vbias(int M) { return M == MODE_GENERAL || M == MODE_GENERAL_B || M == MODE_BIAS_KEYPAD; }
constexpr bool mode_has_vmask(int M) { return M == MODE_GENERAL || M == MODE_GENERAL_M; }
constexpr bool mode_is_vector(int M) { return mode_has_vbias(M) || mode_has_vmask(M); }
constexpr bool mode_has_keypad(int M) { return M == MODE_KEYPAD || M == MODE_BIAS_KEYPAD; }

struct FwdParams {
    const char* q;
    const char* k;
    const char* v;
    char* o;
    float* lse;
    const uint8_t* mask;
    const char* bias;
    int64_t qs[3], ks[3], vs[3], os[3];  // element strides (batch, head, row); feature stride 1
    int64_t ms[4], bs[4];                // mask / bias element strides (batch, head, q, key)
    int B, H, Sq, Sk;
    int nqblk;      // query blocks per head
    int causal;
    int bias_f32;   // bias elements are fp32 (else same 16-bit type as q)
    int bias_vec;   // bias key stride 1 and every row 8-byte (16-bit) / 16-byte (fp32) aligned: 4 keys per load
    int mask_vec;   // mask key stride 1 and every row 4-byte aligned: 4 keys per load
    int batch_inner;  // bias/mask broadcast over batch: schedule the batch innermost so a head's bias tile is reused from L2
    unsigned kbytes, vbytes;  // byte extent of one (b,h) K / V matrix: Sk * row_stride * 2 (buffer descriptor range)
    unsigned bias_bytes, mask_bytes;  // byte extent of one (b,h) bias / mask slice: (Sq-1)*row_stride + Sk elements
    unsigned drop_thr;        // dropout: drop an attention weight iff its 16-bit hash field < drop_thr (0 = no dropout)
    unsigned seed_lo, seed_hi;
    const uint64_t* rng;      // optional device {seed, offset}: replaces seed_lo / seed_hi (graph-safe dropout state)
    float drop_scale;         // 65536 / (65536 - drop_thr): applied to O (and to dP in the backward)
    float c;        // scale * log2(e)
    float n;        // softmax_n
    // split-K (SPLIT kernels, short query / long key "decode" shapes): the keys of one (b,h, query block) are divided over
    // nsplit workgroups of tps tiles each; every workgroup writes its un-normalised fp32 accumulator and (m, l) per row,
    // fasn_fwd_combine_kernel merges them. The sink (+n) belongs to split 0.
    int kvg;        // query heads per K/V head (grouped-query attention); 1 = one K/V head per query head
    int keypad_fallback;   // MODE_KEYPAD launches: the general mode (vector or element-load) the same mask would otherwise take
    int nsplit, tps;
    int pair;       // causal (MODE_CAUSAL kernels): one workgroup takes query block r AND block nqblk-1-r of its head, one after the other
    int kprot;      // length-paired batch elements: rotate the key walk of the second element (kpair_plan's `lead`); 0 = developer A/B
    float* part_o;   // [B*H][nsplit][Sq][D]
    float* part_ml;  // [B*H][nsplit][Sq][2]
    int* xq;         // eight zeroed item counters (caller's workspace, fasn_fwd_ws): dynamic deal of the (head, query block) items across XCDs; nullptr = static deal
#ifdef FASN_DEV_VARIANTS
    unsigned long long* timeline;   // developer library: per workgroup {t_entry, t_loop, t_epilogue, t_end, hw_id, xcc_id, ntiles, 0} (100 MHz clock)
#endif
};
#ifdef FASN_DEV_VARIANTS
#define FASN_STAMP(slot) do { if (p.timeline != nullptr && threadIdx.x == 0) p.timeline[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FASN_STAMP(slot) do { } while (0)
#endif

#ifndef FASN_XADDR
#define FASN_XADDR 1
#endif
constexpr int KT = 64;  // keys per tile
// key-padding modes: visibility words (one per K/V tile) kept in LDS by the forward kernels: 4 KiB, Sk <= 32768. Longer key
// sequences take the dense-mask general mode of the same mask (fasn_api.hip).
constexpr int kFwdKpMaxTiles = 512;
// fast-path guard: a lane's partial row sum over one tile (32 values) must stay <= 2^8; any single p > 2^8 trips it
constexpr float kSumLimit = 256.0f;

// Visibility words of a key-padding mask row: words[t] bit j = key 64t + j is visible (mask byte != 0 and key < Sk), for tiles
// [0, ntiles). Thread `tid` of `nthreads` turns 16 mask bytes into 16 bits per step (one 16-byte load; a row that is not dword
// aligned takes byte loads) and stores them as one uint16 - the caller's barrier publishes the words. mrow == nullptr: no mask.
FASN_DEV void kp_build_words(uint64_t* words, const uint8_t* mrow, int Sk, int ntiles, int tid, int nthreads) {
    uint16_t* const bits16 = reinterpret_cast<uint16_t*>(words);
    const bool aligned = (reinterpret_cast<uintptr_t>(mrow) & 3) == 0;
    const __amdgpu_buffer_rsrc_t mrs16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(mrow), 0, (mrow && aligned) ? (unsigned)((Sk + 3) & ~3) : 0u, 0x00020000);   // whole dwords: the range check works per dword
    for (int c = tid; c * 16 < ntiles * 64; c += nthreads) {
        uint32_t bits = 0;
        if (mrow == nullptr) {
            bits = 0xffffu;
        } else if (aligned) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(mrs16, c * 16, 0, 0);   // bytes past the last dword read as 0
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // byte != 0 -> bit: OR-fold each byte into its bit 0, then gather the four flags by a multiply
                uint32_t t = w[e] | (w[e] >> 4);
                t |= t >> 2;
                t |= t >> 1;
                bits |= (((t & 0x01010101u) * 0x01020408u) >> 24 & 0xfu) << (4 * e);
            }
        } else {
            for (int e = 0; e < 16; ++e)
                if (c * 16 + e < Sk && mrow[c * 16 + e] != 0) bits |= 1u << e;
        }
        const int left = Sk - c * 16;   // keys past Sk are hidden (also the bytes of the last dword behind Sk)
        if (left < 16) bits = left <= 0 ? 0u : (bits & ((1u << left) - 1u));
        bits16[c] = (uint16_t)bits;
    }
}

// Length-paired batch elements (forward and two-wave dQ kernels, bias broadcast over a key-padded batch). Every workgroup derives
// the same plan from the key-padding mask [B,1,1,Sk]: walk length of batch element i = index of its last 16-key chunk with a visible
// key + 1 (what the kernels' tile loops are bounded by), rank by (length descending, index ascending), pair slot r = (r-th longest,
// r-th shortest); with an odd B the median element runs alone (b1 == b0). Returns false - plain schedule - when the mask does not
// qualify (more than 64 batch elements or 64 KiB of mask, rows not movable in 16-byte pieces, a per-head mask) or when the
// batch is not ragged enough to pay for it (mean length >= 0.85 of the longest). `scratch`: B ints of LDS nobody uses yet; barriers inside, so the whole
// workgroup calls it. Uniform result.
constexpr int kPairMaxBatch = 64, kPairMaxBytes = 64 * 1024;
// `lead` (tiles): how many tile steps BEFORE the workgroup of the longest element this workgroup finishes its first element - the second
// passes of a (head, query block) group then start `lead` steps apart. A workgroup that rotates the key walk of its second element by
// that many tiles (walk tile (t - lead) mod n instead of t: online softmax and the dQ sum do not care about the order) meets the other
// workgroups of its group at the same bias tile at the same time, and the tile comes out of the L2 instead of being fetched per workgroup.
FASN_DEV bool kpair_plan(const FwdParams& p, char* scratch, int tid, int slot, int& b0, int& b1, int& lead) {
    lead = 0;
    if (p.mask == nullptr || p.ms[1] != 0 || p.B < 2 || p.B > kPairMaxBatch || (int64_t)p.B * p.Sk > kPairMaxBytes || (p.Sk & 15) != 0 ||
        (p.ms[0] & 15) != 0 || (reinterpret_cast<uintptr_t>(p.mask) & 15) != 0 || p.pair == 2)   // (pair 2 / 3: developer override)
        return false;
    int* const len = reinterpret_cast<int*>(scratch);
    const int nthreads = (int)blockDim.x;
    const int lane = tid & 63;
    for (int i = tid; i < p.B; i += nthreads) len[i] = 0;
    __syncthreads();
    const int cpr = p.Sk >> 4;   // 16-byte chunks per mask row
    // one batch element at a time: a thread keeps the last non-zero chunk it saw, the wave folds its 64 candidates with shuffles and ONE lane
    // posts the result (round 4 posted from every lane: 64 LDS atomics on one address per instruction = all of SQ_LDS_BANK_CONFLICT of the
    // config-4 launches, 1550 cycles per workgroup)
    for (int bb = 0; bb < p.B; ++bb) {
        int mine = 0;
        for (int k16 = tid; k16 < cpr; k16 += nthreads) {
            const u32x4 w = *reinterpret_cast<const u32x4*>(p.mask + (int64_t)bb * p.ms[0] + 16 * k16);
            if ((w[0] | w[1] | w[2] | w[3]) != 0u) mine = k16 + 1;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine = max(mine, __shfl_xor(mine, o));
        if (lane == 0 && mine > 0) atomicMax(&len[bb], mine);
    }
    __syncthreads();
    const int mine = lane < p.B ? len[lane] : -1;
    __syncthreads();   // `len` has been read: the scratch may be reused by the caller
    int rank = 0, lmax = 0, lsum = 0;
    for (int k = 0; k < p.B; ++k) {
        const int lk = __builtin_amdgcn_readlane(mine, k);
        rank += (lk > mine || (lk == mine && k < lane)) ? 1 : 0;
        lmax = max(lmax, lk);
        lsum += lk;
    }
    // worth it? the plain schedule costs max(len) per round, the paired one about mean(len) x 1.18 (measured: its tile steps are
    // slower, the bias is fetched twice): pair when the mean walk length is below 0.85 of the longest
    if ((int64_t)lsum * 20 >= (int64_t)lmax * p.B * 17 && p.pair != 3) return false;
    const uint64_t m0 = __ballot(lane < p.B && rank == slot), m1 = __ballot(lane < p.B && rank == p.B - 1 - slot);
    b0 = __builtin_ctzll(m0);
    b1 = __builtin_ctzll(m1);
    lead = (lmax + 3) / 4 - (__builtin_amdgcn_readlane(mine, b0) + 3) / 4;
    return true;
}

// the instantiations that can deal their items dynamically across XCDs (FwdParams::xq, draw_item below): the plain and causal D = 64 kernels
constexpr bool fwd_xq_kernel(int D, int MODE, int SPLIT, int VH, int DROP) { return D == 64 && (MODE == MODE_PLAIN || MODE == MODE_CAUSAL) && !SPLIT && VH == 1 && !DROP; }
constexpr int kXqSurplus = 16;   // surplus workgroups per XCD of such a launch (2 % speed difference of 256 .. 1024 items per XCD is 5 .. 20 items)
// DROP: attention-weight dropout compiled in (separate instantiations so the no-dropout kernels keep their registers).
// RING: 1 = two staging register sets, K/V tiles are loaded TWO tiles ahead (the loop body is instantiated twice with the
// sets swapped). One tile of lead is about 1.2 us at D=64, less than a first-touch HBM miss under load; in-order vmcnt
// makes any older outstanding load block the wait, so the extra lead has to come from a second register set.
// RING: 2 = no staging registers at all: `buffer_load_dwordx4 ... lds` moves each 16-byte chunk straight from HBM/L2 into
// the LDS tile image (LDS address = wave base + 16*lane, so the swizzle is applied by choosing WHICH global chunk a lane
// fetches), three LDS tile buffers, loads issued two tiles ahead, `s_waitcnt vmcnt` before the barrier that publishes a tile.
// FOLD (round 5, causal launches at the plain kernel's tuning point: 64 rows per wave, two workgroups per CU): (i) the key walk of a block is
// cut in two - the tiles every row of the block sees run the plain tile body (no compare, no per-wave classification, loop unrolled
// by its buffers), the up to five tiles that touch the block's diagonal run a second instantiation of the body that classifies per
// 32-row block; (ii) the wave's two 32-row blocks are block `wave` and block `7 - wave` of the workgroup's eight ("folded"), so every wave
// has work in every diagonal step (with contiguous rows wave 0 sat idle for three of the four steps, behind the barrier) and the diagonal
// steps cost about half a full step each instead of more than a full one (the exact softmax path of a 64-row wave on a tile of which it
// needs a quarter). Reference: the causal bound of flash_attn_triton.py:89-112 (start_n loop to (start_m + 1) * BLOCK_M).
// BF32 (round 5): the additive bias of a vector general mode holds fp32 elements next to 16-bit q / k / v (what Hugging Face models hand over
// as additive masks, reference core/flash_attn.py:100-113): the wave's bias image is [32 rows][64 keys] fp32 (8 KiB, rows of 256 bytes swizzled
// like a D = 128 tile), moved by the same coalesced 16-byte LDS-DMA pieces, read 64 bytes per 32-key block and lane, and the start value of a
// score is one fma on the fp32 value - no 16-bit unpack. Round 4 sent these calls through the element-load kernels (3-5 x slower).
template <typename Tag, int D, int QB, int MODE, int OCC, int NW = 4, int PRIO = 0, int DROP = 0, int RING = 0, int SPLIT = 0, int SEED = 0, int VH = 1, int FOLD = 0, int BF32 = 0>
__global__ void __launch_bounds__(NW * 64, OCC) fasn_fwd_kernel(const FwdParams p) {
    static_assert(!BF32 || (mode_has_vbias(MODE) && !SPLIT), "fp32 bias image: the vector bias modes");
    static_assert(!FOLD || (MODE == MODE_CAUSAL && QB == 2 && RING == 2 && !SPLIT && !DROP && VH == 1 && D <= 64), "folded two-phase walk: the causal 64-rows-per-wave kernel");
    static_assert(!SEED || MODE != MODE_GENERAL_SLOW, "seeded accumulators: not for the element-load kernels");
    static_assert(!SPLIT || (RING != 1 && DROP == 0), "split-K: single-set or direct-to-LDS staging");
    // VH = 2 (D = 256): two workgroups per query block, each with the full QK^T and softmax but HALF of the output features
    // (O^T for 128 features is 64 accumulator registers instead of 128); the grid is doubled, block 2j + v owns feature half v.
    static_assert(VH == 1 || (VH == 2 && !SPLIT), "feature halves: not with split-K");
    using E = ET<Tag>;
    using vec8 = typename E::vec8;
    constexpr bool PSUM = SEED >= 2;   // fast-path row sums from the packed weights (dropout masks the packed weights AFTER the sum)
    constexpr bool UNR3 = RING == 2 && (D <= 64 || (FASN_UNR3_D128 && NW == 8 && !mode_is_vector(MODE) && !DROP));  // direct-to-LDS loop unrolled by its three buffers
    constexpr bool UNR2 = RING == 0 && FASN_FWD_UNR2;   // single-set staging: loop unrolled by its two LDS buffers
    constexpr int NT = NW * 64;
    constexpr int BM = NW * QB * 32;
    constexpr int ROWB = D * 2;
    constexpr int TILEB = KT * ROWB;
    constexpr int KS = D / 16;   // k-steps of QK^T
    constexpr int DB = D / 32 / VH;   // 32-wide output column blocks of this workgroup
    constexpr int CPR = D / 8;   // 16-B chunks per row
    constexpr int NLD = (KT * CPR) / NT;  // staging loads per thread per tensor
    static_assert(NLD >= 1, "tile too small for this workgroup size");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NBUF = RING == 2 ? 3 : 2;
    char* const ldsK = smem;                 // [NBUF][TILEB]
    char* const ldsV = smem + NBUF * TILEB;  // [NBUF][TILEB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    FASN_STAMP(0);
    int bh, qi, split = 0;
    // Dynamic deal of the work items across XCDs (round 5, long plain / causal launches through fasn_fwd_ws): the XCDs
    // of a part differ by up to 4 % and the dispatcher deals each exactly 1/8 of the workgroups. With p.xq set the ITEMS are dealt dynamically
    // instead: one queue per XCD (its heads' query blocks - or block pairs - in the usual order, so an XCD still works on one head at a time), a
    // workgroup draws from the queue of the XCD it runs on and, when that is empty, from the next non-empty one; the grid carries a few workgroups
    // more than there are items and the surplus leaves at once. A workgroup still runs ONE item: no loop, no extra registers; results are the
    // static deal's bit for bit (an item's arithmetic does not depend on who runs it).
    auto draw_item = [&](const int items_per_head) -> bool {
        int* const slot = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            const int per = (p.B * p.H / 8) * items_per_head;
            const int x0 = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7);   // XCC_ID
            int item = -1, qx = 0;
            for (int k = 0; k < 8 && item < 0; ++k) {
                qx = (x0 + k) & 7;
                const int v = atomicAdd(&p.xq[qx], 1);
                if (v < per) item = v;
            }
            slot[0] = item;
            slot[1] = qx;
        }
        __syncthreads();
        const int item = slot[0], qx = slot[1];
        __syncthreads();
        if (item < 0) return false;
        bh = (item / items_per_head) * 8 + qx;
        qi = item % items_per_head;
        return true;
    };
    const int wgid = VH > 1 ? (int)(blockIdx.x / VH) : (int)blockIdx.x;
    const int dv0 = VH > 1 ? (int)(blockIdx.x % VH) * DB : 0;   // first output column block of this workgroup
    constexpr bool VEC = mode_is_vector(MODE);
    constexpr bool SLOW = MODE == MODE_GENERAL_SLOW;
    constexpr bool GEN = VEC || SLOW;
    constexpr bool VBIAS = mode_has_vbias(MODE);   // vector bias present (compile time)
    constexpr bool VMASK = mode_has_vmask(MODE);   // vector mask present (compile time)
    constexpr bool KPERM = VEC;
    // MODE_KEYPAD: the mask is one byte per key for the whole (b,h): each lane fetches the byte of key k0 + lane one tile ahead,
    // a ballot turns the 64 bytes into a wave-uniform bit word; tiles with all keys visible run as plain tiles, tiles with none
    // are skipped, only the boundary tiles start their hidden scores at -inf. Key-padded batches cost what unpadded ones do.
    constexpr bool KP = mode_has_keypad(MODE);
    constexpr bool PAIRABLE = (MODE == MODE_CAUSAL || (FASN_VEC_PAIR && mode_is_vector(MODE) && !mode_has_keypad(MODE))) && !SPLIT && VH == 1 && (!DROP || (FASN_DROP_PAIR && MODE == MODE_CAUSAL));   // (round 6: also the vector mask / bias modes under the causal flag)
    constexpr bool KPAIR = KP && VBIAS && !SPLIT && VH == 1 && !DROP;   // length-paired batch elements (see below)
    int bh2 = -1;   // KPAIR: the (b,h) of the second pass
    int kp_lead = 0;   // KPAIR: tile steps by which this workgroup starts its second pass before the group's last one (kpair_plan)
    if (SPLIT) {
        int blk;
        block_to_work(wgid, p.B * p.H, p.nqblk * p.nsplit, bh, blk);
        qi = blk / p.nsplit;
        split = blk % p.nsplit;
    } else if (GEN && p.batch_inner && (p.H & 7) == 0) {
        // per XCD: (head, q-block, batch) with the batch fastest -> the B workgroups that read the same bias tile run together
        const int xcd = wgid & 7, j = wgid >> 3;
        int bb = j % p.B, rest = j / p.B;
        if constexpr (KPAIR) {
            // Ragged key-padded batch under a batch-broadcast bias (C4): the B workgroups of a bias tile walk as many K/V tiles as
            // their batch element has visible keys, and the in-order dispatcher makes every round as long as its longest workgroup
            // (19 % of the C4 launch were idle CUs). Every workgroup reads the key-padding mask once (B x Sk bytes from L2), ranks
            // the batch elements by their walk length and takes TWO of them, the r-th longest and then the r-th shortest: all
            // workgroups of the launch cost about the same. The other half of the workgroup ids leaves at once (last ids = whole
            // rounds). Batches that are not ragged enough keep the plain schedule (one bias fetch serves B workgroups instead of two).
            int b0 = -1, b1 = -1;
            if (kpair_plan(p, smem, tid, j % ((p.B + 1) / 2), b0, b1, kp_lead)) {
                const int np = (p.B + 1) / 2;
                if (j >= (p.H >> 3) * p.nqblk * np) return;
                rest = j / np;
                bb = b0;
                if (b1 != b0) bh2 = b1 * p.H + (rest / p.nqblk) * 8 + xcd;
            }
        }
        const int nqb = (PAIRABLE && p.pair) ? (p.nqblk + 1) / 2 : p.nqblk;   // (causal flag next to a batch-broadcast bias: block pairs, see below)
        qi = rest % nqb;
        bh = bb * p.H + (rest / nqb) * 8 + xcd;
    } else if (PAIRABLE && p.pair) {
        block_to_work(wgid, p.B * p.H, (p.nqblk + 1) / 2, bh, qi);
        if (fwd_xq_kernel(D, MODE, SPLIT, VH, DROP) && p.xq != nullptr && !draw_item((p.nqblk + 1) / 2)) return;
    } else {
        block_to_work(wgid, p.B * p.H, p.nqblk, bh, qi);
        if (fwd_xq_kernel(D, MODE, SPLIT, VH, DROP) && p.xq != nullptr && !draw_item(p.nqblk)) return;
    }
    // Paired causal launch: the workgroup dispatcher hands workgroups out IN ORDER and waits for the CU whose turn it is (tools/
    // fasn_harness timeline: with 80..128-tile workgroups next to each other a CU idles until the longest of its round is done), so
    // unequal causal blocks leave 7 % of the workgroup slots empty even when sorted by weight. Block r and block nqblk-1-r together
    // always walk nqblk + 1 tiles: every workgroup of a paired launch costs the same. Used when the launch is many rounds long
    // (equal workgroups quantise the last round).
    const int npass = ((PAIRABLE && p.pair && qi != p.nqblk - 1 - qi) || (KPAIR && bh2 >= 0)) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
    if ((PAIRABLE || KPAIR) && pass) __syncthreads();
    if (KPAIR && pass) bh = bh2;
    // causal: heaviest (last) query blocks first
    const int qblk = (MODE != MODE_PLAIN && p.causal) ? (pass == 0 ? p.nqblk - 1 - qi : qi) : qi;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qblk * BM;
    const int qw0 = q0 + wave * (QB * 32);  // first row of this wave
    // first row of the wave's 32-row block qb: contiguous, or (FOLD) block `wave` and block `2 NW - 1 - wave` of the workgroup's 2 NW blocks
    auto rowb = [&](int qb) { return FOLD ? q0 + (qb == 0 ? wave : 2 * NW - 1 - wave) * 32 : qw0 + qb * 32; };

    const char* qbase = p.q + (b * p.qs[0] + h * p.qs[1]) * 2;
    const char* kbase = p.k + (b * p.ks[0] + (h / p.kvg) * p.ks[1]) * 2;
    const char* vbase = p.v + (b * p.vs[0] + (h / p.kvg) * p.vs[1]) * 2;

    const bool causal = (MODE == MODE_CAUSAL) || ((GEN || KP) && p.causal);
    const int coff = p.Sk - p.Sq;  // key j visible to row i iff j <= i + coff

    // ---- number of K/V tiles this workgroup walks
    int ntiles = (p.Sk + KT - 1) / KT;
    if (causal) {
        const int last_row = min(q0 + BM, p.Sq) - 1;
        const int kmax = last_row + coff;  // last visible key of the block
        const int nt_c = kmax < 0 ? 0 : (kmax / KT + 1);
        ntiles = min(ntiles, nt_c);
    }
    // Key-padding mask: the workgroup turns the mask bytes of its key range into one visibility word per K/V tile, ONCE, in LDS
    // (thread c: 16 bytes -> 16 bits; a tile then costs one uniform ds_read_b64 instead of a global load per wave and tile, whose
    // compiler-counted wait also drained the K/V prefetch). Trailing tiles without a visible key are not walked at all.
    constexpr int IMGB = BF32 ? 8192 : 4096;                          // bias image of one 32-row block and tile: [32 rows][64 keys] 16 bit / fp32
    constexpr int IMGM = mode_has_vmask(MODE) ? 2048 : 0;              // mask image [32 rows][64 bytes] (only the instantiations with a dense-mask operand reserve it)
    constexpr int BPC = BF32 ? 16 : 8;                                 // 16-byte chunks per bias image row
    constexpr int BW = BF32 ? 16 : 8;                                  // dwords a lane holds per 32-key block (16 keys)
    uint64_t* const ldsKP = reinterpret_cast<uint64_t*>(smem + 2 * (RING == 2 ? 3 : 2) * (KT * D * 2) + (mode_is_vector(MODE) ? NW * QB * (IMGB + IMGM) : 0));   // [kFwdKpMaxTiles]
    if (KP) {
        kp_build_words(ldsKP, p.mask == nullptr ? nullptr : p.mask + (b * p.ms[0] + h * p.ms[1]), p.Sk, ntiles, tid, NT);
        __syncthreads();
        int last = -1;
        for (int t = lane; t < ntiles; t += 64)
            if (ldsKP[t] != 0ull) last = t;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
        ntiles = min(ntiles, __builtin_amdgcn_readfirstlane(last) + 1);
    }
    const int t_begin = SPLIT ? split * p.tps : 0;   // tps is a multiple of 6: t & 1 and t % 3 select the LDS buffer as if t started at 0
    if (SPLIT) ntiles = min(ntiles, t_begin + p.tps);
    // KPAIR, second pass: step t works on tile (t + rot) mod ntiles, so that the workgroups of a (head, query block) group, which enter their
    // second passes `lead` steps apart, request the same bias tile at the same time (one fetch through the L2 instead of one per workgroup).
    // Steps past the end map to a tile behind the last key: their requests are out of range for the descriptors and fill zeros.
    int rot = 0;
    if (KPAIR && pass && p.kprot && ntiles > 0) rot = (ntiles - kp_lead % ntiles) % ntiles;
    const int nt_all = (p.Sk + KT - 1) / KT;
    auto phys = [&](int t) {
        if constexpr (!KPAIR) return t;
        else {
            const int u = t + rot;
            return t >= ntiles ? nt_all : (u >= ntiles ? u - ntiles : u);
        }
    };

    // ---- Q fragments (B operand: col = q = lane&31, k = 8*hi..8*hi+7 of each 16-wide step)
    vec8 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int row = rowb(qb) + l31;
        const bool ok = row < p.Sq;
        const char* rp = qbase + (int64_t)row * p.qs[2] * 2 + hi * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (ok) raw = gload16(rp + s * 32);
            __builtin_memcpy(&qf[qb][s], &raw, 16);
        }
    }

    // ---- staging: each thread moves NLD 16-byte chunks of K and of V per tile.
    // buffer loads: wave-uniform descriptor (per (b,h) base + Sk rows), per-thread byte offset fixed for the whole
    // kernel, tile offset in an SGPR -> no per-tile vector address arithmetic, and rows >= Sk read back as zeros.
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase), 0, p.kbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase), 0, p.vbytes, 0x00020000);
    unsigned kvoff[NLD], vvoff[NLD];
    int ldsoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int ci = tid + i * NT;
        int row = ci / CPR, ch = ci % CPR;
        if (RING == 2) ch ^= swz_f<D>(row);   // thread `tid` fills LDS slot ci of the image: fetch the chunk that belongs there
        // vector general modes: LDS row rho of a 32-row block holds key kperm(rho), so that the accumulator registers of a
        // lane (MFMA rows (r&3) + 8(r>>2) + 4hi) are the 16 CONSECUTIVE keys 16hi + r: bias / mask rows are then read
        // 32 / 16 contiguous bytes per lane. K and V use the same order, the PV contraction does not see it.
        const int grow = KPERM ? ((row & ~31) | (((row >> 2) & 1) << 4) | (((row >> 3) & 3) << 2) | (row & 3)) : row;
        kvoff[i] = (unsigned)(grow * (int)p.ks[2] * 2 + ch * 16);
        vvoff[i] = (unsigned)(grow * (int)p.vs[2] * 2 + ch * 16);
        ldsoff[i] = tile_off<D>(row, ch);
    }
    const int ktile_bytes = KT * (int)p.ks[2] * 2;
    const int vtile_bytes = KT * (int)p.vs[2] * 2;
    u32x4 stK[(RING == 1) + 1][NLD], stV[(RING == 1) + 1][NLD];
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, (RING == 1)>;
    const u32x4 krw = make_rsrc_words(kbase, p.kbytes), vrw = make_rsrc_words(vbase, p.vbytes);
    const uint32_t ldsK_w = lds_addr(ldsK) + wave * 1024, ldsV_w = lds_addr(ldsV) + wave * 1024;
    // XADDR (loops whose LDS buffer index is a run-time value, D >= 128): the swizzle is an XOR on the chunk index, so the address of
    // k-step s / feature block d is (address of step 0 / block 0) ^ (s << 5) / ^ (d << 6) as long as the row starts are multiples of
    // the row size (the dynamic LDS starts at 0 and every tile is a multiple of ROWB): ONE live address per operand and one v_xor
    // per distinct (s | d) instead of a precomputed offset register per step plus two VALU per read to add the buffer offset
    // (bias + key-padding kernel at D = 128: 228 -> 192 VGPRs, 63 VALU fewer per tile, C4 forward 4.26 -> 4.13 ms).
    constexpr bool XADDR = RING == 2 && !UNR3 && D >= 128 && FASN_XADDR;
    const uint32_t xk0 = lds_addr(ldsK) + l31 * ROWB + ((hi ^ swz_f<D>(l31)) << 4);
    const uint32_t xv0 = lds_addr(ldsV) + (4 * hi + ((lane & 15) >> 2)) * ROWB +
                         (((((lane >> 4) & 1) * 2 + ((lane & 3) >> 1)) ^ swz_f<D>(4 * hi + ((lane & 15) >> 2))) << 4) + (lane & 1) * 8;
    auto stage_direct = [&](int t, int buf) {   // RING 2: tile t -> LDS buffer buf, asynchronously (vmcnt)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            lds_dma16(krw, __builtin_amdgcn_readfirstlane(ldsK_w + buf * TILEB + i * NT * 16), kvoff[i], t * ktile_bytes);
            lds_dma16(vrw, __builtin_amdgcn_readfirstlane(ldsV_w + buf * TILEB + i * NT * 16), vvoff[i], t * vtile_bytes);
        }
    };
    auto stage_load = [&](int t, auto SET) {
        constexpr int S_ = RING == 1 ? decltype(SET)::value : 0;   // RING 0 / 2 pass their LDS buffer index here (one or no register set)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            stK[S_][i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff[i], t * ktile_bytes, 0);
            stV[S_][i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvoff[i], t * vtile_bytes, 0);
        }
    };
    auto stage_store = [&](int buf, auto SET) {
        constexpr int S_ = RING == 1 ? decltype(SET)::value : 0;   // RING 0 / 2 pass their LDS buffer index here (one or no register set)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            *LDS_PTR(u32x4, ldsK + buf * TILEB + ldsoff[i]) = stK[S_][i];
            *LDS_PTR(u32x4, ldsV + buf * TILEB + ldsoff[i]) = stV[S_][i];
        }
    };

    const DropSeed dsd = DROP ? drop_seed(p.seed_lo, p.seed_hi, p.rng) : DropSeed{0u, 0u};
    const DropThr dthr = drop_thr(DROP ? p.drop_thr : 1u);
    const uint32_t drop_rh = drop_rh_of(hi);   // alignbit amount of "rotate left by 16 hi"
    // ---- online-softmax state, per lane = per query row (log2 domain: y = x * log2(e))
    float m_run[QB], l_run[QB];
    f32x16 oacc[QB][DB];
    // SEED: -m (0 while m is still -inf) in all 16 registers of an accumulator-shaped tuple = the C operand of the first QK^T MFMA
    // of every key block; rewritten only when the exact path moves the max. `unseeded`: some row has no finite max yet.
    f32x16 mseed[(SEED && !VEC) ? QB : 1];   // the vector general modes build their start values per element from -m
    bool unseeded = !(p.n > 0.f && split == 0);
#pragma unroll
    for (int qb = 0; qb < ((SEED && !VEC) ? QB : 1); ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mseed[qb][r] = 0.f;
    const int hi_p = FOLD ? (fresh_lane_id() >> 5) : hi;
    float n_p = p.n;
    if constexpr (FOLD) asm volatile("" : "+s"(n_p));   // (per pass: not hoisted in front of the pass loop and parked in scratch)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const bool sink = n_p > 0.f && split == 0;
        m_run[qb] = sink ? 0.f : -INFINITY;
        l_run[qb] = (sink && hi_p == 0) ? n_p : 0.f;  // the two half-lanes' partial sums are added at the end
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
    }

    // ---- general mode addressing
    // VEC : per-(b,h) buffer descriptors (range = that head's [Sq x Sk] slice, so reads past Sk on the last row return 0
    //       instead of faulting; reads past Sk elsewhere return in-range garbage that the visibility select discards) and
    //       a per-lane byte offset of this lane's row + its 4*hi keys; 4 keys per load.
    // SLOW: per-lane row pointers, one element per load.
    u32x4 brw, mrw;
    unsigned bvo[QB][IMGB / 1024], mvo[QB][2];
    const char* bptr[QB];
    const uint8_t* mptr[QB];
    const bool has_bias = VBIAS || (SLOW && p.bias != nullptr);
    const bool has_mask = VMASK || (SLOW && p.mask != nullptr);
    if (GEN) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int rowc = min(qw0 + qb * 32 + l31, p.Sq - 1);
            if (VEC) {
                // slot s = 64*i + lane of the wave's LDS image <- the 16-byte chunk that belongs there (swizzled like the K/V tiles)
#pragma unroll
                for (int i = 0; i < IMGB / 1024; ++i) {
                    const int sl = i * 64 + lane, row = sl / BPC, c = (sl % BPC) ^ (BF32 ? swz_f<128>(row) : swz_f<64>(row));
                    bvo[qb][i] = (unsigned)((qw0 + qb * 32 + row) * (int)p.bs[2] * (BF32 ? 4 : 2) + c * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int sl = i * 64 + lane, row = sl >> 2, c = (sl & 3) ^ swz_f<32>(row);
                    mvo[qb][i] = (unsigned)((qw0 + qb * 32 + row) * (int)p.ms[2] + c * 16);
                }
            } else {
                const int esz = p.bias_f32 ? 4 : 2;
                bptr[qb] = has_bias ? p.bias + (b * p.bs[0] + h * p.bs[1] + (int64_t)rowc * p.bs[2] + 4 * hi * p.bs[3]) * esz : nullptr;
                mptr[qb] = has_mask ? p.mask + (b * p.ms[0] + h * p.ms[1] + (int64_t)rowc * p.ms[2] + 4 * hi * p.ms[3]) : nullptr;
            }
        }
        if (VEC) {
            const char* bb = has_bias ? p.bias + (b * p.bs[0] + h * p.bs[1]) * (BF32 ? 4 : 2) : p.q;
            const char* mb = has_mask ? reinterpret_cast<const char*>(p.mask) + (b * p.ms[0] + h * p.ms[1]) : p.q;
            brw = make_rsrc_words(bb, has_bias ? p.bias_bytes : 0u);
            mrw = make_rsrc_words(mb, has_mask ? p.mask_bytes : 0u);
        }
    }
    // VEC: bias / mask of the wave's own 32*QB rows x 64 keys go HBM -> LDS with coalesced 16-byte `buffer_load ... lds`
    // (8 / 4 lanes cover one row's 128 / 64 bytes; a lane-per-row register load would touch 32 cache lines per instruction
    // and made the texture addresser the bottleneck), into a wave-private image next to the K/V tiles; the lane then reads
    // its row's 32 + 16 bytes per 32-key block with ds_read_b128. Single-buffered: tile t+1 is requested right after
    // tile t has been read into registers, and lands while tile t is computed.
    char* const ldsGB = smem + 2 * NBUF * TILEB + wave * (QB * (IMGB + IMGM));   // [QB][32 rows][128 / 256 B] bias
    char* const ldsGM = ldsGB + QB * IMGB;                                         // [QB][32 rows][64 B] mask
    const uint32_t ldsGB_a = lds_addr(ldsGB), ldsGM_a = lds_addr(ldsGM);
    auto gen_dma = [&](int t) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (VBIAS) {
#pragma unroll
                for (int i = 0; i < IMGB / 1024; ++i) lds_dma16(brw, __builtin_amdgcn_readfirstlane(ldsGB_a + qb * IMGB + i * 1024), bvo[qb][i], t * (KT * (BF32 ? 4 : 2)));
            }
            if (VMASK) {
#pragma unroll
                for (int i = 0; i < 2; ++i) lds_dma16(mrw, __builtin_amdgcn_readfirstlane(ldsGM_a + qb * 2048 + i * 1024), mvo[qb][i], t * KT);
            }
        }
    };

    // VEC, bias present: the bias is folded into the QK^T accumulator INITIAL value (S' = bias*log2e/c + q.k, y = c*S'),
    // so the softmax below is the plain one and no bias register outlives the MFMAs. A lane's pieces for the NEXT tile
    // are loaded while the PV MFMAs of the current tile run.
    // MODE_GENERAL launched without a mask (bias only, where the bias-only instantiation is the worse kernel): every byte reads as set
    const uint32_t nomask = (VMASK && p.mask == nullptr) ? 0x01010101u : 0u;
    constexpr bool bias_fold = VBIAS;
    const float binv = bias_fold ? (SEED ? kLog2e : kLog2e / p.c) : 0.f;   // SEED: Q is pre-scaled, S' = bias*log2e - m + q'.k
    if (RING == 2) {
        if (ntiles > t_begin) {
            if (VEC) gen_dma(phys(t_begin));
            stage_direct(phys(t_begin), 0);
            stage_direct(phys(t_begin + 1), 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");   // tile 0 (and Q) landed; tile 1 in flight
        }
    } else if (ntiles > t_begin) {
        stage_load(phys(t_begin), Set0{});   // (phys: the rotated second walk of a length pair, identity elsewhere)
        stage_store(0, Set0{});
        if (RING) stage_load(phys(1), Set1{});   // tile 1 in flight in the second set
        if (VEC) gen_dma(phys(t_begin));
    }
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int s = 0; s < KS; ++s) retire_loads(qf[qb][s]);
    // SEED: Q is multiplied by c = scale*log2e once, here (rounded to the operand type like core/flash_attn.py:81-83 rounds
    // its pre-scaled q), and the QK^T accumulator starts at -m: the MFMAs deliver y - m and the per-score fma disappears.
    if (SEED) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                uint16_t h[8];
                __builtin_memcpy(h, &qf[qb][s], 16);
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = E::to_f32(h[e]) * p.c;
                qf[qb][s] = E::cvt8(f);
            }
    }
    // static priority for the second-dispatched half of an 8-wave workgroup (waves w and w+4 share a SIMD): the two
    // co-resident waves stop running their matrix / exponential phases in lock step
    if (PRIO && NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(PRIO);
    // 4-wave workgroups, two per CU: PRIO 2 = static priority for every other workgroup of a CU's first fill (the two waves
    // that share a SIMD stop contending symmetrically, one runs ahead and their matrix / exponential phases interleave);
    // PRIO 3 = raised priority while a wave issues its QK^T MFMAs
    if (PRIO == 2 && NW == 4 && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_setprio(1);

    FASN_STAMP(1);
    // rows of this wave: [qw0, qw0 + QB*32)
    const int wave_first_vis = qw0 + coff;                 // last visible key of the wave's first row
    const int wave_last_vis = qw0 + QB * 32 - 1 + coff;    // last visible key of the wave's last row

    // this tile's visibility word is read (uniform address) while the tile before is computed
    uint64_t kp_next = 0;
    if (KP && t_begin < ntiles) kp_next = ldsKP[phys(t_begin)];

    // Direct-to-LDS vector kernels (LATE): the image of a tile is moved LDS -> registers at the END of the tile before, between
    // the wait that precedes the barrier and the barrier itself, and the image after it is requested there: the LDS latency
    // and the request instructions then sit where the wave waits for its neighbours anyway instead of in front of the tile's
    // first MFMA (both waves of a SIMD belong to one workgroup and run in phase, so nothing else covered them there).
    constexpr bool LATE = VEC && RING == 2;
    // (Round 5, measured and not kept - profiles/r05_c4_forward_seeds_beside_pv_ab.log: the start values of tile t+1 built beside the PV MFMAs
    // of tile t instead of at the top of tile t+1. Config 4 forward 4.05 ms against 4.00 ms: with two waves per SIMD the other wave's MFMAs
    // already cover this wave's VALU burst behind the barrier, and the same VALU work between the wave's OWN MFMAs only lengthens them.)
    uint32_t mraw_c[QB][2][4];
    uint32_t braw_c[QB][2][BW];   // the lane's 16 keys of a 32-key block: 8 dwords of 16-bit pairs, or 16 fp32 values
    auto image_to_regs = [&](uint32_t (&mr)[QB][2][4], uint32_t (&br)[QB][2][BW]) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (VBIAS) {
#pragma unroll
                    for (int j = 0; j < BW / 4; ++j) {
                        const u32x4 w = *LDS_PTR(const u32x4, ldsGB + qb * IMGB + (BF32 ? tile_off<128>(l31, kb * 8 + 4 * hi + j) : tile_off<64>(l31, kb * 4 + 2 * hi + j)));
#pragma unroll
                        for (int e = 0; e < 4; ++e) br[qb][kb][4 * j + e] = w[e];
                    }
                }
                if (VMASK) {
                    const u32x4 w = *LDS_PTR(const u32x4, ldsGM + qb * 2048 + tile_off<32>(l31, kb * 2 + hi));
#pragma unroll
                    for (int g = 0; g < 4; ++g) mr[qb][kb][g] = w[g];
                }
            }
    };
    if (LATE && ntiles > t_begin) {   // (the prologue's wait and barrier above published the first image)
        image_to_regs(mraw_c, braw_c);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        gen_dma(phys(t_begin + 1));
    }

    // one K/V tile; LSET = register set that receives the prefetch issued here, SSET = set written to LDS at the end
    // DIAG_ (FOLD kernels): the instantiation for the tiles that touch the block's diagonal - run-time LDS buffer index, per-32-row-block
    // classification; the main walk (DIAG_ false) of a FOLD kernel sees fully visible tiles only and tests nothing
    const int lane_o = lane, l31_o = l31, hi_o = hi;   // (the names `compute` shadows)
    auto tile_body = [&](const int t, auto LSET, auto SSET, auto DIAG_) {
        constexpr bool DIAG = decltype(DIAG_)::value;
        if constexpr (PRIO == 4 && NW == 4) {
            // PRIO 4 (round 5, launches of ONE round): the wave's priority falls with its progress through the key walk, so the two workgroups
            // that share a CU stay within a quarter of the walk of each other. With the arbiter's oldest-first rule alone the first-dispatched
            // workgroup of a CU runs ahead and finishes at ~70 % of the span; its partner then runs the last 30 % alone, at half the CU's
            // throughput (tools/fasn_harness timeline, config 2). Launches of several rounds WANT that stagger (a fresh workgroup's prologue
            // overlaps the older one's loop) and keep PRIO 0.
            const int done4 = (t - t_begin) * 4, span = ntiles - t_begin;
            if (done4 < span) __builtin_amdgcn_s_setprio(3);
            else if (done4 < 2 * span) __builtin_amdgcn_s_setprio(2);
            else if (done4 < 3 * span) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // RING 1: the loop is unrolled by two (even tile: LSET = set 0, odd tile: LSET = set 1), so the LDS buffer index is a
        // compile-time constant there and buf*TILEB folds into the ds_read immediate offsets instead of two VALU per read
        // RING 2: the loop is unrolled by three and LSET carries the tile's LDS buffer (t % 3) as a compile-time constant, so the
        // buffer offset folds into the ds_read immediates and the DMA's M0 values instead of two VALU per LDS address
        // (D <= 64 and the plain 8-wave D = 128 kernel; the D = 128 mask / bias and 4-wave kernels measured 1-2 % slower with the
        // tripled loop body, instruction cache)
        const int buf = ((UNR3 || UNR2) && !DIAG) ? decltype(LSET)::value : (RING == 2 ? t % 3 : ((RING == 1 && QB == 1) ? decltype(LSET)::value : (t & 1)));
        const int buf2 = (UNR3 && !DIAG) ? (decltype(LSET)::value + 2) % 3 : (t + 2) % 3;   // RING 2: buffer of the tile requested now
        const int k0 = phys(t) * KT;
        if (RING == 2 && !VEC) stage_direct(phys(t + 2), buf2);   // past-the-end tiles are out of range for the descriptor
        else if (!VEC && (RING || t + 1 < ntiles)) stage_load(t + 1 + RING, LSET);

        uint64_t kp_bits = ~0ull;
        if (KP) {   // (the builtin returns a signed int)
            kp_bits = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(kp_next >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)kp_next);
            kp_next = ldsKP[phys(min(t + 1, ntiles - 1))];
        }
        // wave-uniform tile classification
        bool skip = false;       // no visible element for this wave
        bool need_mask = false;  // some element needs the element-wise path
        if (causal && !FOLD) {
            skip = k0 > wave_last_vis;
            need_mask = (k0 + KT - 1) > wave_first_vis;
        }
        if (k0 + KT > p.Sk && !(FOLD && !DIAG)) need_mask = true;   // (a FOLD kernel's main walk ends in front of the diagonal: every key below Sk)
        // per 32-row block (FOLD, diagonal tiles): hidden altogether / needs the element-wise path
        bool skipq[QB], maskq[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            skipq[qb] = false;
            maskq[qb] = need_mask;
            if (FOLD && DIAG) {
                const int first_vis = rowb(qb) + coff;   // last visible key of the block's first row
                skipq[qb] = k0 > first_vis + 31;
                maskq[qb] = need_mask || (k0 + KT - 1) > first_vis;
            }
        }
        if (KP && kp_bits == 0) skip = true;
        // short query blocks (decode shapes): a wave without rows only helps staging the tiles (QB = 1 kernels: the QB = 2
        // ones are only dispatched for Sq >= 256 and keep their register allocation)
        if ((QB == 1 || SPLIT) && qw0 >= p.Sq) skip = true;

        // VEC: mask bytes of this lane's elements, 4 consecutive keys per load; SLOW: every tile takes the exact path
        uint32_t mraw[QB][2][4];
        uint32_t braw[QB][2][BW];
        if (SLOW) {
            need_mask = true;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) maskq[qb] = true;
        }
        if (LATE) {   // the image is already in registers (end of the previous tile); only the requests of this tile remain
            stage_direct(phys(t + 2), buf2);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int g = 0; g < BW; ++g) braw[qb][kb][g] = braw_c[qb][kb][g];
#pragma unroll
                    for (int g = 0; g < 4; ++g) mraw[qb][kb][g] = mraw_c[qb][kb][g];
                }
        } else if (VEC) {   // unconditional (also for skipped tiles): the request / wait pattern stays the same for every tile
            // this tile's bias / mask image has landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            image_to_regs(mraw, braw);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is in registers before the next one is requested
            gen_dma(phys(t + 1));                                  // past-the-end tiles are out of range: zeros
            stage_load(phys(t + 1 + RING), LSET);
        }
        // the tile for the row blocks QLO .. QHI-1 of the wave (all of them, except in the diagonal tiles of a FOLD kernel: one block at a time
        // there - half the score registers, so that the second instantiation of the body does not push the main walk's registers out)
        auto compute = [&](auto QLO_, auto QHI_) {
            constexpr int QLO = decltype(QLO_)::value, QHI = decltype(QHI_)::value;
            // FOLD, diagonal tiles: everything lane-dependent in here (LDS addresses at a run-time buffer index, row limits) derives from a
            // fresh lane id, so that none of it is computed in front of the pass loop and parked in scratch across the main walk
            const int lane_c = (FOLD && DIAG) ? fresh_lane_id() : lane_o;
            const int lane = lane_c, l31 = (FOLD && DIAG) ? (lane_c & 31) : l31_o, hi = (FOLD && DIAG) ? (lane_c >> 5) : hi_o;
            const char* tK = ldsK + buf * TILEB;
            const char* tV = ldsV + buf * TILEB;

            // ---- S^T = K Q^T : acc[qb][kb], 32 keys x 32 queries each
            f32x16 sacc[QB][2];
            // (instantiated once per start-value variant below: the common variant then feeds the -m tuple itself to the first MFMA
            // of every key block as an untied C operand instead of merging with the boundary-tile variant through copies)
            auto qk_gemm = [&]() {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        vec8 kf;
                        if (XADDR) {
                            const u32x4 raw = *LDS_PTR(const u32x4, (uint32_t)(((xk0 + buf * TILEB) ^ (uint32_t)(s << 5)) + kb * 32 * ROWB));
                            __builtin_memcpy(&kf, &raw, 16);
                        } else {
                            kf = lds_read_rowfrag<E, D>(tK, kb * 32 + l31, s, hi);
                        }
#pragma unroll
                        for (int qb = QLO; qb < QHI; ++qb) sacc[qb][kb] = E::mfma(kf, qf[qb][s], sacc[qb][kb]);
                    }
                }
            };
            if (VEC) {
                // S' starts from the additive term: bias*log2e/c where the mask byte is set, -inf where it is clear
                // (S' = add + q.k, y = c*S'): from here on the tile is handled exactly like a plain one
#pragma unroll
                for (int qb = QLO; qb < QHI; ++qb) {
                    const float mneg = (SEED && m_run[qb] != -INFINITY) ? -m_run[qb] : 0.f;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = mneg;
                            if (VBIAS) {
                                if constexpr (BF32) {
                                    v = __builtin_fmaf(__uint_as_float(braw[qb][kb][r]), binv, mneg);
                                } else {
                                    const uint32_t w = braw[qb][kb][r >> 1];
                                    v = __builtin_fmaf(E::to_f32((uint16_t)((r & 1) ? (w >> 16) : (w & 0xffffu))), binv, mneg);
                                }
                            }
                            if (VMASK) v = (((mraw[qb][kb][r >> 2] | nomask) >> (8 * (r & 3))) & 0xffu) ? v : -INFINITY;
                            sacc[qb][kb][r] = v;
                        }
                }
                if (KP && kp_bits != ~0ull) {   // boundary tile of the key-padding mask (wave-uniform): hidden keys start at -inf
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const uint32_t w = (uint32_t)(kp_bits >> (32 * kb)) >> (16 * hi);   // key-permuted rows: register r = key 16*hi + r
#pragma unroll
                        for (int r = 0; r < 16; ++r)
#pragma unroll
                            for (int qb = QLO; qb < QHI; ++qb) sacc[qb][kb][r] = ((w >> r) & 1u) ? sacc[qb][kb][r] : -INFINITY;
                    }
                }
                qk_gemm();
            } else if (KP && kp_bits != ~0ull) {   // boundary tile of a key-padding mask: hidden keys start at -inf
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const uint32_t w = (uint32_t)(kp_bits >> (32 * kb)) >> (4 * hi);   // bit (r&3) + 8(r>>2) = key of register r
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool vis_r = ((w >> ((r & 3) + 8 * (r >> 2))) & 1u) != 0;
#pragma unroll
                        for (int qb = QLO; qb < QHI; ++qb) sacc[qb][kb][r] = vis_r ? (SEED ? mseed[qb][r] : 0.f) : -INFINITY;
                    }
                }
                qk_gemm();
            } else {
#pragma unroll
                for (int qb = QLO; qb < QHI; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[qb][kb][r] = SEED ? mseed[qb][r] : 0.f;
                qk_gemm();
            }
            // keep the -m tuple visibly alive past the QK^T MFMAs: the compiler then uses it as an UNTIED C operand for every key
            // block's first MFMA instead of copying it into the second block's accumulator (16 v_mov_b64 per 64-row tile)
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass cannot take a 16-register tuple as an asm operand)
            if (SEED && !VEC && FASN_SEED_KEEPALIVE) {
#pragma unroll
                for (int qb = QLO; qb < QHI; ++qb) asm volatile("" ::"v"(mseed[qb]));
            }
#endif
            if (PRIO == 3 && NW == 4) __builtin_amdgcn_s_setprio(0);

            // dropout of the 16 packed weights of (qb, kb), applied to the PACKED pairs (stream definition 2, fasn_common.h: DropBlock):
            // three packed 16-bit instructions per pair instead of a compare and a select per weight
            vec8 pf[QB][2][2];  // [qb][kb][t]: B operand of the PV MFMA
            auto drop_pack = [&](int qb, int kb) {
                const DropBlock<KPERM> db(drop_row_base(dsd.lo, (uint32_t)bh, (uint32_t)(qw0 + qb * 32 + l31)), dsd.hi, (uint32_t)((k0 + kb * 32) >> 4), hi, drop_rh);
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    uint32_t w[4];
                    __builtin_memcpy(w, &pf[qb][kb][t2], 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] &= db.keep_mask_pk(4 * t2 + e, dthr);   // dword e of t2 = registers 8 t2 + 2 e, + 1
                    __builtin_memcpy(&pf[qb][kb][t2], w, 16);
                }
            };

            // ---- online softmax_n per query block
            // Fast path (no hidden element in the tile): exponentiate against the CURRENT running max without computing
            // the tile max first; the result is exact as long as nothing overflows, which the row sum itself reveals
            // (any p > 2^8, an unset max (-inf) or a NaN makes the lane's partial sum exceed kSumLimit / compare false).
            // Only then - wave-uniformly - is the tile redone on the exact path, which re-centres the max.
#pragma unroll
            for (int qb = QLO; qb < QHI; ++qb) {
                bool exact = (FOLD ? maskq[qb] : need_mask) || (SEED && unseeded);   // (non-FOLD kernels: the wave-level flag itself - through the per-block array the split-K mask / bias kernel lost the uniform branch and 100 registers)
                if (!exact) {
                    float rs = 0.f;
                    const float mneg = -m_run[qb];
                    // two elements per VALU instruction where the ISA has packed fp32 (v_pk_fma_f32 for s*c - m, v_pk_add_f32
                    // for the row sum: two independent partial sums); the exponential and the mask multiply stay scalar
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 rs2 = {0.f, 0.f};
                    const f32x2 c2 = {p.c, p.c}, m2 = {mneg, mneg};
                    auto fast = [&](auto HM) {
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int t2 = 0; t2 < 2; ++t2) {
                                f32x8 x;
#pragma unroll
                                for (int e = 0; e < 8; e += 2) {
                                    const int r = 8 * t2 + e;
                                    const f32x2 s2 = {sacc[qb][kb][r], sacc[qb][kb][r + 1]};
                                    const f32x2 t = SEED ? s2 : __builtin_elementwise_fma(s2, c2, m2);
                                    const f32x2 pv = {fast_exp2(t[0]), fast_exp2(t[1])};
                                    x[e] = pv[0];
                                    x[e + 1] = pv[1];
                                    if (!PSUM) rs2 += pv;
                                }
                                pf[qb][kb][t2] = E::cvt8(x);
                                if (PSUM) {   // row sum of the ROUNDED weights (what the PV MFMA multiplies), two per instruction
                                    uint32_t w[4];
                                    __builtin_memcpy(w, &pf[qb][kb][t2], 16);
                                    rs2[0] = E::pair_sum(w[0], rs2[0]);
                                    rs2[1] = E::pair_sum(w[1], rs2[1]);
                                    rs2[0] = E::pair_sum(w[2], rs2[0]);
                                    rs2[1] = E::pair_sum(w[3], rs2[1]);
                                }
                            }
                        rs = rs2[0] + rs2[1];
                    };
                    using T_ = std::true_type;
                    using F_ = std::false_type;
                    fast(F_{});
                    if (__any(!(rs <= kSumLimit))) exact = true;
                    else l_run[qb] += rs;
                }
                if (exact) {
                    const int row = rowb(qb) + l31;   // (FOLD, diagonal tiles: l31 is the fresh copy of `compute`)
                    const int vis = causal ? (row + coff) : 0x7fffffff;  // last visible key of this row
                    float mx = -INFINITY;
                    if (!SLOW) {
                        // plain / causal / vector general (mask already folded into S'): hide by key range and the causal limit
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int key = k0 + kb * 32 + (KPERM ? 16 * hi + r : (r & 3) + 8 * (r >> 2) + 4 * hi);
                                const bool show = (key < p.Sk) && (key <= vis);
                                const float y = show ? (SEED ? sacc[qb][kb][r] : sacc[qb][kb][r] * p.c) : -INFINITY;   // SEED: already y - mref
                                sacc[qb][kb][r] = y;
                                mx = fmaxf(mx, y);
                            }
                    } else {
                    // (1) y = s*c, plus the additive bias (tile-uniform choice of load path: no per-element branching)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[qb][kb][r] *= p.c;
                    if (SLOW && has_bias) {
                        // fp32 / unaligned / strided bias: element loads, predicated on the key range
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int kofs = k0 + kb * 32 + (r & 3) + 8 * (r >> 2);  // + 4*hi is in bptr
                                float bv = 0.f;
                                if (kofs + 4 * hi < p.Sk) {
                                    if (p.bias_f32) bv = reinterpret_cast<const float*>(bptr[qb])[(int64_t)kofs * p.bs[3]];
                                    else bv = E::to_f32(reinterpret_cast<const uint16_t*>(bptr[qb])[(int64_t)kofs * p.bs[3]]);
                                }
                                sacc[qb][kb][r] = __builtin_fmaf(bv, kLog2e, sacc[qb][kb][r]);
                            }
                    }
                    // (2) visibility bits of this lane's 32 elements: key range, causal limit, boolean mask
                    uint32_t showbits = 0u;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            if ((key < p.Sk) && (key <= vis)) showbits |= 1u << (kb * 16 + r);
                        }
                    if (SLOW && has_mask) {
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int kofs = k0 + kb * 32 + (r & 3) + 8 * (r >> 2);
                                if (kofs + 4 * hi < p.Sk) {
                                    if (mptr[qb][(int64_t)kofs * p.ms[3]] == 0) showbits &= ~(1u << (kb * 16 + r));
                                }
                            }
                    }
                    // (3) hide, row max
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float y = ((showbits >> (kb * 16 + r)) & 1u) ? sacc[qb][kb][r] : -INFINITY;
                            sacc[qb][kb][r] = y;
                            mx = fmaxf(mx, y);
                        }
                    }
                    mx = max_across_halves(mx);
                    const float mref = (SEED && m_run[qb] != -INFINITY) ? m_run[qb] : 0.f;   // SEED: what the scores are relative to
                    const float m_new = fmaxf(m_run[qb], mx + mref);
                    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // fully hidden so far
                    const float alpha = fast_exp2(m_run[qb] - m_use);
                    const float m_sub = m_use - mref;
                    float rs = 0.f;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            f32x8 x;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                x[e] = fast_exp2(sacc[qb][kb][8 * t2 + e] - m_sub);
                                rs += x[e];
                            }
                            pf[qb][kb][t2] = E::cvt8(x);
                        }
                    l_run[qb] = l_run[qb] * alpha + rs;
                    m_run[qb] = m_new;
                    if (SEED && !VEC) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) mseed[qb][r] = -m_use;
                    }
                    if (!__all(alpha == 1.0f)) {
#pragma unroll
                        for (int d = 0; d < DB; ++d)
#pragma unroll
                            for (int r = 0; r < 16; ++r) oacc[qb][d][r] *= alpha;
                    }
                }
            }
            if constexpr (DROP != 0) {   // the row sums above are those of the undropped weights (LSE is dropout-free)
#pragma unroll
                for (int qb = QLO; qb < QHI; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) drop_pack(qb, kb);
            }
            if (SEED && unseeded) {   // rows that saw only hidden keys so far keep the wave on the exact path
                bool u = false;
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) u |= (m_run[qb] == -INFINITY);
                unseeded = __any(u);
            }

            // ---- O^T += V^T P^T
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int d = 0; d < DB; ++d) {
                        vec8 vf;
                        if (XADDR) {
                            const uint32_t xa = (xv0 + buf * TILEB) ^ (uint32_t)((dv0 + d) << 6);
                            const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, (uint32_t)(xa + (kb * 32 + 16 * t2) * ROWB)));
                            const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, (uint32_t)((xa ^ 32u) + (kb * 32 + 16 * t2 + 8) * ROWB)));
                            const s16x8 ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
                            __builtin_memcpy(&vf, &ab, 16);
                        } else {
                            vf = lds_read_trfrag<E, D>(tV, kb * 32 + 16 * t2, dv0 + d, lane);
                        }
#pragma unroll
                        for (int qb = QLO; qb < QHI; ++qb) oacc[qb][d] = E::mfma(vf, pf[qb][kb][t2], oacc[qb][d]);
                    }
            }
        };
        if constexpr (FOLD && DIAG) {
            if (!skipq[0]) compute(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            if (!skipq[QB - 1]) compute(std::integral_constant<int, QB - 1>{}, std::integral_constant<int, QB>{});
        } else {
            if (!skip) compute(std::integral_constant<int, 0>{}, std::integral_constant<int, QB>{});
        }

        if (RING == 2) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");   // tile t+1 (and its image) is in LDS, tile t+2 still in flight
            if (LATE) {   // next tile's image -> registers, the one after it requested (ordered by an explicit lgkmcnt wait: a DMA landing before a queued read would be silent corruption)
                image_to_regs(mraw_c, braw_c);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is in registers before its slot is requested again
                gen_dma(phys(t + 2));
            }
            __syncthreads();
        } else {
            if (VEC || RING || t + 1 < ntiles) stage_store(buf ^ 1, SSET);
            __syncthreads();
        }
    };
    if (RING == 1) {
        for (int t = 0; t < ntiles; t += 2) {
            tile_body(t, Set0{}, Set1{}, std::false_type{});       // even tile: tile t+1 sits in set 1, tile t+2 goes to set 0
            if (t + 1 < ntiles) tile_body(t + 1, Set1{}, Set0{}, std::false_type{});
        }
    } else if (UNR3) {
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        using B2 = std::integral_constant<int, 2>;
        // FOLD: the tiles every row of the block sees (all keys <= the first row's limit) take the plain body, the rest - up to the block's
        // last visible key - the diagonal body; the LDS buffer of tile t is t % 3 in both
        const int nmain = FOLD ? max(0, min(ntiles, (q0 + coff + 1) / KT)) : ntiles;
        for (int t = t_begin; t < nmain; t += 3) {   // t_begin is a multiple of 3 (split-K: tps is a multiple of 6)
            tile_body(t, B0{}, B0{}, std::false_type{});
            if (t + 1 < nmain) tile_body(t + 1, B1{}, B1{}, std::false_type{});
            if (t + 2 < nmain) tile_body(t + 2, B2{}, B2{}, std::false_type{});
        }
        if constexpr (FOLD) {
            for (int t = nmain; t < ntiles; ++t) tile_body(t, B0{}, B0{}, std::true_type{});
        }
    } else if (UNR2) {
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        for (int t = t_begin; t < ntiles; t += 2) {   // t_begin is even
            tile_body(t, B0{}, B0{}, std::false_type{});
            if (t + 1 < ntiles) tile_body(t + 1, B1{}, B1{}, std::false_type{});
        }
    } else {
        for (int t = t_begin; t < ntiles; ++t) tile_body(t, Set0{}, Set0{}, std::false_type{});
    }
    // direct-to-LDS requests issued for tiles past the end must land before this workgroup's LDS can be handed to another one
    if (RING == 2 || VEC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FASN_STAMP(2);

    if (SPLIT) {   // ---- partial result of this key range: un-normalised accumulator + (m, l) per row
        float* po = p.part_o + ((int64_t)bh * p.nsplit + split) * p.Sq * D;
        float* pml = p.part_ml + ((int64_t)bh * p.nsplit + split) * p.Sq * 2;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int row = rowb(qb) + l31;
            const float l_tot = sum_across_halves(l_run[qb]);
            if (row < p.Sq) {
                if (hi == 0) {
                    pml[row * 2] = m_run[qb];
                    pml[row * 2 + 1] = l_tot;
                }
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = oacc[qb][d][4 * g + e];
                        *reinterpret_cast<f32x4*>(po + (int64_t)row * D + d * 32 + 8 * g + 4 * hi) = x;
                    }
            }
        }
        return;
    }
    // ---- epilogue: O = acc / l, LSE = ln2 * (m + log2 l)
    char* obase = p.o + (b * p.os[0] + h * p.os[1]) * 2;
    // FOLD: the lane's row and its output addresses are derived HERE from an opaque copy of the lane id - computed before the pass loop
    // (where the compiler hoists them: they are loop invariant) they would be live across both tile loops, and 13 registers went to scratch
    const int lane_e = FOLD ? fresh_lane_id() : lane;   // (v_mbcnt: not even the lane id has to stay live)
    const int l31e = FOLD ? (lane_e & 31) : l31, hie = FOLD ? (lane_e >> 5) : hi;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int row = rowb(qb) + l31e;
        const float l_tot = sum_across_halves(l_run[qb]);
        const float inv = l_tot > 0.f ? (DROP ? p.drop_scale : 1.0f) / l_tot : 0.f;
        if (row < p.Sq) {
            if (p.lse != nullptr && hie == 0 && dv0 == 0) {
                const float m_use = (m_run[qb] == -INFINITY) ? 0.f : m_run[qb];
                p.lse[(int64_t)bh * p.Sq + row] = l_tot > 0.f ? (m_use + __builtin_log2f(l_tot)) * kLn2 : -INFINITY;   // (row: from the opaque lane copy)
            }
            char* rp = obase + (int64_t)row * p.os[2] * 2;
            // 16-byte stores (round 5, store_block_wide in fasn_common.h: half the store instructions of the row-per-lane epilogue;
            // same box: M0 0.4953 -> 0.4872 ms, C2 0.0437 -> 0.0414, C3 0.3032 -> 0.2989, C5 2.321 -> 2.272; profiles/r05_wide_output_stores_ab.log)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                if constexpr (D == 32 && DROP) store_block_narrow<E>(rp + (dv0 + d) * 64, oacc[qb][d], inv, hie);   // (at their register limit)
                else store_block_wide<E>(rp + (dv0 + d) * 64, oacc[qb][d], inv, hie);
            }
        }
    }
#ifdef FASN_DEV_VARIANTS
    if (p.timeline != nullptr && threadIdx.x == 0) {
        p.timeline[(size_t)blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
        p.timeline[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        p.timeline[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
        p.timeline[(size_t)blockIdx.x * 8 + 6] = (pass ? p.timeline[(size_t)blockIdx.x * 8 + 6] : 0ull) + (unsigned long long)ntiles;
    }
#endif
    }   // pass
}

// Merge the split-K partials: m* = max_s m_s, l = sum_s l_s 2^(m_s - m*), O = sum_s acc_s 2^(m_s - m*) / l.
// One thread per (row, 4 features).
template <typename Tag, int D>
__global__ void __launch_bounds__(256) fasn_fwd_combine_kernel(const FwdParams p) {
    using E = ET<Tag>;
    constexpr int TPR = D / 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t rowg = gid / TPR;   // (bh, row)
    const int c4 = (int)(gid % TPR) * 4;
    if (rowg >= (int64_t)p.B * p.H * p.Sq) return;
    const int bh = (int)(rowg / p.Sq), row = (int)(rowg % p.Sq);
    const int b = bh / p.H, h = bh % p.H;
    float mstar = -INFINITY;
    for (int s = 0; s < p.nsplit; ++s) mstar = fmaxf(mstar, p.part_ml[(((int64_t)bh * p.nsplit + s) * p.Sq + row) * 2]);
    const float m_use = (mstar == -INFINITY) ? 0.f : mstar;
    float l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.nsplit; ++s) {
        const int64_t base = ((int64_t)bh * p.nsplit + s) * p.Sq + row;
        const float ms = p.part_ml[base * 2], ls = p.part_ml[base * 2 + 1];
        const float w = (ms == -INFINITY) ? 0.f : fast_exp2(ms - m_use);
        l += ls * w;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.part_o + base * D + c4);
        acc += a * w;
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    typename E::vec4 y = E::cvt4(acc * inv);
    u32x2 raw;
    __builtin_memcpy(&raw, &y, 8);
    gstore8(p.o + (b * p.os[0] + h * p.os[1] + (int64_t)row * p.os[2] + c4) * 2, raw);
    if (p.lse != nullptr && c4 == 0) p.lse[(int64_t)bh * p.Sq + row] = l > 0.f ? (m_use + __builtin_log2f(l)) * kLn2 : -INFINITY;
}

}  // namespace fasn
