// Attention propagation over the unified bipartite CSR (user rows then item rows), forward.
// Replaces graphattention/SPUIGACF.py:359-397 (see include/ngacf.h).
//
// A lane owns a 16-byte slice of a feature row, so LPR = d*sizeof(T)/16 lanes ("a slot") cover
// one neighbour row and a wave64 has EPI = 64/LPR slots.  Two traversal modes in ONE launch:
//
//   short rows (degree <= plan.short_max, the bulk): one slot per destination row, EPI rows
//     per wave, rows pre-sorted by degree so the slots of a wave finish together.  No
//     cross-lane reduction at all; prologue/epilogue amortised over EPI rows.
//   long rows (short_max < degree <= seg_len): one wave per row, slots stride over the edges, one cross-slot fold at the end.
//   hub rows (degree > seg_len): cut into seg_len-edge slices, one wave each; every slice leaves (sum m e H, sum e) partials in
//     its slot and the row's LAST-ARRIVING slice adds them in slot order and finishes the row inside this launch (ticket per row,
//     write-through stores, agent-scope atomic loads: DESIGN.md section 3.3).  Plans beyond 2^18 slices (or tickets == NULL) leave
//     the sums to prop_fwd_split_kernel.  This is the DEFAULT layout.
//   OPTIONAL, off by default (build_plan(hub_cap=None), measured slower: plan.py default_hub_cap): rows of 4*seg_len < degree
//     <= hub_cap as one WORKGROUP per row (WORK_HUB below), the four waves fold through LDS.
//
// The first version (wave per row everywhere) was VALU-issue bound at 540 instructions per
// 28-edge row (rocprof: profiles/r01_v0_*); this layout needs ~4 VALU per edge.
#include "ngacf_common.hpp"
#include "mfma_rows.hpp"

namespace ngacf {

struct PropArgs {
    const int4* segs;      // long mode work items
    int nseg;
    const int4* srec;      // short mode: {row, first edge, degree, 0} of the dst rows, sorted by degree, descending
    int nshort;
    const int* hrow;       // hub mode: one workgroup per row (users first, by degree descending)
    const int* rowptr;     // int32 [N+1]
    const int* col;
    const void* H;
    const float* S_dst;
    const float* S_src;
    const float* avec;     // [2][d]
    void* out;
    void* agg;             // optional [N][d] in the storage type: the normalised aggregate, for the backward
    float* R;
    float* part_acc;
    float* part_r;
    const int* split;      // {row, first part, parts} of the hub rows
    const int* part_split; // hub row (index into split) of every partial-sum slot
    int* tickets;          // arrival counter per hub row (all zero between launches); NULL = finish in a 2nd kernel
    const uint32_t* needed; // optional, one bit per row: only rows with needed[row] != 0 are computed (the others' out/agg/R are left untouched)
    const int* rows;       // with needed: the same rows as a list (duplicates allowed, -1 = skip); its short rows get a lane group each
    int short_max;         // with needed: degree bound of the plan's short rows
    int U, N;
    WorkOrder order;
    float alpha;
    uint32_t drop_thr;
    float drop_scale;
    uint64_t seed;
    const uint64_t* seed_dev;   // optional device-side addend (hipGraph replays advance it)
    int flags;
    // the NEXT stage's row transform in the epilogue (ngacf_prop_fwd_next; next_H == nullptr: none)
    const uint4* next_panel;    // [2][D/16][NEXT/32][64] bf16x8 A fragments of W' (ngacf_transform_fwd_pack, mfma_rows.hpp)
    const float* next_avec;     // [2][NEXT]
    void* next_H;               // [N][NEXT] bf16
    float* next_S;              // [N][next heads]
    uint32_t next_drop_thr;     // the next stage's feature dropout (0 = none)
    float next_drop_scale;
    uint64_t next_seed;
};

constexpr int XROW_U4 = 9;      // a staged row: 128 B of bf16 + 16 B pad (36-dword stride: the fragment reads of 16 rows hit 64 distinct banks)

template <typename T, int D, int HEADS>
struct Shape {
    static constexpr int VEC = Slice<T>::VEC;
    static constexpr int LPR = D / VEC;            // lanes per row (slot width)
    static constexpr int EPI = WAVE / LPR;         // slots per wave
    static constexpr int HW = D / HEADS;           // head width in columns
    static constexpr int LPH = HW >= VEC ? HW / VEC : 1;   // lanes per head
    static constexpr int ROW_SHIFT = D * sizeof(T) == 128 ? 7 : D * sizeof(T) == 256 ? 8 : 9;
    static_assert(D * sizeof(T) == (1u << ROW_SHIFT), "row bytes must be 128, 256 or 512");
    static_assert(LPR >= 8, "column-index broadcast uses lanes 0..7 of a slot");
};

#ifndef NGACF_PIPE_TRAIN
#define NGACF_PIPE_TRAIN 1
#endif
#ifndef NGACF_FWD_PIN
#define NGACF_FWD_PIN 1
#endif
#ifndef NGACF_FWD_PIN_TRAIN
#define NGACF_FWD_PIN_TRAIN 2
#endif
constexpr int UNROLL = 4;

// 16-byte slice `q` of row `c`: uniform base + 32-bit byte offset (table < 4 GiB, checked by the host entry)
template <typename T, int SHIFT>
__device__ __forceinline__ void load_slice(const void* base, int c, int q, float (&f)[Slice<T>::VEC]) {
    const uint32_t off = (static_cast<uint32_t>(c) << SHIFT) | (static_cast<uint32_t>(q) << 4);
    Slice<T>::load(reinterpret_cast<const T*>(static_cast<const char*>(base) + off), f);
}

// Measurement-only instantiations (ROOF != 0; NGACF_ROOF_ADD / NGACF_ROOF_XOR in `flags`, tools/gather_roof.py): the SAME work order, lane
// layout, index loads, row gathers, loop structure and stores as the kernel they are stripped from, with the per-edge arithmetic
// reduced to what keeps the loads live -- ROOF 1: one add per gathered element (bf16 rows are still unpacked); ROOF 2: one xor per
// gathered DWORD (no unpack: the memory system's own ceiling for this access pattern).  No score, exp, mask, division or ELU.  Their
// results mean nothing; their duration is the measured roof bench.py grades the traversal against.
template <typename T, int SHIFT, int ROOF>
__device__ __forceinline__ void gather_slice(const void* base, int c, int q, float (&f)[Slice<T>::VEC]) {
    if constexpr (ROOF == 2) {
        const uint32_t off = (static_cast<uint32_t>(c) << SHIFT) | (static_cast<uint32_t>(q) << 4);
        const float4 v = *reinterpret_cast<const float4*>(static_cast<const char*>(base) + off);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
#pragma unroll
        for (int j = 4; j < Slice<T>::VEC; ++j) f[j] = 0.f;
    } else {
        load_slice<T, SHIFT>(base, c, q, f);
    }
}

// ELU with a short series near 0 (expm1f costs ~25 VALU per value; this is ~9)
__device__ __forceinline__ float elu_fast(float x) {
    const float series = x * (1.f + x * (0.5f + x * (1.f / 6.f + x * (1.f / 24.f + x * (1.f / 120.f)))));
    const float em1 = x > -0.125f ? series : __expf(x) - 1.f;
    return x > 0.f ? x : em1;
}
// The value a bf16 table stores keeps 8 bits: exp(x) - 1 straight from v_exp_f32 is exact to ~1e-7 ABSOLUTE, three orders below
// the rounding of the store (the series above is there for fp32 storage's 1e-4 bar near 0).  4 VALU per value instead of 14: an
// 8-column lane saves what three edges cost.
template <typename T>
__device__ __forceinline__ float elu_store(float x) {
    if constexpr (sizeof(T) == 2) return x > 0.f ? x : __builtin_amdgcn_exp2f(x * 1.4426950408889634f) - 1.f;
    else return elu_fast(x);
}

// value of lane (slot*LPR + J) broadcast to every lane of the slot: ds_swizzle bit-mode, no LDS memory
template <int LPR, int J>
__device__ __forceinline__ int slot_bcast(int v) {
    constexpr int and_mask = (~(LPR - 1)) & 0x1f;
    return __builtin_amdgcn_ds_swizzle(v, (J << 5) | and_mask);
}

// The negative-side sums a train-mode forward keeps for the backward (SAVE instantiations; propagate_bwd.hip, "split-sign"
// form): over the edges whose score s = S[dst] + S[src] is not positive,
//     accn = sum m e H[src],   rneg = sum e          (agg_neg = accn / R is stored behind agg, rneg behind R)
// With them the half of dS[n] that multiplies the row's OWN gradient is row-local in the backward,
//     sum_src -lrelu'(s) e (m Gn[n].H[src] - t[n]) = (1 - alpha) (R Gn[n].agg_neg[n] - rneg t[n]),
// and the edge backward gathers one row (Gn[src]) and 8 bytes per head (S[src], t[src]) per edge instead of two rows and 4.
template <int VEC>
struct NegSums {
    float accn[VEC];
    float rneg;
};

// The raw 16 bytes of slice `q` of row `c` (the gather itself: nothing but the address and the load) and their unpacking.  The
// traversal loops keep TWO batches of UNROLL raw slices per lane: the next batch's gathers are issued before the current batch
// is unpacked and used, so a wave always has UNROLL..2 UNROLL row requests in flight (the r1-r4 loops issued UNROLL, waited,
// computed with none in flight, and relied on the other waves of the SIMD alone to cover the round trip).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // one 128-bit register tuple: pinned whole, it stays one
template <int SHIFT>
__device__ __forceinline__ u32x4 gather_raw(const void* base, int c, int q) {
    const uint32_t off = (static_cast<uint32_t>(c) << SHIFT) | (static_cast<uint32_t>(q) << 4);
    return *reinterpret_cast<const u32x4*>(static_cast<const char*>(base) + off);
}
template <typename T, int ROOF>
__device__ __forceinline__ void unpack_raw(const u32x4& v, float (&f)[Slice<T>::VEC]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    if constexpr (sizeof(T) == 4 || ROOF == 2) {           // fp32 rows; ROOF 2: the raw dwords (no unpack)
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = __uint_as_float(w[j]);
#pragma unroll
        for (int j = 4; j < Slice<T>::VEC; ++j) f[j] = 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j]     = __uint_as_float(w[j] << 16);
            f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
    }
}

// A finished row: the values (calc) and the stores (store).  The fused launch (NEXT) puts the workgroup's barrier BETWEEN them: a
// wave with stores in flight waits for their acknowledgement at the barrier (vmcnt counts stores too), and a train-mode forward has
// five of them per row (out, agg, aggn, R, Rn) -- 45 against 37 us per launch.
template <typename T, int D, int HEADS, bool SAVE, int ROOF = 0>
__device__ __forceinline__ void finish_row_calc(const PropArgs& a, int row, int q, int head, float (&acc)[Slice<T>::VEC], float rsum,
                                                NegSums<Slice<T>::VEC>& ng, float (&o)[Slice<T>::VEC]) {
    using SH = Shape<T, D, HEADS>;
    constexpr int VEC = SH::VEC;
    float own[VEC];
    // (requesting the row's own slice when the wave learns its row instead of here was measured: no difference -- the other waves
    //  of the SIMD cover this round trip -- and it holds four more VGPRs through the loop)
    load_slice<T, SH::ROW_SHIFT>(a.H, row, q, own);
    if constexpr (ROOF != 0) {                            // (measurement only: the row's own load and one add per element)
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = own[j] + acc[j];
        return;
    }
    const float inv = rsum > 0.f ? 1.f / rsum : 0.f;   // isolated node: aggregate is 0 (SPUIGACF.py:389)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        acc[j] *= inv;
        const float pre = (a.flags & NGACF_RESIDUAL) ? own[j] + acc[j] : acc[j];
        o[j] = (a.flags & NGACF_ELU) ? elu_store<T>(pre) : pre;
    }
    if constexpr (SAVE) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) ng.accn[j] *= inv;
    }
}

template <typename T, int D, int HEADS, bool SAVE>
__device__ __forceinline__ void finish_row_store(const PropArgs& a, int row, int q, int head, const float (&acc)[Slice<T>::VEC], float rsum,
                                                 const NegSums<Slice<T>::VEC>& ng, const float (&o)[Slice<T>::VEC]) {
    using SH = Shape<T, D, HEADS>;
    constexpr int VEC = SH::VEC;
    if (a.out) Slice<T>::store(static_cast<T*>(a.out) + (size_t)row * D + q * VEC, o);     // (NULL: only the next stage's operands are wanted)
    if constexpr (SAVE) {                                 // agg / R (both given: the launcher picks SAVE by them), two planes each
        Slice<T>::store(static_cast<T*>(a.agg) + (size_t)row * D + q * VEC, acc);            // (bf16 storage: 128 B per row instead of 256)
        Slice<T>::store(static_cast<T*>(a.agg) + ((size_t)a.N + row) * D + q * VEC, ng.accn);
        if ((q * VEC) % SH::HW == 0) {
            a.R[(size_t)row * HEADS + head] = rsum;
            a.R[((size_t)a.N + row) * HEADS + head] = ng.rneg;
        }
    }
}

template <typename T, int D, int HEADS, bool SAVE, int ROOF = 0>
__device__ __forceinline__ void finish_row(const PropArgs& a, int row, int q, int head, float (&acc)[Slice<T>::VEC], float rsum,
                                           NegSums<Slice<T>::VEC>& ng) {
    float o[Slice<T>::VEC];
    finish_row_calc<T, D, HEADS, SAVE, ROOF>(a, row, q, head, acc, rsum, ng, o);
    finish_row_store<T, D, HEADS, SAVE>(a, row, q, head, acc, rsum, ng, o);
}

// What a lane group hands to the fused tail (NEXT): the row it finished; the tail stores it, behind its barrier.
template <int VEC>
struct RowOut {
    float o[VEC], agg[VEC];
    NegSums<VEC> ng;
    float rsum;
    int row;
    bool valid;
};

template <typename T, int D, int HEADS, bool SAVE>
__device__ __forceinline__ void finish_row_defer(const PropArgs& a, int row, int q, int head, float (&acc)[Slice<T>::VEC], float rsum,
                                                 NegSums<Slice<T>::VEC>& ng, RowOut<Slice<T>::VEC>& r) {
    finish_row_calc<T, D, HEADS, SAVE>(a, row, q, head, acc, rsum, ng, r.o);
    if constexpr (SAVE) {
#pragma unroll
        for (int j = 0; j < Slice<T>::VEC; ++j) { r.agg[j] = acc[j]; r.ng.accn[j] = ng.accn[j]; }
        r.ng.rneg = ng.rneg;
        r.rsum = rsum;
    }
    r.row = row;
    r.valid = true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The next stage's transform of the rows THIS WORKGROUP has just finished (ngacf_prop_fwd_next): the tail of the fused kernel.
// Lane (slot, q) holds columns 8q..8q+7 of the row its lane group finished in `o` (valid: it did finish one: every group of a
// short-row wave, group 0 of a long-row wave).
//   1. every wave writes its rows as the next stage's bf16 operand (the values `out` stores, after the next stage's feature
//      dropout) into the workgroup's 32-row LDS tile (row 8 * wave + slot) -- and has meanwhile received its share of the
//      weight panel by LDS-DMA (issued at kernel start);
//   2. ONE barrier publishes tile and panel; waves 1..3 leave;
//   3. wave 0 runs exactly the tile step of transform_mfma.hip on it: B fragments = 16 bytes of a node's row per k-step, KS x OT
//      v_mfma_f32_32x32x16_bf16, mfma_fwd_epilogue (H' rows as 16-byte stores, S' from the rounded H').  Same arithmetic, same
//      order, same bits as the stand-alone launch on the stored table.
// ---------------------------------------------------------------------------------------------------------------------------
template <int D, int NEXT, bool NONE, bool STAGE_OUT, typename STORES>
__device__ __forceinline__ void next_transform_tail(const PropArgs& a, uint4* xtile, int* xmeta, const uint4* panel, const float* av,
                                                    bool valid, int row, int slot, int q, const float (&o)[8], STORES&& stores) {
    constexpr int KS = D / 16, OT = NEXT / 32, NH = NONE ? 1 : NEXT / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tr = 8 * wave + slot;
    if (valid) {
        uint32_t w[4];
        if (a.next_drop_thr) {
            const uint32_t salt = salt_of(a.next_seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u);
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[j] = Slice<bf16_t>::round(o[j]);            // the stored value, then the mask: what the stand-alone launch reads and does
                x[j] = feat_bits(salt, row, 8 * q + j) >= a.next_drop_thr ? x[j] * a.next_drop_scale : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = f32_to_bf16(x[2 * j]) | (f32_to_bf16(x[2 * j + 1]) << 16);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = f32_to_bf16(o[2 * j]) | (f32_to_bf16(o[2 * j + 1]) << 16);
        }
        xtile[tr * XROW_U4 + q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (q == 0) xmeta[tr] = valid ? row : -1;
    __syncthreads();                                          // (the compiler drains the LDS-DMA of the panel in front of it)
    stores();                                                 // the row's own stores (out / agg / R), behind the barrier
    if (wave != 0) return;
    const int r = lane & 31, h = lane >> 5;
    const int nrow = xmeta[r];
    const bool ok = nrow >= 0;
    bf16x8 bf[KS];
#pragma unroll
    for (int sk = 0; sk < KS; ++sk) bf[sk] = __builtin_bit_cast(bf16x8, xtile[r * XROW_U4 + 2 * sk + h]);
    f32x16 acc[OT];
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int sk = 0; sk < KS; ++sk)
#pragma unroll
        for (int t = 0; t < OT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, panel[(sk * OT + t) * 64 + lane]), bf[sk], acc[t], 0, 0, 0);
    // (H' leaves through the tile too -- every lane of this wave has its fragments, nobody else reads the tile any more)
    mfma_fwd_epilogue<OT, NH, NEXT, STAGE_OUT>(acc, ok ? nrow : 0, ok, h, av, a.next_H, a.next_S, xtile, r,
                                    [&](int tr) { return xmeta[tr]; });
}

// The same for fp32 storage (mfma_rows.hpp: f32_tile_step).  A lane group finishes a row of 64 fp32 columns with 16 lanes, so a
// wave finishes 4 rows and the workgroup 16: exactly the 16 node columns of v_mfma_f32_16x16x4_f32, the exact-fp32 matrix-core
// instruction (its accumulation is an fmaf chain, k ascending: the bits of the stand-alone launch, which runs the same step).
// ---------------------------------------------------------------------------------------------------------------------------
template <int D, int NEXT, bool NONE, typename STORES>
__device__ __forceinline__ void next_transform_tail_f32(const PropArgs& a, float* xtile, int* xmeta, const float* panel, const float* av,
                                                        float* sx, bool valid, int row, int slot, int q, const float (&o)[4], STORES&& stores) {
    static_assert(D == 64, "fp32 fused transform: 64 input columns");
    const int wave = threadIdx.x >> 6;
    const int tr = 4 * wave + slot;
    if (valid) {
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = o[j];
        if (a.next_drop_thr) {
            const uint32_t salt = salt_of(a.next_seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u);
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = feat_bits(salt, row, 4 * q + j) >= a.next_drop_thr ? x[j] * a.next_drop_scale : 0.f;
        }
        *reinterpret_cast<float4*>(xtile + tr * XROW_F32 + 4 * q) = make_float4(x[0], x[1], x[2], x[3]);
    }
    if (q == 0) xmeta[tr] = valid ? row : -1;
    __syncthreads();                                          // tile + panel (LDS-DMA drained in front of it) are in LDS
    stores();
    f32_tile_step<NEXT, NONE>(xtile, xmeta, panel, av, sx, a.next_H, a.next_S);
}

constexpr float NEG_LOG2E = -1.4426950408889634f;

// Ordering pins.  The compiler is free to unpack all four edges of a batch first and accumulate them last (32 VGPRs of unpacked
// columns live at once) and to hoist the next batch's gathers above the current batch's arithmetic; either pushes the raw slices
// of the batch in flight out to scratch, with an s_waitcnt vmcnt(0) in front of the spill.  A scheduling fence
// (__builtin_amdgcn_sched_barrier) does not bind pure arithmetic at instruction selection, so the order is pinned by DATA: an
// empty volatile asm that takes the running sums (and the next edge's raw slice) in and out -- volatile asms keep their order,
// an edge's arithmetic must be complete in front of its pin, the next edge's unpack can only start behind it -- and an empty
// volatile asm with a memory clobber at the phase boundaries, which no load crosses.
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(int& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(u32x4& v) { asm volatile("" : "+v"(v)); }
template <int N>
__device__ __forceinline__ void pin(float (&x)[N]) {
    if constexpr (N == 4) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    else if constexpr (N == 8) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    else {
#pragma unroll
        for (int j = 0; j < N; ++j) pin(x[j]);
    }
}
#define NGACF_PHASE_FENCE() asm volatile("" ::: "memory")
template <bool B> struct BoolC { static constexpr bool value = B; };

// The running sums of a lane while it walks its edges, as the register PAIRS v_pk_fma_f32 works on (pinned as pairs, they stay
// pairs): eval (acc[2j], acc[2j + 1]); SAVE (acc[j], accn[j]) -- both sums of a column advance in one packed FMA.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pin(f32x2& x) { asm volatile("" : "+v"(x)); }
template <int VEC, bool SAVE>
struct Sums {
    static constexpr int NP2 = SAVE ? VEC : VEC / 2;
    f32x2 a2[NP2];
    float rsum, rneg;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int j = 0; j < NP2; ++j) a2[j] = f32x2{0.f, 0.f};
        rsum = rneg = 0.f;
    }
    template <int ROOF>
    __device__ __forceinline__ void pin_all() {
#pragma unroll
        for (int j = 0; j < NP2; ++j) pin(a2[j]);
        if constexpr (ROOF == 0) pin(rsum);
        if constexpr (SAVE) pin(rneg);
    }
    __device__ __forceinline__ void get(float (&acc)[VEC], float& r, NegSums<VEC>& ng) const {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if constexpr (SAVE) { acc[j] = a2[j].x; ng.accn[j] = a2[j].y; }
            else acc[j] = (j & 1) ? a2[j / 2].y : a2[j / 2].x;
        }
        r = rsum;
        if constexpr (SAVE) ng.rneg = rneg;
    }
};

// one edge: score, exp(-leakyrelu), optional dropout, accumulate.
// The scores arrive PRE-SCALED by -log2(e) (sd = -log2e S[dst], av = -log2e a[src type]; a gathered S[src] is scaled here), so that
//     e = exp(-max(s, alpha s)) = exp2(min(s', alpha s')),   s' = -log2e s
// is one multiply, one min and the v_exp_f32 itself (exp() of an unscaled score spends a third multiply on the conversion), and the
// recomputed source score a . h runs as VEC / 2 packed FMAs (v_pk_fma_f32) that start from (sd, 0) where a lane holds a whole head.
// s > 0 (the backward's lrelu' = 1 side) is s' < 0.
template <typename T, int D, int HEADS, bool FROM_ROW, bool DROP, bool SAVE, int ROOF = 0>
__device__ __forceinline__ void edge_step(const PropArgs& a, const float (&hv)[Slice<T>::VEC], float ss, bool ok, int c,
                                          float sd, const float (&av)[Slice<T>::VEC], bool dst_is_user,
                                          uint32_t row_key, uint32_t salt_h, Sums<Slice<T>::VEC, SAVE>& sm) {
    using SH = Shape<T, D, HEADS>;
    constexpr int VEC = SH::VEC;
    if constexpr (ROOF == 1) {                            // one add per gathered element
#pragma unroll
        for (int j = 0; j < VEC / 2; ++j) sm.a2[j] += f32x2{hv[2 * j], hv[2 * j + 1]};
        return;
    } else if constexpr (ROOF == 2) {                     // one xor per gathered dword
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            sm.a2[j].x = __uint_as_float(__float_as_uint(sm.a2[j].x) ^ __float_as_uint(hv[2 * j]));
            sm.a2[j].y = __uint_as_float(__float_as_uint(sm.a2[j].y) ^ __float_as_uint(hv[2 * j + 1]));
        }
        return;
    }
    float s;
    if (FROM_ROW) {
        f32x2 p = {SH::LPH == 1 ? sd : 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < VEC; j += 2) {
            const f32x2 h2 = {hv[j], hv[j + 1]}, a2 = {av[j], av[j + 1]};
            p = __builtin_elementwise_fma(h2, a2, p);
        }
        s = p.x + p.y;
        if constexpr (SH::LPH > 1) s = lanes_sum<SH::LPH>(s) + sd;
    } else {
        s = fmaf(NEG_LOG2E, ss, sd);                      // (ss: the source's score as stored)
    }
    const float lr = fminf(s, a.alpha * s);               // -log2e * leaky relu, 0 < alpha < 1
    float ev = ok ? __builtin_amdgcn_exp2f(lr) : 0.f;
    sm.rsum += ev;
    const bool neg = !(s < 0.f);                          // the unscaled score is not positive (the backward's lrelu' takes alpha there)
    if constexpr (SAVE) sm.rneg += neg ? ev : 0.f;
    if (DROP) {
        const uint32_t other = dst_is_user ? (uint32_t)c * RNG_K2 + salt_h : (uint32_t)c * RNG_K1;
        ev = lowbias32(row_key ^ other) >= a.drop_thr ? ev * a.drop_scale : 0.f;
    }
    if constexpr (SAVE) {
        // both sums of a column in ONE packed FMA: (acc[j], accn[j]) += (ev, evn) * (h[j], h[j]) -- the column broadcast by op_sel
        const f32x2 w = {ev, neg ? ev : 0.f};
#pragma unroll
        for (int j = 0; j < VEC; ++j) sm.a2[j] = __builtin_elementwise_fma(w, f32x2{hv[j], hv[j]}, sm.a2[j]);
    } else {
        const f32x2 w = {ev, ev};
#pragma unroll
        for (int j = 0; j < VEC / 2; ++j) sm.a2[j] = __builtin_elementwise_fma(w, f32x2{hv[2 * j], hv[2 * j + 1]}, sm.a2[j]);
    }
}

// The per-row constants of edge_step, pre-scaled (see there)
template <typename T, int D, int HEADS, bool FROM_ROW>
__device__ __forceinline__ void row_consts(const PropArgs& a, int row, int q, int head, bool dst_is_user, float& sd, float (&av)[Slice<T>::VEC]) {
    constexpr int VEC = Slice<T>::VEC;
    sd = NEG_LOG2E * a.S_dst[(size_t)row * HEADS + head];
    if (FROM_ROW) {
        const float* ap = a.avec + (dst_is_user ? D : 0) + q * VEC;   // sources are of the other type
#pragma unroll
        for (int j = 0; j < VEC; ++j) av[j] = NEG_LOG2E * ap[j];
    }
}

// The traversal of one wave's work item(s).  NEXT (the fused form, ngacf_prop_fwd_next): the finished row of this lane's group
// is handed back in `ro` for the workgroup's transform tail (which also stores it), and no path leaves before it.
template <typename T, int D, int HEADS, bool FROM_ROW, bool DROP, bool SPARSE, bool NEXT, bool SAVE, int ROOF = 0>
__device__ __forceinline__ void prop_fwd_body(const PropArgs& a, int kind, int widx, int wend, RowOut<Slice<T>::VEC>& ro) {
    using SH = Shape<T, D, HEADS>;
    constexpr int VEC = SH::VEC, LPR = SH::LPR, EPI = SH::EPI, SHIFT = SH::ROW_SHIFT;
    const int lane = threadIdx.x & 63;
    const int slot = lane / LPR, q = lane % LPR;
    const int head = (q * VEC) / SH::HW;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    float rsum = 0.f;
    NegSums<VEC> ng;                                      // (SAVE only: dead code otherwise)
#pragma unroll
    for (int j = 0; j < VEC; ++j) ng.accn[j] = 0.f;
    ng.rneg = 0.f;
    constexpr int NP = SAVE ? 2 : 1;                      // planes of a hub row's partial sums: (acc, rsum) and, SAVE, (accn, rneg)
    float av[VEC];
    // edges of a batch whose arithmetic the compiler may interleave (the order pins sit behind every PIN_EVERY-th edge): 1 keeps the
    // fewest registers live (eval: 5 waves per SIMD); a train-mode forward (dropout hash, two planes of sums: 4 waves anyway) can
    // afford a second edge in flight for its slow integer multiplies to hide behind
    constexpr int PIN_EVERY = (SAVE || DROP) ? NGACF_FWD_PIN_TRAIN : NGACF_FWD_PIN;
    // Two batches of gathers in flight per wave.  NGACF_PIPE_TRAIN=0 keeps ONE for a train-mode forward (which carries the second
    // plane of sums and the dropout hash: the second batch costs it a wave per SIMD) -- measured: alone 45.6 / 43.8 us against 42.7 /
    // 39.5 with two (8 heads / 1), the two-chain bf16 train step the same (0.620 / 0.621 ms), the fp32 one 0.945 against 0.931.
    constexpr bool PIPE = SAVE ? (NGACF_PIPE_TRAIN != 0) : true;
    Sums<VEC, SAVE> sm;                                   // the running sums inside the edge loops (pairs); acc / rsum / ng take them over behind
    sm.zero();
    if (kind == WORK_HUB) {
        // ------------------------------------------------------------------ hub rows: workgroup per row
        // The four waves stride over the row's edges (EPI * UNROLL per wave and step); slots fold inside each wave,
        // waves through LDS in wave order (deterministic); wave 0 finishes the row.  No partial sums in memory.
        __shared__ float hub_acc[3][NP * D];
        __shared__ float hub_r[3][NP * HEADS];
        const int wv = threadIdx.x >> 6;
        const int row = uniform(a.hrow[widx]);
        if (SPARSE && !flagged(a.needed, row)) return;                // (block-uniform: no wave reaches the barrier below)
        const int beg = uniform(a.rowptr[row]), end = uniform(a.rowptr[row + 1]);
        const bool dst_is_user = row < a.U;
        float s_dst;
        row_consts<T, D, HEADS, FROM_ROW>(a, row, q, head, dst_is_user, s_dst, av);
        uint32_t salt_h = 0, row_key = 0;
        if (DROP) {
            salt_h = salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 1u + head);
            row_key = dst_is_user ? (uint32_t)row * RNG_K1 : (uint32_t)row * RNG_K2 + salt_h;
        }
        for (int base = beg + wv * EPI * UNROLL; base < end; base += 4 * EPI * UNROLL) {
            int c[UNROLL];
            bool ok[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int e = base + u * EPI + slot;
                ok[u] = e < end;
                c[u] = a.col[ok[u] ? e : end - 1];
            }
            float hv[UNROLL][VEC], ss[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                gather_slice<T, SHIFT, ROOF>(a.H, c[u], q, hv[u]);
                ss[u] = FROM_ROW ? 0.f : a.S_src[(size_t)c[u] * HEADS + head];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                edge_step<T, D, HEADS, FROM_ROW, DROP, SAVE, ROOF>(a, hv[u], ss[u], ok[u], c[u], s_dst, av, dst_is_user, row_key, salt_h, sm);
        }
        sm.get(acc, rsum, ng);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
        rsum = group_sum<LPR, WAVE>(rsum);
        if constexpr (SAVE) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) ng.accn[j] = group_sum<LPR, WAVE>(ng.accn[j]);
            ng.rneg = group_sum<LPR, WAVE>(ng.rneg);
        }
        if (wv > 0 && slot == 0) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) hub_acc[wv - 1][q * VEC + j] = acc[j];
            if ((q * VEC) % SH::HW == 0) hub_r[wv - 1][head] = rsum;
            if constexpr (SAVE) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) hub_acc[wv - 1][D + q * VEC + j] = ng.accn[j];
                if ((q * VEC) % SH::HW == 0) hub_r[wv - 1][HEADS + head] = ng.rneg;
            }
        }
        __syncthreads();
        if (wv == 0 && slot == 0) {
#pragma unroll
            for (int x = 0; x < 3; ++x) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += hub_acc[x][q * VEC + j];
                rsum += hub_r[x][head];
                if constexpr (SAVE) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) ng.accn[j] += hub_acc[x][D + q * VEC + j];
                    ng.rneg += hub_r[x][HEADS + head];
                }
            }
            finish_row<T, D, HEADS, SAVE, ROOF>(a, row, q, head, acc, rsum, ng);
        }
        return;
    }
    if (widx >= wend) return;
    // A batch of UNROLL gathers in flight: the raw slices, the sources' scores (only without NGACF_SCORE_ROW) and their ids.
    struct Batch {
        u32x4 r[UNROLL];
        float ss[UNROLL];
        int c[UNROLL];
    };
    auto issue = [&](Batch& b) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            b.r[u] = gather_raw<SHIFT>(a.H, b.c[u], q);
            b.ss[u] = FROM_ROW ? 0.f : a.S_src[(size_t)b.c[u] * HEADS + head];
        }
    };
    if (kind == WORK_SHORT) {
        // ------------------------------------------------------------------ short rows: slot per row
        const int ridx = widx + slot;
        const int4 rec = a.srec[ridx < wend ? ridx : wend - 1];
        const bool live = ridx < wend && rec.x >= 0;         // row -1 = padding between the user and item lists
        const int row = live ? rec.x : 0;
        const int beg = live ? rec.y : 0;
        const int deg = live ? rec.z : 0;
        const int maxdeg = uniform(deg);                  // rows are sorted: slot 0 holds the longest
        const bool dst_is_user = row < a.U;
        float s_dst;
        row_consts<T, D, HEADS, FROM_ROW>(a, row, q, head, dst_is_user, s_dst, av);
        uint32_t salt_h = 0, row_key = 0;
        if (DROP) {
            salt_h = salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 1u + head);
            row_key = dst_is_user ? (uint32_t)row * RNG_K1 : (uint32_t)row * RNG_K2 + salt_h;
        }
        const int last = deg > 0 ? beg + deg - 1 : 0;     // any valid index for the masked tail
        // edges k .. k + UNROLL - 1 of every slot's row.  Rows are sorted by degree, so the wave's shortest row sits in its last
        // slot: while k + UNROLL <= that degree no slot has a masked edge and the compare + select per edge are left out.
        const int mindeg = __builtin_amdgcn_readlane(deg, 63);
        auto compute_t = [&](Batch& b, int k, auto masked) {
            constexpr bool MASKED = decltype(masked)::value;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                float hv[VEC];
                if (u == 0) pin(b.r[0]);                  // (the batch's first unpack stays behind the phase fence in front of it)
                unpack_raw<T, ROOF>(b.r[u], hv);
                edge_step<T, D, HEADS, FROM_ROW, DROP, SAVE, ROOF>(a, hv, b.ss[u], MASKED ? k + u < deg : true, b.c[u], s_dst, av, dst_is_user,
                                                                   row_key, salt_h, sm);
                if ((u + 1) % PIN_EVERY == 0 || u + 1 == UNROLL) {
                    sm.template pin_all<ROOF>();
                    if (u + 1 < UNROLL) pin(b.r[u + 1]);
                }
            }
        };
        auto compute = [&](Batch& b, int k) {
            if (k + UNROLL <= mindeg) compute_t(b, k, BoolC<false>{});
            else compute_t(b, k, BoolC<true>{});
        };
        // Lanes 0..7 of a slot hold the next 8 column indices of the slot's row: batch 2i = the first four, batch 2i + 1 = the last
        // four.  Batches alternate between A and B; inside the steady-state loop every issue is UNCONDITIONAL (a load behind a
        // branch makes the compiler wait for vmcnt(0) at the next use of anything older) and the phases are fenced for the
        // scheduler, which otherwise hoists the next issue above the compute and copies the batch: issue B | compute A | issue A' |
        // compute B.
        static_assert(UNROLL == 4, "two batches of four per 8 broadcast indices");
        const int nb = (maxdeg + UNROLL - 1) / UNROLL;    // batches of the longest row (uniform)
        int colv = a.col[min(beg + (q & 7), last)];
        Batch A, B;
        auto pick_lo = [&](Batch& b, int cv) {
            b.c[0] = slot_bcast<LPR, 0>(cv); b.c[1] = slot_bcast<LPR, 1>(cv); b.c[2] = slot_bcast<LPR, 2>(cv); b.c[3] = slot_bcast<LPR, 3>(cv);
        };
        auto pick_hi = [&](Batch& b, int cv) {
            b.c[0] = slot_bcast<LPR, 4>(cv); b.c[1] = slot_bcast<LPR, 5>(cv); b.c[2] = slot_bcast<LPR, 6>(cv); b.c[3] = slot_bcast<LPR, 7>(cv);
        };
        if constexpr (!PIPE) {                            // one batch at a time (train mode: see PIPE)
            for (int j = 0; j < nb; j += 2) {
                const int coln = a.col[min(beg + 4 * j + 8 + (q & 7), last)];
                pick_lo(A, colv);
                issue(A);
                compute(A, 4 * j);
                if (j + 1 < nb) {
                    pick_hi(A, colv);
                    issue(A);
                    compute(A, 4 * j + 4);
                }
                colv = coln;
            }
        } else if (nb > 0) {
            pick_lo(A, colv);
            issue(A);
            int j = 0;
            for (; j + 2 < nb; j += 2) {                  // batches j (A, in flight), j + 1 and j + 2 exist
                const int coln = a.col[min(beg + 4 * j + 8 + (q & 7), last)];     // the next 8 indices
                pick_hi(B, colv);
                issue(B);
                NGACF_PHASE_FENCE();
                compute(A, 4 * j);
                NGACF_PHASE_FENCE();
                colv = coln;
                pick_lo(A, colv);
                issue(A);
                NGACF_PHASE_FENCE();
                compute(B, 4 * j + 4);
                NGACF_PHASE_FENCE();
            }
            if (j + 1 < nb) {                             // two batches left: A in flight, B the last
                pick_hi(B, colv);
                issue(B);
                NGACF_PHASE_FENCE();
                compute(A, 4 * j);
                compute(B, 4 * j + 4);
            } else {
                compute(A, 4 * j);
            }
        }
        sm.get(acc, rsum, ng);
        if (live) {
            if constexpr (NEXT) finish_row_defer<T, D, HEADS, SAVE>(a, row, q, head, acc, rsum, ng, ro);
            else finish_row<T, D, HEADS, SAVE, ROOF>(a, row, q, head, acc, rsum, ng);
        }
        return;
    }

    // ---------------------------------------------------------------------- long rows: wave per segment
    const int w = uniform(widx);
    int row, beg, end, part;
    if (SPARSE && w < a.order.nseg_u) {
        // SPARSE, first group: the batch's own row list, a WAVE per short row (the lane groups share its edges like a segment's:
        // a row of 64 edges is two rounds of gathers, not eight).  Longer rows are carried by their segments below.
        row = uniform(a.rows[w]);
        if (row < 0) return;                                 // an id out of range (reported by the scoring kernel)
        beg = uniform(a.rowptr[row]); end = uniform(a.rowptr[row + 1]); part = -1;
        if (end - beg > a.short_max) return;
    } else {
        const int4 sg = a.segs[SPARSE ? w - a.order.nseg_u : w];
        row = uniform(sg.x); beg = uniform(sg.y); end = uniform(sg.z); part = uniform(sg.w);
        if (SPARSE && !flagged(a.needed, row)) return;       // (its partial-sum slot stays unwritten: the second kernel skips the row too)
    }
    const bool dst_is_user = row < a.U;
    float s_dst;
    row_consts<T, D, HEADS, FROM_ROW>(a, row, q, head, dst_is_user, s_dst, av);
    uint32_t salt_h = 0, row_key = 0;
    if (DROP) {
        salt_h = salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 1u + head);
        row_key = dst_is_user ? (uint32_t)row * RNG_K1 : (uint32_t)row * RNG_K2 + salt_h;
    }
    {
        // A step covers STEP = EPI * UNROLL edges: slot `sl`, gather `u` takes edge base + u * EPI + sl (the order of r1-r4).  The
        // column indices of EIGHT steps are fetched at once -- lane q & 7 of a slot keeps the slot's four indices of step q & 7 --
        // and handed out with ds_bpermute, so a step no longer starts with a dependent index load in front of its row gathers; the
        // row gathers of step t + 1 are issued before step t is computed (two batches in flight, as in the short rows).
        constexpr int STEP = EPI * UNROLL, CHUNK = 8 * STEP;
        const int bp_base = (lane & ~(LPR - 1)) << 2;        // byte address of lane 0 of this slot, for ds_bpermute
        for (int cb = beg; cb < end; cb += CHUNK) {           // (one chunk per segment at the default seg_len with 128-byte rows)
            int colr[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) colr[u] = a.col[min(cb + (q & 7) * STEP + u * EPI + slot, end - 1)];
            const int nstep = min(8, (end - cb + STEP - 1) / STEP);       // (uniform)
            auto pick = [&](Batch& b, int t) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) b.c[u] = __builtin_amdgcn_ds_bpermute(bp_base + (t << 2), colr[u]);
            };
            auto compute = [&](Batch& b, int t) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    float hv[VEC];
                    if (u == 0) pin(b.r[0]);
                    unpack_raw<T, ROOF>(b.r[u], hv);
                    edge_step<T, D, HEADS, FROM_ROW, DROP, SAVE, ROOF>(a, hv, b.ss[u], cb + t * STEP + u * EPI + slot < end, b.c[u], s_dst, av,
                                                                       dst_is_user, row_key, salt_h, sm);
                    if ((u + 1) % PIN_EVERY == 0 || u + 1 == UNROLL) {
                        sm.template pin_all<ROOF>();
                        if (u + 1 < UNROLL) pin(b.r[u + 1]);
                    }
                }
            };
            Batch A, B;
            if constexpr (!PIPE) {
                for (int t = 0; t < nstep; ++t) {
                    pick(A, t);
                    issue(A);
                    compute(A, t);
                }
                continue;
            }
            pick(A, 0);
            issue(A);
            int t = 0;
            for (; t + 2 < nstep; t += 2) {                   // unconditional issues, fenced phases (see the short rows)
                pick(B, t + 1);
                issue(B);
                NGACF_PHASE_FENCE();
                compute(A, t);
                NGACF_PHASE_FENCE();
                pick(A, t + 2);
                issue(A);
                NGACF_PHASE_FENCE();
                compute(B, t + 1);
                NGACF_PHASE_FENCE();
            }
            if (t + 1 < nstep) {
                pick(B, t + 1);
                issue(B);
                NGACF_PHASE_FENCE();
                compute(A, t);
                compute(B, t + 1);
            } else {
                compute(A, t);
            }
        }
    }
    sm.get(acc, rsum, ng);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
    rsum = group_sum<LPR, WAVE>(rsum);
    if constexpr (SAVE) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) ng.accn[j] = group_sum<LPR, WAVE>(ng.accn[j]);
        ng.rneg = group_sum<LPR, WAVE>(ng.rneg);
    }
    if (part < 0) {
        if (slot == 0) {                                      // one finished row per wave, in lane group 0
            if constexpr (NEXT) finish_row_defer<T, D, HEADS, SAVE>(a, row, q, head, acc, rsum, ng, ro);
            else finish_row<T, D, HEADS, SAVE, ROOF>(a, row, q, head, acc, rsum, ng);
        }
        return;
    }
    // slice of a hub row: leave the partial sums in slot `part` (SAVE: a slot is [2][D] / [2][HEADS], the negative-side sums second)
    if (a.tickets == nullptr) {                            // prop_fwd_split_kernel finishes the row
        if (slot == 0) {
            float* pa = a.part_acc + (size_t)part * NP * D + q * VEC;
#pragma unroll
            for (int j = 0; j < VEC; j += 4) *reinterpret_cast<float4*>(pa + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
            if ((q * VEC) % SH::HW == 0) a.part_r[(size_t)part * NP * HEADS + head] = rsum;
            if constexpr (SAVE) {
#pragma unroll
                for (int j = 0; j < VEC; j += 4)
                    *reinterpret_cast<float4*>(pa + D + j) = make_float4(ng.accn[j], ng.accn[j + 1], ng.accn[j + 2], ng.accn[j + 3]);
                if ((q * VEC) % SH::HW == 0) a.part_r[(size_t)part * NP * HEADS + HEADS + head] = ng.rneg;
            }
        }
        return;
    }
    // The slice that arrives LAST finishes the row in this launch (cdna_hip_programming.md Guideline 16, counter form):
    // write-through (sc1) stores of the partial sums -> drain -> relaxed agent-scope ticket; the last arriver: agent-scope
    // acquire -> plain loads.  No release fence (a buffer_wbl2 per slice tripled the kernel time), no wave ever waits on
    // another one.  Partials are added in slot order, so the result does not depend on which slice came last.
    if (slot == 0) {
        unsigned long long* pa = reinterpret_cast<unsigned long long*>(a.part_acc + (size_t)part * NP * D + q * VEC);
#pragma unroll
        for (int j = 0; j < VEC; j += 2) {
            const unsigned long long bits = (unsigned long long)__float_as_uint(acc[j]) | ((unsigned long long)__float_as_uint(acc[j + 1]) << 32);
            __hip_atomic_store(pa + j / 2, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((q * VEC) % SH::HW == 0)
            __hip_atomic_store(reinterpret_cast<unsigned*>(a.part_r + (size_t)part * NP * HEADS + head), __float_as_uint(rsum),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (SAVE) {
#pragma unroll
            for (int j = 0; j < VEC; j += 2) {
                const unsigned long long bits = (unsigned long long)__float_as_uint(ng.accn[j]) | ((unsigned long long)__float_as_uint(ng.accn[j + 1]) << 32);
                __hip_atomic_store(pa + (D + j) / 2, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if ((q * VEC) % SH::HW == 0)
                __hip_atomic_store(reinterpret_cast<unsigned*>(a.part_r + (size_t)part * NP * HEADS + HEADS + head), __float_as_uint(ng.rneg),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int si = uniform(a.part_split[part]);
    const int p0 = uniform(a.split[3 * si + 1]), np = uniform(a.split[3 * si + 2]);
    int tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add(a.tickets + si, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (uniform(tk) != np - 1) return;
    // No acquire fence: an agent-scope acquire invalidates this XCD's L2 (buffer_inv sc1) under a kernel that lives on its
    // cache hits -- it cost the launch 4.5 us.  The partial sums are read with agent-scope atomic loads instead (sc1: served
    // at the coherent level, where the other slices' write-through stores are, whatever this L2 holds).
    if (lane == 0) __hip_atomic_store(a.tickets + si, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc[j] = 0.f; ng.accn[j] = 0.f; }
    rsum = 0.f;
    ng.rneg = 0.f;
#pragma unroll 4
    for (int p = p0 + slot; p < p0 + np; p += EPI) {
        unsigned long long* pa = reinterpret_cast<unsigned long long*>(a.part_acc + (size_t)p * NP * D + q * VEC);
#pragma unroll
        for (int j = 0; j < VEC; j += 2) {
            const unsigned long long bits = __hip_atomic_load(pa + j / 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc[j] += __uint_as_float((unsigned)bits); acc[j + 1] += __uint_as_float((unsigned)(bits >> 32));
        }
        rsum += __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned*>(a.part_r + (size_t)p * NP * HEADS + head), __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT));
        if constexpr (SAVE) {
#pragma unroll
            for (int j = 0; j < VEC; j += 2) {
                const unsigned long long bits = __hip_atomic_load(pa + (D + j) / 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ng.accn[j] += __uint_as_float((unsigned)bits); ng.accn[j + 1] += __uint_as_float((unsigned)(bits >> 32));
            }
            ng.rneg += __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned*>(a.part_r + (size_t)p * NP * HEADS + HEADS + head),
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
    rsum = group_sum<LPR, WAVE>(rsum);
    if constexpr (SAVE) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) ng.accn[j] = group_sum<LPR, WAVE>(ng.accn[j]);
        ng.rneg = group_sum<LPR, WAVE>(ng.rneg);
    }
    if (slot == 0) {
        if constexpr (NEXT) finish_row_defer<T, D, HEADS, SAVE>(a, row, q, head, acc, rsum, ng, ro);
        else finish_row<T, D, HEADS, SAVE, ROOF>(a, row, q, head, acc, rsum, ng);
    }
}

// Waves per SIMD the instantiations are bounded to (= their VGPR budget: 512 / waves, in steps of 8).  With two batches of gathers
// in flight a lane holds 32 VGPRs of raw slices; the budgets below are the largest occupancy each family reaches without spilling
// (tools/kres.py over -Rpass-analysis=kernel-resource-usage), A/B'd on the GPU through the -D overrides (tools/build_variant.sh).
#ifndef NGACF_W_EVAL_BF16
#define NGACF_W_EVAL_BF16 5
#endif
#ifndef NGACF_W_NEXT_BF16
#define NGACF_W_NEXT_BF16 4
#endif
#ifndef NGACF_W_SAVE_BF16
#define NGACF_W_SAVE_BF16 (NGACF_PIPE_TRAIN ? 4 : 5)
#endif
#ifndef NGACF_W_NEXT_SAVE_BF16
#define NGACF_W_NEXT_SAVE_BF16 (NGACF_PIPE_TRAIN ? 4 : 5)
#endif
#ifndef NGACF_W_SAVE_F32
#define NGACF_W_SAVE_F32 5
#endif
#ifndef NGACF_W_NEXT_SAVE_F32
#define NGACF_W_NEXT_SAVE_F32 5
#endif
template <typename T, int D, bool SPARSE, int NEXT, bool SAVE>
constexpr int fwd_waves() {
    constexpr bool B16 = sizeof(T) == 2;
    if (NEXT == 128) return 4;
    if (NEXT == 64) return SAVE ? (B16 ? NGACF_W_NEXT_SAVE_BF16 : NGACF_W_NEXT_SAVE_F32) : (B16 ? NGACF_W_NEXT_BF16 : 1);
    if (D != 64) return 1;
    if (B16 && !SPARSE) return SAVE ? NGACF_W_SAVE_BF16 : NGACF_W_EVAL_BF16;
    if (!B16 && SAVE) return NGACF_W_SAVE_F32;
    return 1;
}

// NEXT > 0 (= the next stage's output width; NONE: that stage has ONE head): the fused form of ngacf_prop_fwd_next.
// SAVE: a train-mode forward (agg / R given): also the negative-side sums the backward wants (NegSums above).
template <typename T, int D, int HEADS, bool FROM_ROW, bool DROP, bool SPARSE, int NEXT = 0, bool NONE = false, bool STAGE_OUT = false, bool SAVE = false, int ROOF = 0>
__global__ __launch_bounds__(256, (fwd_waves<T, D, SPARSE, NEXT, SAVE>())) void prop_fwd_kernel(const PropArgs a) {
    using SH = Shape<T, D, HEADS>;
    constexpr int VEC = SH::VEC, EPI = SH::EPI;
    static_assert(NEXT == 0 || (D == 64 && !SPARSE), "fused next-stage transform: rows of 64 columns");
    static_assert(ROOF == 0 || (NEXT == 0 && FROM_ROW && !DROP && !SPARSE && !SAVE), "measurement-only instantiations: the plain eval traversal");
    int kind, widx, wend;
    locate(a.order, blockIdx.x, threadIdx.x >> 6, EPI, kind, widx, wend);
    RowOut<VEC> ro;
    ro.valid = false;
    ro.row = 0;
    const int lane_q = (threadIdx.x & 63) % SH::LPR;
    auto stores = [&]() {
        if (ro.valid) finish_row_store<T, D, HEADS, SAVE>(a, ro.row, lane_q, (lane_q * VEC) / SH::HW, ro.agg, ro.rsum, ro.ng, ro.o);
    };
    if constexpr (NEXT > 0 && sizeof(T) == 2) {
        // the workgroup's weight panel (its rows are all users or all items) goes global -> LDS by LDS-DMA right here and lands
        // while the waves traverse; the barrier of the tail publishes it together with the finished rows
        constexpr int PANEL_U4 = (D / 16) * (NEXT / 32) * 64;
        __shared__ __attribute__((aligned(16))) uint4 panel_lds[PANEL_U4];
        __shared__ __attribute__((aligned(16))) uint4 xtile_lds[32 * (NEXT / 8 + 1)];      // (rows of 9 going in, of NEXT / 8 + 1 going out)
        __shared__ int xmeta_lds[32];
        __shared__ __attribute__((aligned(16))) float4 avec_lds[NEXT / 4];
        const bool users = (int)blockIdx.x < a.order.b2;
        const uint4* src = a.next_panel + (users ? 0 : PANEL_U4);
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int piece = threadIdx.x >> 6; piece < PANEL_U4 / 64; piece += 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(panel_lds + piece * 64), 16, 0, 0);
        if (threadIdx.x < NEXT / 4)                             // the next stage's attention vector of this node type, the same way
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.next_avec + (users ? 0 : NEXT) + 4 * lane),
                                             (__attribute__((address_space(3))) void*)avec_lds, 16, 0, 0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) ro.o[j] = 0.f;
        prop_fwd_body<T, D, HEADS, FROM_ROW, DROP, SPARSE, true, SAVE>(a, kind, widx, wend, ro);
        next_transform_tail<D, NEXT, NONE, STAGE_OUT>(a, xtile_lds, xmeta_lds, panel_lds, reinterpret_cast<const float*>(avec_lds), ro.valid, ro.row,
                                           lane / SH::LPR, lane % SH::LPR, ro.o, stores);
    } else if constexpr (NEXT > 0) {
        // fp32 storage: the panel is the next stage's fp32 W'[type] itself ([64][NEXT], no packing); a 1 KB LDS-DMA piece = 256
        // consecutive floats of the LDS image; the lane's 16 bytes come from the column block the source-side swizzle names
        __shared__ __attribute__((aligned(16))) float panel_f32[D * NEXT];
        __shared__ __attribute__((aligned(16))) float xtile_f32[16 * XROW_F32];
        __shared__ int xmeta_f32[16];
        __shared__ __attribute__((aligned(16))) float avec_f32[NEXT];
        __shared__ float sx_f32[(NEXT / 64) * 64];
        const bool users = (int)blockIdx.x < a.order.b2;
        const float* src = reinterpret_cast<const float*>(a.next_panel) + (users ? 0 : D * NEXT);
        const int lane = threadIdx.x & 63;
        load_panel_f32<NEXT>(panel_f32, src);
        if (threadIdx.x < NEXT / 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.next_avec + (users ? 0 : NEXT) + 4 * lane),
                                             (__attribute__((address_space(3))) void*)avec_f32, 16, 0, 0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) ro.o[j] = 0.f;
        prop_fwd_body<T, D, HEADS, FROM_ROW, DROP, SPARSE, true, SAVE>(a, kind, widx, wend, ro);
        next_transform_tail_f32<D, NEXT, NONE>(a, xtile_f32, xmeta_f32, panel_f32, avec_f32, sx_f32, ro.valid, ro.row, lane / SH::LPR,
                                               lane % SH::LPR, ro.o, stores);
    } else {
        prop_fwd_body<T, D, HEADS, FROM_ROW, DROP, SPARSE, false, SAVE, ROOF>(a, kind, widx, wend, ro);
    }
}

// Hub rows: add the partial sums of the row's slices (slots take interleaved slices, fixed order) and finish.
template <typename T, int D, int HEADS, bool SAVE>
__global__ __launch_bounds__(256) void prop_fwd_split_kernel(const PropArgs a, const int* __restrict__ split, int nsplit) {
    using SH = Shape<T, D, HEADS>;
    constexpr int VEC = SH::VEC, LPR = SH::LPR, EPI = SH::EPI, NP = SAVE ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nsplit) return;
    const int row = split[3 * w], p0 = split[3 * w + 1], np = split[3 * w + 2];
    if (a.needed && !flagged(a.needed, row)) return;
    const int slot = lane / LPR, q = lane % LPR, head = (q * VEC) / SH::HW;
    float acc[VEC];
    NegSums<VEC> ng;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc[j] = 0.f; ng.accn[j] = 0.f; }
    float rsum = 0.f;
    ng.rneg = 0.f;
#pragma unroll 4
    for (int p = p0 + slot; p < p0 + np; p += EPI) {
        const float* pa = a.part_acc + (size_t)p * NP * D + q * VEC;
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const float4 v = *reinterpret_cast<const float4*>(pa + j);
            acc[j] += v.x; acc[j + 1] += v.y; acc[j + 2] += v.z; acc[j + 3] += v.w;
        }
        rsum += a.part_r[(size_t)p * NP * HEADS + head];
        if constexpr (SAVE) {
#pragma unroll
            for (int j = 0; j < VEC; j += 4) {
                const float4 v = *reinterpret_cast<const float4*>(pa + D + j);
                ng.accn[j] += v.x; ng.accn[j + 1] += v.y; ng.accn[j + 2] += v.z; ng.accn[j + 3] += v.w;
            }
            ng.rneg += a.part_r[(size_t)p * NP * HEADS + HEADS + head];
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
    rsum = group_sum<LPR, WAVE>(rsum);
    if constexpr (SAVE) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) ng.accn[j] = group_sum<LPR, WAVE>(ng.accn[j]);
        ng.rneg = group_sum<LPR, WAVE>(ng.rneg);
    }
    if (slot != 0) return;
    finish_row<T, D, HEADS, SAVE>(a, row, q, head, acc, rsum, ng);
}

template <typename T, int D, int HEADS, bool SAVE>
static int launch_fwd_s(PropArgs a, const int* split, int nsplit, bool from_row, bool drop, hipStream_t st, int next_dout, bool next_one,
                        bool next_staged) {
    constexpr int EPI = Shape<T, D, HEADS>::EPI;
    int nb = 0;
    a.order = make_order(a.nseg, a.order.nseg_u, a.nshort, a.order.nshort_u, a.order.nhub, a.order.nhub_u, EPI, &nb);
    if (next_dout) {
        if constexpr (D == 64 && HEADS == 8) {
            if (nb == 0) return 0;
            const dim3 grid(nb), block(256);
#define NGACF_NEXT(ND, ONE, STG)                                                                                         \
            do {                                                                                                         \
                if (drop) launch_k(NGACF_TAG_PROP_FWD, prop_fwd_kernel<T, D, HEADS, true, true, false, ND, ONE, STG, SAVE>, grid, block, 0, st, a);   \
                else      launch_k(NGACF_TAG_PROP_FWD, prop_fwd_kernel<T, D, HEADS, true, false, false, ND, ONE, STG, SAVE>, grid, block, 0, st, a);  \
            } while (0)
            constexpr bool B16 = sizeof(T) == 2;             // (the staged form exists for bf16 storage only)
            if (B16 && next_staged) {
                if (next_dout == 64 && !next_one) NGACF_NEXT(64, false, B16);
                else if (next_dout == 64) NGACF_NEXT(64, true, B16);
                else if (next_dout == 128 && !next_one) NGACF_NEXT(128, false, B16);
                else if (next_dout == 128) NGACF_NEXT(128, true, B16);
                else return NGACF_E_SHAPE;
            } else if (next_dout == 64 && !next_one) NGACF_NEXT(64, false, false);
            else if (next_dout == 64) NGACF_NEXT(64, true, false);
            else if (next_dout == 128 && !next_one) NGACF_NEXT(128, false, false);
            else if (next_dout == 128) NGACF_NEXT(128, true, false);
            else return NGACF_E_SHAPE;
#undef NGACF_NEXT
            NGACF_CHECK_LAUNCH();
            return 0;
        } else {
            return NGACF_E_SHAPE;
        }
    }
    if (a.flags & (NGACF_ROOF_ADD | NGACF_ROOF_XOR)) {          // measurement only (tools/gather_roof.py): see gather_slice
        if constexpr (!SAVE && D == 64) {
            if (a.needed || !from_row || drop || (nsplit > 0 && a.tickets == nullptr)) return NGACF_E_ARG;
            if (nb == 0) return 0;
            const dim3 grid(nb), block(256);
            if (a.flags & NGACF_ROOF_XOR) launch_k(NGACF_TAG_PROP_FWD, prop_fwd_kernel<T, D, HEADS, true, false, false, 0, false, false, false, 2>, grid, block, 0, st, a);
            else                          launch_k(NGACF_TAG_PROP_FWD, prop_fwd_kernel<T, D, HEADS, true, false, false, 0, false, false, false, 1>, grid, block, 0, st, a);
            NGACF_CHECK_LAUNCH();
            return 0;
        } else {
            return NGACF_E_SHAPE;
        }
    }
    if (nb > 0) {
        const dim3 grid(nb), block(256);
        if (a.needed) {                 // (only with NGACF_SCORE_ROW: checked by the caller)
            if (drop) launch_k(NGACF_TAG_PROP_FWD, prop_fwd_kernel<T, D, HEADS, true, true, true, 0, false, false, SAVE>, grid, block, 0, st, a);
            else      launch_k(NGACF_TAG_PROP_FWD, prop_fwd_kernel<T, D, HEADS, true, false, true, 0, false, false, SAVE>, grid, block, 0, st, a);
        } else if (from_row) {
            if (drop) launch_k(NGACF_TAG_PROP_FWD, prop_fwd_kernel<T, D, HEADS, true, true, false, 0, false, false, SAVE>, grid, block, 0, st, a);
            else      launch_k(NGACF_TAG_PROP_FWD, prop_fwd_kernel<T, D, HEADS, true, false, false, 0, false, false, SAVE>, grid, block, 0, st, a);
        } else {
            if (drop) launch_k(NGACF_TAG_PROP_FWD, prop_fwd_kernel<T, D, HEADS, false, true, false, 0, false, false, SAVE>, grid, block, 0, st, a);
            else      launch_k(NGACF_TAG_PROP_FWD, prop_fwd_kernel<T, D, HEADS, false, false, false, 0, false, false, SAVE>, grid, block, 0, st, a);
        }
        NGACF_CHECK_LAUNCH();
    }
    if (nsplit > 0 && a.tickets == nullptr) {
        launch_k(NGACF_TAG_PROP_FWD_SPLIT, prop_fwd_split_kernel<T, D, HEADS, SAVE>, dim3((nsplit + 3) / 4), dim3(256), 0, st, a, split, nsplit);
        NGACF_CHECK_LAUNCH();
    }
    return 0;
}

// agg and R given (a train-mode forward): the SAVE instantiations, which also keep the negative-side sums (second planes of agg / R)
template <typename T, int D, int HEADS>
static int launch_fwd(PropArgs a, const int* split, int nsplit, bool from_row, bool drop, hipStream_t st, int next_dout = 0, bool next_one = false,
                      bool next_staged = false) {
    if (a.agg) return launch_fwd_s<T, D, HEADS, true>(a, split, nsplit, from_row, drop, st, next_dout, next_one, next_staged);
    return launch_fwd_s<T, D, HEADS, false>(a, split, nsplit, from_row, drop, st, next_dout, next_one, next_staged);
}

}  // namespace ngacf

using namespace ngacf;

extern "C" int ngacf_prop_fwd_next(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                                   const int32_t* srow, int nshort, int nseg_user, int nshort_user,
                                   const int32_t* hrow, int nhub, int nhub_user, const int32_t* rowptr,
                                   const int32_t* col, const void* H, int dtype,
                                   const float* S_dst, const float* S_src, const float* avec,
                                   void* out, void* agg, float* R, float* part_acc, float* part_r,
                                   const int32_t* part_split, int32_t* tickets,
                                   const uint32_t* needed, const int32_t* rows, int nrows, int short_max,
                                   int U, int N, int d, int heads, float alpha,
                                   float drop_p, uint64_t seed, const uint64_t* seed_dev, int flags,
                                   const ngacf_next_stage* next, void* stream) {
    if (nseg < 0 || nsplit < 0 || nshort < 0 || U < 0 || N < U) return NGACF_E_ARG;
    if (needed && !(flags & NGACF_SCORE_ROW)) return NGACF_E_ARG;              // row flags: the SPUIGACF stages only
    if (needed && (nrows < 0 || (nrows > 0 && !rows))) return NGACF_E_ARG;
    if (nseg_user < 0 || nseg_user > nseg || nshort_user < 0 || nshort_user > nshort || (nshort_user % 16 != 0 && nshort_user != nshort))
        return NGACF_E_ARG;
    if (nhub < 0 || nhub_user < 0 || nhub_user > nhub) return NGACF_E_ARG;
    if (nseg + nshort + nhub == 0) return 0;
    if (!col || !H || !S_dst || !rowptr) return NGACF_E_ARG;
    if (!out && !next) return NGACF_E_ARG;                                      // (out may be NULL only when the next stage's operands are the result)
    if ((agg == nullptr) != (R == nullptr)) return NGACF_E_ARG;                 // saved together (two planes each) or not at all
    if ((nseg > 0 && !segs) || (nshort > 0 && !srow) || (nhub > 0 && !hrow)) return NGACF_E_ARG;
    const bool from_row = (flags & NGACF_SCORE_ROW) != 0;
    if (from_row ? !avec : !S_src) return NGACF_E_ARG;
    if (nsplit > 0 && (!split || !part_acc || !part_r)) return NGACF_E_ARG;
    if (!(drop_p >= 0.f && drop_p < 1.f) || !(alpha > 0.f && alpha < 1.f)) return NGACF_E_ARG;
    const uint64_t elem = dtype == NGACF_BF16 ? 2 : 4;
    if ((uint64_t)N * (uint64_t)d * elem >= (1ull << 32)) return NGACF_E_SHAPE;   // 32-bit row offsets
    PropArgs a;
    a.segs = reinterpret_cast<const int4*>(segs); a.nseg = nseg; a.srec = reinterpret_cast<const int4*>(srow); a.nshort = nshort; a.rowptr = rowptr;
    a.hrow = hrow; a.order.nhub = nhub; a.order.nhub_u = nhub_user;
    a.col = col; a.H = H; a.S_dst = S_dst; a.S_src = S_src; a.avec = avec; a.out = out; a.agg = agg; a.R = R;
    a.part_acc = part_acc; a.part_r = part_r; a.split = split; a.part_split = part_split; a.needed = needed; a.rows = rows; a.short_max = short_max;
    if (needed) {          // groups: [the batch's row list, a wave per short row][-][all segments, by flag][-]
        a.nshort = 0; nshort_user = 0; a.nseg = nrows + nseg; nseg_user = nrows;
    }
    a.tickets = (part_split && nsplit > 0) ? tickets : nullptr; a.U = U; a.N = N; a.order.nseg_u = nseg_user; a.order.nshort_u = nshort_user; a.alpha = alpha;
    a.drop_thr = drop_threshold(drop_p); a.drop_scale = 1.f / (1.f - drop_p); a.seed = seed; a.seed_dev = seed_dev; a.flags = flags;
    a.next_panel = nullptr; a.next_avec = nullptr; a.next_H = nullptr; a.next_S = nullptr; a.next_drop_thr = 0; a.next_drop_scale = 1.f; a.next_seed = 0;
    const bool drop = drop_p > 0.f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int next_dout = 0;
    bool next_one = false, next_staged = false;
    if (next) {
        if (!next->panel || !next->avec || !next->H || !next->S) return NGACF_E_ARG;
        if (d != 64 || heads != 8 || !from_row || needed || nhub > 0) return NGACF_E_SHAPE;
        if (nsplit > 0 && a.tickets == nullptr) return NGACF_E_ARG;             // hub rows must be finished inside the launch
        if ((next->dout != 64 && next->dout != 128) || (next->heads != 1 && next->heads != next->dout / 8)) return NGACF_E_SHAPE;
        if (!(next->feat_drop_p >= 0.f && next->feat_drop_p < 1.f)) return NGACF_E_ARG;
        if ((uint64_t)N * (uint64_t)next->dout * elem >= (1ull << 32)) return NGACF_E_SHAPE;
        a.next_panel = static_cast<const uint4*>(next->panel); a.next_avec = next->avec; a.next_H = next->H; a.next_S = next->S;
        a.next_drop_thr = next->feat_drop_p > 0.f ? drop_threshold(next->feat_drop_p) : 0u;
        if (next->feat_drop_p > 0.f && a.next_drop_thr == 0) a.next_drop_thr = 1;     // (p below 2^-32: keep the branch, drop nothing measurable)
        a.next_drop_scale = 1.f / (1.f - next->feat_drop_p); a.next_seed = next->seed;
        next_dout = next->dout; next_one = next->heads == 1;
        next_staged = dtype == NGACF_BF16 && stage_out_pays((uint64_t)N, next->dout);
    }
#define NGACF_CASE(T, D, HD) return launch_fwd<T, D, HD>(a, split, nsplit, from_row, drop, st, next_dout, next_one, next_staged)
    if (dtype == NGACF_F32) {
        if (d == 64 && heads == 8) NGACF_CASE(float, 64, 8);
        if (d == 64 && heads == 1) NGACF_CASE(float, 64, 1);
        if (d == 128 && heads == 1) NGACF_CASE(float, 128, 1);
        if (d == 128 && heads == 16) NGACF_CASE(float, 128, 16);
    } else if (dtype == NGACF_BF16) {
        if (d == 64 && heads == 8) NGACF_CASE(bf16_t, 64, 8);
        if (d == 64 && heads == 1) NGACF_CASE(bf16_t, 64, 1);
        if (d == 128 && heads == 1) NGACF_CASE(bf16_t, 128, 1);
        if (d == 128 && heads == 16) NGACF_CASE(bf16_t, 128, 16);
    }
#undef NGACF_CASE
    return NGACF_E_SHAPE;
}

extern "C" int ngacf_prop_fwd(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                              const int32_t* srow, int nshort, int nseg_user, int nshort_user,
                              const int32_t* hrow, int nhub, int nhub_user, const int32_t* rowptr,
                              const int32_t* col, const void* H, int dtype,
                              const float* S_dst, const float* S_src, const float* avec,
                              void* out, void* agg, float* R, float* part_acc, float* part_r,
                              const int32_t* part_split, int32_t* tickets,
                              const uint32_t* needed, const int32_t* rows, int nrows, int short_max,
                              int U, int N, int d, int heads, float alpha,
                              float drop_p, uint64_t seed, const uint64_t* seed_dev, int flags, void* stream) {
    return ngacf_prop_fwd_next(segs, nseg, split, nsplit, srow, nshort, nseg_user, nshort_user, hrow, nhub, nhub_user, rowptr, col, H, dtype,
                               S_dst, S_src, avec, out, agg, R, part_acc, part_r, part_split, tickets, needed, rows, nrows, short_max,
                               U, N, d, heads, alpha, drop_p, seed, seed_dev, flags, nullptr, stream);
}
