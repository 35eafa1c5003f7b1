// Attention propagation, backward (pull-only, no atomics): autograd of
// graphattention/SPUIGACF.py:359-397; formulas SURVEY.md 8(a) row A4b, regrouped ("split-sign" form):
//
//   e = exp(-lrelu(s)),  s = S[n] + S[c],  m = edge dropout multiplier,
//   ds(n,c) = -lrelu'(s) e (m (Gn[n].H[c] + Gn[c].H[n])_h - t[n] - t[c]),        dS[n,h] = sum_c ds(n,c)
//   splits into the terms that carry the row's OWN gradient and those that carry the neighbour's:
//     own:    sum_c -lrelu'(s) e (m Gn[n].H[c] - t[n])  =  (1 - alpha) (R[n] (Gn[n].aggn[n])_h - Rn[n] t[n])        =: P1[n,h]
//             (lrelu' is 1 where s > 0 and alpha elsewhere; sum_c m e H[c] = R agg and t R = Gn.(R agg) cancel; what is left is
//              the sum over the edges with s <= 0, which the train-mode forward keeps: aggn = that sum / R, Rn = the sum of
//              their e -- the second planes of agg / R, propagate.hip NegSums)
//     other:  sum_c -lrelu'(s) e (m Gn[c].H[n] - t[c])                                                  (the edge step)
//
// Step 1 (per node):  g = gout * ELU'(out);  Gn = g / R;  t = (Gn . agg)_h;  ST[n,h] = (S[n,h], t[n,h]);  dS[n,h] = P1[n,h]
// Step 2 (per directed edge, pulled by the destination row n, source c): gathers the row Gn[c] and the pair ST[c,h]
//     dS[n,h] += -e lrelu'(s) (m (Gn[c].H[n])_h - t[c])                 acc[n] += m e Gn[c]
//   epilogue: dH[n] = g[n] + acc[n] + dS[n,h] * avec[type(n)]
// One row gather and 8 bytes per head per edge (the form with Gn[n].H[c] in the edge step gathered H[c] too: 292 against 196
// bytes and three against two cache-line requests per edge at d = 64, bf16).
//
// Same two traversal modes and lane layout as the forward (propagate.hip): a lane owns a 16-byte slice of the normalised-
// gradient row Gn[c], which is stored in the same type T as H (bf16 storage halves the gather).
#include "ngacf_common.hpp"

namespace ngacf {

constexpr int UNROLL = 4;            // edges of gathers in flight per lane group (one row + one pair each); per pair of launches: 2 -> 86.0, 4 -> 81.5, 8 -> 88.4 us (4 waves / SIMD)
constexpr int UNROLL_Q = 2;          // edges in flight per lane group while a row walks its queue (row flags: 2-5 entries, typically)

struct PropBwdArgs {
    const int4* segs;
    int nseg;
    const int4* srec;              // short rows: {row, first edge, degree, 0}, sorted by degree, descending
    int nshort;
    const int* hrow;               // hub rows: one workgroup per row (see propagate.hip)
    const int* rowptr;
    const int* col;
    const void* H;
    const float* S_dst;
    const float* S_src;            // step 1 copies the node's score from here into ST
    const float* avec;
    const void* out;
    const void* agg;               // [2][N][d] in the storage type (as ngacf_prop_fwd saved it: agg, then the negative-side aggn)
    const float* R;                // [2][N][heads]: R, then Rn
    const void* gout;              // [rows][d] of the stage's storage type (the dtype autograd hands back)
    float* G;
    void* Gn;
    float2* ST;                    // [N][heads] pairs (S, t)
    void* dH;                      // [N][d] in H's storage type: its consumer (transform_bwd.hip) rounds a bf16 stage's dH to bf16 anyway
    float* dS;                     // step 1 leaves P1 here, step 2 adds the edge sums
    float* part_acc;
    float* part_r;
    const int* split;              // {row, first slot, slots} of the hub rows
    const int* part_split;         // hub row (index into split) of every partial-sum slot
    int* tickets;                  // arrival counter per hub row (zero between launches); NULL = a second kernel finishes the hub rows
    int U, N;
    int nu_lo, nu_n, ni_lo, ni_n;  // step-1 row windows: users [nu_lo, +nu_n), items [ni_lo, +ni_n) (global row = U + i)
    int gout_local;                // gout holds only the window rows, users first (node-partitioned runs)
    WorkOrder order;
    float alpha;
    uint32_t drop_thr;
    float drop_scale;
    uint64_t seed;
    const uint64_t* seed_dev;
    int flags;
    const uint32_t* needed;         // optional, one bit per row: gout is zero outside these rows (the last stage of a scored batch), so Gn, t and
                                   // P1 are zero there and only the edges whose SOURCE is flagged contribute anything, to any row
    const int* rows;               // with needed: the same rows as a list (duplicates allowed, -1 = skip): step 1 runs over it
    int nrows;
    // The fused tail (ngacf_prop_bwd_dx; fx_dX == nullptr: none): the stage's transform backward dX = mask * (dH @ W[type]^T) of the
    // rows a workgroup has just finished.  (Round 4 also ran the PREVIOUS stage's node step on that dX row in the tail: measured
    // worth nothing inside the train step -- 0.6194 against 0.6180 ms -- and deleted in round 5)
    const void* fx_panel;          // [2][d/16][din/32][64] x 16 B: bf16 A fragments of W[type]^T (ngacf_transform_fwd_pack, kind 1)
    void* fx_dX;                   // [N][din]: bf16, or fp32 (the first stage: the embedding table's gradient)
    uint32_t fx_drop_thr;          // the stage's feature dropout (0 = none)
    float fx_drop_scale;
    uint64_t fx_seed;
};

template <typename T, int D, int HEADS>
struct BShape {
    static constexpr int VEC = Slice<T>::VEC;
    static constexpr int LPR = D / VEC, EPI = WAVE / LPR, HW = D / HEADS;
    static constexpr int LPH = HW >= VEC ? HW / VEC : 1;
    static constexpr int SHIFT = D * sizeof(T) == 128 ? 7 : D * sizeof(T) == 256 ? 8 : 9;   // row bytes of H and Gn
    static_assert(LPR >= 8, "column-index broadcast uses lanes 0..7 of a slot");
};

template <typename T, int D, int HEADS>
__device__ __forceinline__ void load_row(const void* base, int c, int q, float (&f)[Slice<T>::VEC]) {
    const uint32_t off = (static_cast<uint32_t>(c) << BShape<T, D, HEADS>::SHIFT) | (static_cast<uint32_t>(q) << 4);
    Slice<T>::load(reinterpret_cast<const T*>(static_cast<const char*>(base) + off), f);
}

template <int VEC>
__device__ __forceinline__ void load_f32(const float* p, float (&f)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; j += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + j);
        f[j] = v.x; f[j + 1] = v.y; f[j + 2] = v.z; f[j + 3] = v.w;
    }
}
template <int VEC>
__device__ __forceinline__ void store_f32(float* p, const float (&f)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; j += 4) *reinterpret_cast<float4*>(p + j) = make_float4(f[j], f[j + 1], f[j + 2], f[j + 3]);
}

template <int LPR, int J>
__device__ __forceinline__ int slot_bcast_b(int v) {
    constexpr int and_mask = (~(LPR - 1)) & 0x1f;
    return __builtin_amdgcn_ds_swizzle(v, (J << 5) | and_mask);
}

// Step 1, per node.  One lane per 16-byte slice of the row.
template <typename T, int D, int HEADS>
__global__ __launch_bounds__(256) void prop_bwd_node_kernel(const PropBwdArgs a) {
    using SH = BShape<T, D, HEADS>;
    constexpr int LPR = SH::LPR, HW = SH::HW, VEC = SH::VEC;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t li = gid / LPR;                              // index inside the row windows (with a.needed: inside the batch's row list)
    const int q = (int)(gid % LPR);
    bool ok = li < (size_t)(a.needed ? a.nrows : a.nu_n + a.ni_n);
    const size_t lj = ok ? li : 0;
    size_t n = lj < (size_t)a.nu_n ? a.nu_lo + lj : (size_t)a.U + a.ni_lo + (lj - a.nu_n);
    if (a.needed) {                                          // only the batch's rows: nothing outside them is read by step 2
        const int r = a.rows[lj];
        ok = ok && r >= 0;
        n = r >= 0 ? r : 0;
    }
    const size_t nn = n;
    const int head = (q * VEC) / HW;
    float g[VEC], o[VEC], ag[VEC], an[VEC];
    if (a.flags & NGACF_GOUT_F32) load_f32<VEC>(static_cast<const float*>(a.gout) + nn * D + q * VEC, g);
    else Slice<T>::load(static_cast<const T*>(a.gout) + (a.gout_local ? lj : nn) * D + q * VEC, g);
    Slice<T>::load(static_cast<const T*>(a.agg) + nn * D + q * VEC, ag);
    Slice<T>::load(static_cast<const T*>(a.agg) + ((size_t)a.N + nn) * D + q * VEC, an);
    if (a.flags & NGACF_ELU) {
        Slice<T>::load(static_cast<const T*>(a.out) + nn * D + q * VEC, o);
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] *= o[j] > 0.f ? 1.f : o[j] + 1.f;   // ELU' = exp(pre) = out + 1 on the negative side
    }
    const float r = a.R[nn * HEADS + head], rn = a.R[((size_t)a.N + nn) * HEADS + head];
    const float inv = r > 0.f ? 1.f / r : 0.f;                // isolated node: the aggregate is a constant 0
    float gn[VEC], p = 0.f, pn = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        gn[j] = Slice<T>::round(g[j] * inv);                  // t from the stored (rounded) Gn: both directions agree
        p = fmaf(gn[j], ag[j], p);
        pn = fmaf(gn[j], an[j], pn);
    }
    p = lanes_sum<SH::LPH>(p);
    pn = lanes_sum<SH::LPH>(pn);
    if (!ok) return;
    if (a.G) store_f32<VEC>(a.G + n * D + q * VEC, g);      // G == NULL: step 2 recomputes g from gout and out (same arithmetic)
    Slice<T>::store(static_cast<T*>(a.Gn) + n * D + q * VEC, gn);
    if ((q * VEC) % HW == 0) {
        a.ST[n * HEADS + head] = make_float2(a.S_src[n * HEADS + head], p);
        a.dS[n * HEADS + head] = own_term(a.alpha, r, rn, p, pn);
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float NEG_LOG2E = -1.4426950408889634f;

// The running sums of a lane while it walks its edges: acc as the register pairs v_pk_fma_f32 works on, and the head's dS sum.
template <int VEC>
struct BSums {
    f32x2 a2[VEC / 2];
    float dsum;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int j = 0; j < VEC / 2; ++j) a2[j] = f32x2{0.f, 0.f};
        dsum = 0.f;
    }
    __device__ __forceinline__ void get(float (&acc)[VEC], float& d) const {
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = (j & 1) ? a2[j / 2].y : a2[j / 2].x;
        d = dsum;
    }
};

template <int VEC>
struct RowCtx {
    float h_own[VEC];
    float s_dst;
    uint32_t row_key, salt_h;
    bool dst_is_user;
};

template <typename T, int D, int HEADS, bool DROP>
__device__ __forceinline__ void load_ctx(const PropBwdArgs& a, int row, int q, int head, RowCtx<Slice<T>::VEC>& x) {
    x.dst_is_user = row < a.U;
    load_row<T, D, HEADS>(a.H, row, q, x.h_own);
    x.s_dst = NEG_LOG2E * a.S_dst[(size_t)row * HEADS + head];        // pre-scaled: e = exp2(min(s', alpha s')), s' = -log2e (S[dst] + S[src])
    x.salt_h = x.row_key = 0;
    if (DROP) {
        x.salt_h = salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 1u + head);
        x.row_key = x.dst_is_user ? (uint32_t)row * RNG_K1 : (uint32_t)row * RNG_K2 + x.salt_h;
    }
}

// ROOF != 0: the measurement-only instantiations (NGACF_ROOF_ADD / NGACF_ROOF_XOR, include/ngacf.h; propagate.hip gather_slice): the
// same gathers (Gn row + (S, t) pair per head), loop structure and stores, one add per gathered element / one xor per gathered dword.
template <typename T, int D, int HEADS, bool DROP, int ROOF = 0>
__device__ __forceinline__ void edge_step_b(const PropBwdArgs& a, const RowCtx<Slice<T>::VEC>& x, const float (&gv)[Slice<T>::VEC],
                                            float2 st, bool ok, int c, BSums<Slice<T>::VEC>& sm) {
    constexpr int LPH = BShape<T, D, HEADS>::LPH, VEC = Slice<T>::VEC;
    if constexpr (ROOF == 1) {
#pragma unroll
        for (int j = 0; j < VEC / 2; ++j) sm.a2[j] += f32x2{gv[2 * j], gv[2 * j + 1]};
        sm.dsum += st.x + st.y;
        return;
    } else if constexpr (ROOF == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            sm.a2[j].x = __uint_as_float(__float_as_uint(sm.a2[j].x) ^ __float_as_uint(gv[2 * j]));
            sm.a2[j].y = __uint_as_float(__float_as_uint(sm.a2[j].y) ^ __float_as_uint(gv[2 * j + 1]));
        }
        sm.dsum = __uint_as_float(__float_as_uint(sm.dsum) ^ __float_as_uint(st.x) ^ __float_as_uint(st.y));
        return;
    }
    f32x2 p = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < VEC; j += 2) p = __builtin_elementwise_fma(f32x2{gv[j], gv[j + 1]}, f32x2{x.h_own[j], x.h_own[j + 1]}, p);
    const float dot = lanes_sum<LPH>(p.x + p.y);
    const float s = fmaf(st.x, NEG_LOG2E, x.s_dst);             // -log2e (S[dst] + S[src])
    const float lr = fminf(s, a.alpha * s);
    const float ev = ok ? __builtin_amdgcn_exp2f(lr) : 0.f;
    float m = 1.f;
    if (DROP) {
        const uint32_t other = x.dst_is_user ? (uint32_t)c * RNG_K2 + x.salt_h : (uint32_t)c * RNG_K1;
        m = lowbias32(x.row_key ^ other) >= a.drop_thr ? a.drop_scale : 0.f;
    }
    // (explicit fused operations: left to -ffp-contract the dense loop and the row-flag queue walk were contracted differently,
    //  and the two must agree bit for bit)
    const float de = fmaf(m, dot, -st.y);
    sm.dsum = fmaf(-(ev * de), s < 0.f ? 1.f : a.alpha, sm.dsum);   // (s' < 0: the unscaled score is positive, lrelu' = 1)
    const float wgt = m * ev;
    const f32x2 w = {wgt, wgt};
#pragma unroll
    for (int j = 0; j < VEC / 2; ++j) sm.a2[j] = __builtin_elementwise_fma(w, f32x2{gv[2 * j], gv[2 * j + 1]}, sm.a2[j]);
}

// The gathers of one edge: the Gn row and the (S, t) pair of the lane's head.  SPARSE (hub-row workgroups, which walk ALL their
// edges): Gn / ST of a neighbour outside the batch were never written -- such an edge contributes nothing, and its two loads are
// pointed at the row's own entries instead (one cached line for all such lanes; the values are discarded).
template <typename T, int D, int HEADS, bool SPARSE, int ROOF = 0>
__device__ __forceinline__ void load_edge_b(const PropBwdArgs& a, int c, int row, int q, int head, float (&gv)[Slice<T>::VEC], float2& st) {
    const bool in = !SPARSE || flagged(a.needed, c);
    const int cg = in ? c : row;
    if constexpr (ROOF == 2) {                               // raw dwords, no unpack
        const uint32_t off = (static_cast<uint32_t>(cg) << BShape<T, D, HEADS>::SHIFT) | (static_cast<uint32_t>(q) << 4);
        const float4 v = *reinterpret_cast<const float4*>(static_cast<const char*>(a.Gn) + off);
        gv[0] = v.x; gv[1] = v.y; gv[2] = v.z; gv[3] = v.w;
#pragma unroll
        for (int j = 4; j < Slice<T>::VEC; ++j) gv[j] = 0.f;
    } else load_row<T, D, HEADS>(a.Gn, cg, q, gv);
    st = a.ST[(size_t)cg * HEADS + head];
    if (SPARSE) {
#pragma unroll
        for (int j = 0; j < Slice<T>::VEC; ++j) gv[j] = in ? gv[j] : 0.f;
        st = in ? st : make_float2(0.f, 0.f);
    }
}

// SPARSE traversal of one lane group's edges (positions ebase + k * estride, k < count): per chunk of QCAP edges,
// (1) every lane of the group tests ONE edge -- it takes part when its source is in the batch (one flag bit) -- and the columns
//     that take part are appended, in order, to the group's queue in LDS;
// (2) the queue is walked exactly like the dense loop (UNROLL_Q edges of gathers in flight).
// ~|batch| / N of a row's edges survive, whether the row itself is in the batch or not (its own gradient's terms are P1, from
// step 1): the gathers of the others are never issued.  The order of the additions is the dense kernel's, minus terms that are
// exactly zero there: same bits.
constexpr int QCAP = 64;
template <typename T, int D, int HEADS, bool DROP>
__device__ __forceinline__ void sparse_edges_b(const PropBwdArgs& a, const RowCtx<Slice<T>::VEC>& x, int* __restrict__ qbuf, int row,
                                               int ebase, int estride, int count, int slot, int q, int head,
                                               BSums<Slice<T>::VEC>& sm) {
    constexpr int LPR = BShape<T, D, HEADS>::LPR, VEC = Slice<T>::VEC;
    int mc = count;
#pragma unroll
    for (int m = LPR; m < WAVE; m <<= 1) mc = max(mc, __shfl_xor(mc, m, WAVE));
    const int maxcount = uniform(mc);
    const unsigned long long below = (1ull << q) - 1ull, group = (LPR == 64) ? ~0ull : ((1ull << LPR) - 1ull);
    for (int k0 = 0; k0 < maxcount; k0 += QCAP) {
        // all the chunk's column loads first, then all the flag loads: two round trips, not two per LPR edges
        constexpr int NS = QCAP / LPR;
        int c[NS];
        bool act[NS];
        const int rem = maxcount - k0;                          // (uniform) edges left in the longest group: rounds beyond it are skipped
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            act[i] = false; c[i] = 0;
            if (i * LPR >= rem) continue;
            const int k = k0 + i * LPR + q;
            act[i] = k < count;
            c[i] = a.col[act[i] ? ebase + k * estride : 0];
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (i * LPR >= rem) continue;
            act[i] = act[i] && flagged(a.needed, c[i]);
        }
        int n = 0;                                              // queued so far (the same in every lane of the group)
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (i * LPR >= rem) continue;
            const unsigned long long mine = (__ballot(act[i]) >> (slot * LPR)) & group;
            if (act[i]) qbuf[n + __popcll(mine & below)] = c[i];
            n += __popcll(mine);
        }
        int mn = n;
#pragma unroll
        for (int m = LPR; m < WAVE; m <<= 1) mn = max(mn, __shfl_xor(mn, m, WAVE));
        const int maxn = uniform(mn);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int b = 0; b < maxn; b += UNROLL_Q) {
            int c[UNROLL_Q];
            bool ok[UNROLL_Q];
#pragma unroll
            for (int u = 0; u < UNROLL_Q; ++u) {
                ok[u] = b + u < n;
                c[u] = ok[u] ? qbuf[b + u] : row;               // (idle: the own row, a cached line)
            }
            float gv[UNROLL_Q][VEC];
            float2 st[UNROLL_Q];
#pragma unroll
            for (int u = 0; u < UNROLL_Q; ++u) {
                load_row<T, D, HEADS>(a.Gn, c[u], q, gv[u]);
                st[u] = a.ST[(size_t)c[u] * HEADS + head];
                if (!ok[u]) {                                   // idle: Gn / ST of the own row may never have been written
#pragma unroll
                    for (int j = 0; j < VEC; ++j) gv[u][j] = 0.f;
                    st[u] = make_float2(0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL_Q; ++u)
                edge_step_b<T, D, HEADS, DROP>(a, x, gv[u], st[u], ok[u], c[u], sm);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// dS[n, h] = P1 (step 1 left it in dS) + the edge sums; dH[n] = g[n] + acc[n] + dS[n, h] * avec[type(n)]: the row's value (calc) and
// the two stores (store).  The fused launch puts the workgroup's barrier BETWEEN them: a wave that has stores in flight waits for
// their acknowledgement at the barrier (vmcnt counts stores too), which held every wave for a write round trip -- 114 against 81 us
// per launch inside the train step.
template <typename T, int D, int HEADS, int VEC>
__device__ __forceinline__ void finish_row_store(const PropBwdArgs& a, int row, int q, int head, const float (&o)[VEC], float dsum) {
    constexpr int HW = D / HEADS;
    Slice<T>::store(static_cast<T*>(a.dH) + (size_t)row * D + q * VEC, o);
    if ((q * VEC) % HW == 0) a.dS[(size_t)row * HEADS + head] = dsum;
}

template <typename T, int D, int HEADS, int VEC>
__device__ __forceinline__ void finish_row_calc(const PropBwdArgs& a, int row, int q, int head, const float (&acc)[VEC], float& dsum, float (&o)[VEC]) {
    float g[VEC];
    const bool outside = a.needed && !flagged(a.needed, row);     // outside the batch: g = 0, P1 = 0, out is not valid
    if (!outside) dsum += a.dS[(size_t)row * HEADS + head];        // P1 (step 1)
    if ((a.flags & NGACF_RESIDUAL) && a.G) load_f32<VEC>(a.G + (size_t)row * D + q * VEC, g);
    else if ((a.flags & NGACF_RESIDUAL) && outside) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] = 0.f;
    } else if (a.flags & NGACF_RESIDUAL) {         // g = gout * ELU'(out), exactly as step 1 computed it (no 4-byte G table)
        if (a.flags & NGACF_GOUT_F32) load_f32<VEC>(static_cast<const float*>(a.gout) + (size_t)row * D + q * VEC, g);
        else Slice<T>::load(static_cast<const T*>(a.gout) + (size_t)row * D + q * VEC, g);
        if (a.flags & NGACF_ELU) {
            Slice<T>::load(static_cast<const T*>(a.out) + (size_t)row * D + q * VEC, o);
#pragma unroll
            for (int j = 0; j < VEC; ++j) g[j] *= o[j] > 0.f ? 1.f : o[j] + 1.f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] = 0.f;
    }
    const float* ao = a.avec + (row < a.U ? 0 : D) + q * VEC;      // the node's own type
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = g[j] + acc[j] + dsum * ao[j];
}

template <typename T, int D, int HEADS, int VEC>
__device__ __forceinline__ void finish_row_b(const PropBwdArgs& a, int row, int q, int head, const float (&acc)[VEC], float dsum) {
    float o[VEC];
    finish_row_calc<T, D, HEADS, VEC>(a, row, q, head, acc, dsum, o);
    finish_row_store<T, D, HEADS, VEC>(a, row, q, head, o, dsum);
}

// What a lane group hands to the fused tail: the dH row it has finished (this lane's 16-byte slice, fp32 before the store's rounding;
// the tail stores it, behind its barrier) and the row's dS entry.
template <int VEC>
struct RowDone {
    float o[VEC];
    float dsum;
    int row;
    bool valid;
};

template <typename T, int D, int HEADS, int FUSE, int VEC>
__device__ __forceinline__ void finish_row_t(const PropBwdArgs& a, int row, int q, int head, const float (&acc)[VEC], float dsum, RowDone<VEC>& rd) {
    if constexpr (FUSE == 0) {
        finish_row_b<T, D, HEADS, VEC>(a, row, q, head, acc, dsum);
    } else {
        finish_row_calc<T, D, HEADS, VEC>(a, row, q, head, acc, dsum, rd.o);
        rd.dsum = dsum;
        rd.row = row;
        rd.valid = true;
    }
}

// The traversal of one wave's work item(s).  FUSE != 0 (ngacf_prop_bwd_dx): the row this lane's group finished is also handed
// back in `rd` for the workgroup's tail, and no path leaves the KERNEL before that tail (the returns below leave this function).
template <typename T, int D, int HEADS, bool DROP, bool SPARSE, int FUSE, int ROOF = 0>
__device__ __forceinline__ void prop_bwd_body(const PropBwdArgs& a, int kind, int widx, int wend, int* qmem, RowDone<Slice<T>::VEC>& rd) {
    using SH = BShape<T, D, HEADS>;
    constexpr int UNR = UNROLL;
    constexpr int LPR = SH::LPR, EPI = SH::EPI, HW = SH::HW, VEC = SH::VEC;
    const int lane = threadIdx.x & 63;
    const int slot = lane / LPR, q = lane % LPR;
    const int head = (q * VEC) / HW;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    float dsum = 0.f;
    BSums<VEC> sm;                                          // the running sums inside the edge loops; acc / dsum take them over behind
    sm.zero();
    RowCtx<VEC> x;
    int* qbuf = qmem + (SPARSE ? ((threadIdx.x >> 6) * EPI + slot) * (QCAP + 1) : 0);
    if (kind == WORK_HUB) {
        // ------------------------------------------------------------------ hub rows: workgroup per row (as in the forward)
        __shared__ float hub_acc[3][D];
        __shared__ float hub_d[3][HEADS];
        const int wv = threadIdx.x >> 6;
        const int row = uniform(a.hrow[widx]);
        const int beg = uniform(a.rowptr[row]), end = uniform(a.rowptr[row + 1]);
        load_ctx<T, D, HEADS, DROP>(a, row, q, head, x);
        for (int base = beg + wv * EPI * UNR; base < end; base += 4 * EPI * UNR) {
            int c[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int e = base + u * EPI + slot;
                ok[u] = e < end;
                c[u] = a.col[ok[u] ? e : end - 1];
            }
            float gv[UNR][VEC];
            float2 st[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u)                 // (SPARSE: every edge is visited; the ones from outside the batch add exact zeros)
                load_edge_b<T, D, HEADS, SPARSE, ROOF>(a, c[u], row, q, head, gv[u], st[u]);
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                edge_step_b<T, D, HEADS, DROP, ROOF>(a, x, gv[u], st[u], ok[u], c[u], sm);
        }
        sm.get(acc, dsum);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
        dsum = group_sum<LPR, WAVE>(dsum);
        if (wv > 0 && slot == 0) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) hub_acc[wv - 1][q * VEC + j] = acc[j];
            if ((q * VEC) % HW == 0) hub_d[wv - 1][head] = dsum;
        }
        __syncthreads();
        if (wv == 0 && slot == 0) {
#pragma unroll
            for (int w2 = 0; w2 < 3; ++w2) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += hub_acc[w2][q * VEC + j];
                dsum += hub_d[w2][head];
            }
            finish_row_b<T, D, HEADS, VEC>(a, row, q, head, acc, dsum);
        }
        return;
    }
    if (widx >= wend) return;
    if (kind == WORK_SHORT) {
        // ------------------------------------------------------------------ short rows: slot per row
        const int ridx = widx + slot;
        const int4 rec = a.srec[ridx < wend ? ridx : wend - 1];
        const bool live = ridx < wend && rec.x >= 0;         // row -1 = padding between the user and item lists
        const int row = live ? rec.x : 0;
        const int beg = live ? rec.y : 0;
        const int deg = live ? rec.z : 0;
        const int maxdeg = uniform(deg);                     // rows are sorted: slot 0 holds the longest
        load_ctx<T, D, HEADS, DROP>(a, row, q, head, x);
        if (SPARSE) {                                        // only the edges out of the batch take part (the queue)
            sparse_edges_b<T, D, HEADS, DROP>(a, x, qbuf, row, beg, 1, deg, slot, q, head, sm);
            sm.get(acc, dsum);
            if (live) finish_row_t<T, D, HEADS, FUSE, VEC>(a, row, q, head, acc, dsum, rd);
            return;
        }
        // (The forward's two-batches-in-flight loop was tried here too, propagate.hip: the stripped instantiations gain 2-5 %, the
        //  kernel with dropout -- the one a train step runs -- LOSES 1.5 us alone and the two-chain step 2 %: it holds a second batch
        //  of rows and (S, t) pairs on top of the row's own H slice, at the same four waves per SIMD.  The r4 loop stays.)
        const int last = deg > 0 ? beg + deg - 1 : 0;
        int colv = a.col[min(beg + (q & 7), last)];
        for (int k0 = 0; k0 < maxdeg; k0 += 8) {
            const int coln = a.col[min(beg + k0 + 8 + (q & 7), last)];
            int c[8];
            c[0] = slot_bcast_b<LPR, 0>(colv); c[1] = slot_bcast_b<LPR, 1>(colv);
            c[2] = slot_bcast_b<LPR, 2>(colv); c[3] = slot_bcast_b<LPR, 3>(colv);
            c[4] = slot_bcast_b<LPR, 4>(colv); c[5] = slot_bcast_b<LPR, 5>(colv);
            c[6] = slot_bcast_b<LPR, 6>(colv); c[7] = slot_bcast_b<LPR, 7>(colv);
#pragma unroll
            for (int h0 = 0; h0 < 8; h0 += UNR) {
                if (h0 > 0 && k0 + h0 >= maxdeg) break;
                float gv[UNR][VEC];
                float2 st[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    load_edge_b<T, D, HEADS, false, ROOF>(a, c[h0 + u], row, q, head, gv[u], st[u]);
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    edge_step_b<T, D, HEADS, DROP, ROOF>(a, x, gv[u], st[u], k0 + h0 + u < deg, c[h0 + u], sm);
            }
            colv = coln;
        }
        sm.get(acc, dsum);
        if (live) finish_row_t<T, D, HEADS, FUSE, VEC>(a, row, q, head, acc, dsum, rd);
        return;
    }

    // ---------------------------------------------------------------------- long rows: wave per segment
    const int w = uniform(widx);
    const int4 sg = a.segs[w];
    const int row = uniform(sg.x), beg = uniform(sg.y), end = uniform(sg.z), part = uniform(sg.w);
    load_ctx<T, D, HEADS, DROP>(a, row, q, head, x);
    if (SPARSE) {                                   // the lane group's edges: beg + slot, + EPI, ... (the dense loop's assignment)
        const int mine = end - beg - slot;
        sparse_edges_b<T, D, HEADS, DROP>(a, x, qbuf, row, beg + slot, EPI, mine > 0 ? (mine + EPI - 1) / EPI : 0, slot, q, head, sm);
    } else {
        for (int base = beg; base < end; base += EPI * UNR) {
            int c[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int e = base + u * EPI + slot;
                ok[u] = e < end;
                c[u] = a.col[ok[u] ? e : end - 1];
            }
            float gv[UNR][VEC];
            float2 st[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                load_edge_b<T, D, HEADS, false, ROOF>(a, c[u], row, q, head, gv[u], st[u]);
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                edge_step_b<T, D, HEADS, DROP, ROOF>(a, x, gv[u], st[u], ok[u], c[u], sm);
        }
    }
    sm.get(acc, dsum);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
    dsum = group_sum<LPR, WAVE>(dsum);
    if (part < 0) {
        if (slot == 0) finish_row_t<T, D, HEADS, FUSE, VEC>(a, row, q, head, acc, dsum, rd);
        return;
    }
    // slice of a hub row: leave the partial sums in slot `part`
    if (a.tickets == nullptr) {                            // prop_bwd_split_kernel finishes the row
        if (slot == 0) {
            store_f32<VEC>(a.part_acc + (size_t)part * D + q * VEC, acc);
            if ((q * VEC) % HW == 0) a.part_r[(size_t)part * HEADS + head] = dsum;
        }
        return;
    }
    // The slice that arrives LAST finishes the row in this launch -- the forward's protocol (propagate.hip): write-through stores
    // of the partial sums, drain, relaxed agent-scope ticket; the last arriver reads all partials with agent-scope atomic
    // loads (no acquire fence: it would invalidate this XCD's L2) and adds them in slot order.  Nobody waits for anybody.
    if (slot == 0) {
        unsigned long long* pa = reinterpret_cast<unsigned long long*>(a.part_acc + (size_t)part * D + q * VEC);
#pragma unroll
        for (int j = 0; j < VEC; j += 2) {
            const unsigned long long bits = (unsigned long long)__float_as_uint(acc[j]) | ((unsigned long long)__float_as_uint(acc[j + 1]) << 32);
            __hip_atomic_store(pa + j / 2, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((q * VEC) % HW == 0)
            __hip_atomic_store(reinterpret_cast<unsigned*>(a.part_r + (size_t)part * HEADS + head), __float_as_uint(dsum),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int si = uniform(a.part_split[part]);
    const int p0 = uniform(a.split[3 * si + 1]), np = uniform(a.split[3 * si + 2]);
    int tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add(a.tickets + si, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (uniform(tk) != np - 1) return;
    if (lane == 0) __hip_atomic_store(a.tickets + si, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    dsum = 0.f;
#pragma unroll 4
    for (int p = p0 + slot; p < p0 + np; p += EPI) {
        unsigned long long* pa = reinterpret_cast<unsigned long long*>(a.part_acc + (size_t)p * D + q * VEC);
#pragma unroll
        for (int j = 0; j < VEC; j += 2) {
            const unsigned long long bits = __hip_atomic_load(pa + j / 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc[j] += __uint_as_float((unsigned)bits); acc[j + 1] += __uint_as_float((unsigned)(bits >> 32));
        }
        dsum += __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned*>(a.part_r + (size_t)p * HEADS + head), __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT));
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
    dsum = group_sum<LPR, WAVE>(dsum);
    if (slot == 0) finish_row_t<T, D, HEADS, FUSE, VEC>(a, row, q, head, acc, dsum, rd);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The fused tail (ngacf_prop_bwd_dx), the mirror image of the forward's next-stage transform (propagate.hip): the rows THIS
// WORKGROUP has just finished go through the stage's transform backward while they are on chip,
//     dX[n, :] = mask * (dH[n, :] @ W[type(n)]^T)                    (transform_bwd.hip: the dX half of transform_dw_mfma_kernel)
//   1. every wave writes its finished dH rows -- the bf16 values the dH table stores -- into the workgroup's 32-row LDS tile;
//   2. ONE barrier; waves 1..3 leave;
//   3. wave 0: DKS x DOT v_mfma_f32_32x32x16_bf16 with the node on the lane, k ascending -- the arithmetic, order and bits of
//      transform_dw_mfma_kernel<WITH_DX> -- then mask, rounding, the half-waves' exchange (bf16 rows: one 16-byte store per lane).
// FUSE 1: fp32 dX rows (the first stage: the embedding table's gradient); FUSE 3: bf16 dX rows.
// A launch's workgroups hold rows of one node type (work order), with row flags too: one panel per workgroup.
// ---------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bw_bf16x8;
typedef __attribute__((ext_vector_type(16))) float bw_f32x16;
constexpr int GROW_U4 = 9;           // a staged row: 128 B of bf16 + 16 B pad (36-dword stride: 16 rows hit 64 distinct banks)

template <typename T, int HEADS, int D, int DIN, int FUSE>
__device__ __forceinline__ void bwd_dx_tail(const PropBwdArgs& a, uint4* gtile, int gwave, int* xmeta, int primary, const RowDone<8>& rd, int slot, int q) {
    static_assert(D == 64 && DIN == 64, "fused dX: 64 -> 64 stages");
    constexpr int DKS = D / 16, DOT = DIN / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tr = 8 * wave + slot;
    if (rd.valid) {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = f32_to_bf16(rd.o[2 * j]) | (f32_to_bf16(rd.o[2 * j + 1]) << 16);
        gtile[wave * gwave + slot * GROW_U4 + q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (q == 0) xmeta[tr] = rd.valid ? rd.row : -1;
    __syncthreads();
    // The weight panel (8 KB of bf16 A fragments, L2-resident) goes straight into wave 0's registers, requested in front of the
    // wave's own stores: the MFMAs then wait for these loads only.  (As an LDS-DMA image it cost every workgroup 8 KB of LDS for
    // its whole life: two of these launches share the CUs in a train step, and with row flags that halved the workgroups in flight.)
    bw_bf16x8 af[DKS * DOT];
    if (wave == 0) {
        const uint4* src = reinterpret_cast<const uint4*>(a.fx_panel) + (size_t)primary * DKS * DOT * 64;
#pragma unroll
        for (int i = 0; i < DKS * DOT; ++i) af[i] = __builtin_bit_cast(bw_bf16x8, src[i * 64 + lane]);
    }
    if (rd.valid) finish_row_store<T, D, HEADS, 8>(a, rd.row, q, (q * 8) / (D / HEADS), rd.o, rd.dsum);
    // Wave 0 alone runs the tile step.  (Sharing it among the four waves by output columns -- each its own 4-MFMA chain and two of
    // the eight pieces -- was measured and lost: 74-76 against 69-73 us per dense launch, 53 against 52 with row flags: four times
    // the fragment reads and twice the MFMAs for a tail that is not what the launch waits for.)
    if (wave != 0) return;
    const int r = lane & 31, hh = lane >> 5;
    const int nrow = xmeta[r];
    const bool ok = nrow >= 0;
    const int node = ok ? nrow : 0;
    bw_bf16x8 bf[DKS];
#pragma unroll
    for (int sk = 0; sk < DKS; ++sk) bf[sk] = __builtin_bit_cast(bw_bf16x8, gtile[(r >> 3) * gwave + (r & 7) * GROW_U4 + 2 * sk + hh]);
    const uint32_t salt = a.fx_drop_thr ? salt_of(a.fx_seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u) : 0u;
    {
        const bool mine = ok;
#pragma unroll
        for (int t = 0; t < DOT; ++t) {
            bw_f32x16 dacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) dacc[e] = 0.f;
#pragma unroll
            for (int sk = 0; sk < DKS; ++sk)
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sk * DOT + t], bf[sk], dacc, 0, 0, 0);
            // lane: node r (+ half hh); dacc[4g + j] = dX column 32t + 8g + 4hh + j
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                float v[2][4];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[gg][j] = dacc[4 * (gp + gg) + j];
                        if (a.fx_drop_thr)
                            v[gg][j] = feat_bits(salt, node, 32 * t + 8 * (gp + gg) + 4 * hh + j) >= a.fx_drop_thr ? v[gg][j] * a.fx_drop_scale : 0.f;
                    }
                if constexpr (FUSE == 1) {                    // fp32 rows (the embedding table's gradient): no previous stage
                    if (mine) {
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg)
                            *reinterpret_cast<float4*>(static_cast<float*>(a.fx_dX) + (size_t)node * DIN + 32 * t + 8 * (gp + gg) + 4 * hh) =
                                make_float4(v[gg][0], v[gg][1], v[gg][2], v[gg][3]);
                    }
                } else {
                    uint2 pk[2];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg)
                        pk[gg] = make_uint2(f32_to_bf16(v[gg][0]) | (f32_to_bf16(v[gg][1]) << 16), f32_to_bf16(v[gg][2]) | (f32_to_bf16(v[gg][3]) << 16));
                    const auto rx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                    const auto ry = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                    // this lane now holds piece pc = 4t + gp + hh of its node's dX row: columns 8 pc .. 8 pc + 7
                    const uint32_t w[4] = {rx[0], ry[0], rx[1], ry[1]};
                    const int pc = 4 * t + gp + hh;
                    if (mine) *reinterpret_cast<uint4*>(static_cast<char*>(a.fx_dX) + (size_t)node * DIN * 2 + 16 * pc) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
    }
}

template <typename T, int D, int HEADS, bool DROP, bool SPARSE, int FUSE = 0, int ROOF = 0>
__global__ __launch_bounds__(256, FUSE != 0 ? 4 : (SPARSE ? (DROP ? 5 : 4) : 1)) void prop_bwd_kernel(const PropBwdArgs a) {
    static_assert(ROOF == 0 || (FUSE == 0 && !DROP && !SPARSE), "measurement-only instantiations: the dense edge backward alone");
    // (FUSE: 4 waves per SIMD = 128 VGPRs.  Bounded to 5 = 96 VGPRs the 8-head dropout instantiations spill one or two registers to
    //  scratch, and the one with row flags then faulted on the GPU; 5 against 4 waves per SIMD is worth ~1 % in this kernel.)
    using SH = BShape<T, D, HEADS>;
    constexpr int EPI = SH::EPI, VEC = SH::VEC;
    int kind, widx, wend;
    locate(a.order, blockIdx.x, threadIdx.x >> 6, EPI, kind, widx, wend);
    __shared__ __attribute__((aligned(16))) int qmem[SPARSE ? 4 * EPI * (QCAP + 1) : 4];   // SPARSE: a column queue per lane group (odd stride: no bank clash)
    RowDone<VEC> rd;
    rd.valid = false;
    rd.row = 0;
    if constexpr (FUSE != 0) {
        static_assert(sizeof(T) == 2 && D == 64, "fused dX: bf16 rows of 64 columns");
        constexpr int DIN = 64;
        // the finished dH rows: 8 rows of 144 B per wave.  With row flags they take over the wave's OWN column queues (2 080 B per
        // wave, dead once the wave has finished its rows -- and only the wave itself ever touched them)
        constexpr int QWAVE_U4 = EPI * (QCAP + 1) * 4 / 16;                 // a wave's queues in uint4 units (130)
        static_assert(!SPARSE || (EPI * (QCAP + 1) * 4) % 16 == 0, "queue region of a wave: whole uint4s");
        static_assert(!SPARSE || QWAVE_U4 >= 8 * GROW_U4, "a wave's rows fit its queues");
        __shared__ __attribute__((aligned(16))) uint4 gtile_own[SPARSE ? 1 : 32 * GROW_U4];
        uint4* gtile_lds = SPARSE ? reinterpret_cast<uint4*>(qmem) : gtile_own;
        constexpr int gwave = SPARSE ? QWAVE_U4 : 8 * GROW_U4;
        __shared__ int xmeta_lds[32];
        const int lane = threadIdx.x & 63;
        const int primary = (int)blockIdx.x < a.order.b2 ? 0 : 1;       // (work order: user rows first)
        prop_bwd_body<T, D, HEADS, DROP, SPARSE, FUSE>(a, kind, widx, wend, qmem, rd);
        bwd_dx_tail<T, HEADS, D, DIN, FUSE>(a, gtile_lds, gwave, xmeta_lds, primary, rd, lane / SH::LPR, lane % SH::LPR);
    } else {
        prop_bwd_body<T, D, HEADS, DROP, SPARSE, 0, ROOF>(a, kind, widx, wend, qmem, rd);
    }
}


template <typename T, int D, int HEADS>
__global__ __launch_bounds__(256) void prop_bwd_split_kernel(const PropBwdArgs a, const int* __restrict__ split, int nsplit) {
    using SH = BShape<T, D, HEADS>;
    constexpr int LPR = SH::LPR, EPI = SH::EPI, HW = SH::HW, VEC = SH::VEC;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nsplit) return;
    const int row = split[3 * w], p0 = split[3 * w + 1], np = split[3 * w + 2];
    const int slot = lane / LPR, q = lane % LPR, head = (q * VEC) / HW;
    float acc[VEC], dsum = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int p = p0 + slot; p < p0 + np; p += EPI) {
        float v[VEC];
        load_f32<VEC>(a.part_acc + (size_t)p * D + q * VEC, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += v[j];
        dsum += a.part_r[(size_t)p * HEADS + head];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
    dsum = group_sum<LPR, WAVE>(dsum);
    if (slot != 0) return;
    finish_row_b<T, D, HEADS, VEC>(a, row, q, head, acc, dsum);
}

template <typename T, int D, int HEADS>
static int launch_bwd(PropBwdArgs a, const int* split, int nsplit, bool drop, hipStream_t st, bool do_node, bool do_edges, int fuse = 0) {
    constexpr int EPI = BShape<T, D, HEADS>::EPI;
    if (do_node && (a.needed ? a.nrows : a.nu_n + a.ni_n) > 0) {
        const size_t threads = (size_t)(a.needed ? a.nrows : a.nu_n + a.ni_n) * BShape<T, D, HEADS>::LPR;
        hipLaunchKernelGGL((prop_bwd_node_kernel<T, D, HEADS>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a);
        NGACF_CHECK_LAUNCH();
    }
    if (!do_edges) return 0;
    int nb = 0;
    a.order = make_order(a.nseg, a.order.nseg_u, a.nshort, a.order.nshort_u, a.order.nhub, a.order.nhub_u, EPI, &nb);
    if (fuse != 0) {                    // ngacf_prop_bwd_dx: the transform's dX (and the previous stage's step 1) in the tail
        if constexpr (sizeof(T) == 2 && D == 64) {
            if (nb == 0) return 0;
            const dim3 grid(nb), block(256);
#define NGACF_FX(F)                                                                                                          \
            do {                                                                                                             \
                if (a.needed && drop) launch_k(NGACF_TAG_PROP_BWD, prop_bwd_kernel<T, D, HEADS, true, true, F>, grid, block, 0, st, a);    \
                else if (a.needed)    launch_k(NGACF_TAG_PROP_BWD, prop_bwd_kernel<T, D, HEADS, false, true, F>, grid, block, 0, st, a);   \
                else if (drop)        launch_k(NGACF_TAG_PROP_BWD, prop_bwd_kernel<T, D, HEADS, true, false, F>, grid, block, 0, st, a);   \
                else                  launch_k(NGACF_TAG_PROP_BWD, prop_bwd_kernel<T, D, HEADS, false, false, F>, grid, block, 0, st, a);  \
            } while (0)
            if (fuse == 1) NGACF_FX(1);
            else NGACF_FX(3);
#undef NGACF_FX
            NGACF_CHECK_LAUNCH();
            return 0;
        } else {
            return NGACF_E_SHAPE;
        }
    }
    if (a.flags & (NGACF_ROOF_ADD | NGACF_ROOF_XOR)) {          // measurement only (tools/gather_roof.py)
        if constexpr (D == 64) {
            if (a.needed || drop || (nsplit > 0 && a.tickets == nullptr)) return NGACF_E_ARG;
            if (nb == 0) return 0;
            if (a.flags & NGACF_ROOF_XOR) launch_k(NGACF_TAG_PROP_BWD, prop_bwd_kernel<T, D, HEADS, false, false, 0, 2>, dim3(nb), dim3(256), 0, st, a);
            else                          launch_k(NGACF_TAG_PROP_BWD, prop_bwd_kernel<T, D, HEADS, false, false, 0, 1>, dim3(nb), dim3(256), 0, st, a);
            NGACF_CHECK_LAUNCH();
            return 0;
        } else {
            return NGACF_E_SHAPE;
        }
    }
    if (nb > 0) {
        const dim3 grid(nb), block(256);
        if (a.needed) {
            if (drop) launch_k(NGACF_TAG_PROP_BWD, prop_bwd_kernel<T, D, HEADS, true, true>, grid, block, 0, st, a);
            else      launch_k(NGACF_TAG_PROP_BWD, prop_bwd_kernel<T, D, HEADS, false, true>, grid, block, 0, st, a);
        } else {
            if (drop) launch_k(NGACF_TAG_PROP_BWD, prop_bwd_kernel<T, D, HEADS, true, false>, grid, block, 0, st, a);
            else      launch_k(NGACF_TAG_PROP_BWD, prop_bwd_kernel<T, D, HEADS, false, false>, grid, block, 0, st, a);
        }
        NGACF_CHECK_LAUNCH();
    }
    if (nsplit > 0 && a.tickets == nullptr) {
        hipLaunchKernelGGL((prop_bwd_split_kernel<T, D, HEADS>), dim3((nsplit + 3) / 4), dim3(256), 0, st, a, split, nsplit);
        NGACF_CHECK_LAUNCH();
    }
    return 0;
}

}  // namespace ngacf

using namespace ngacf;

static int prop_bwd_impl(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                         const int32_t* srow, int nshort, int nseg_user, int nshort_user,
                         const int32_t* hrow, int nhub, int nhub_user, const int32_t* rowptr,
                         const int32_t* col, const void* H, int dtype,
                         const float* S_dst, const float* S_src, const float* avec,
                         const void* out, const void* agg, const float* R, const void* gout,
                         float* G, void* Gn, float* ST, void* dH, float* dS,
                         float* part_acc, float* part_r, const int32_t* part_split, int32_t* tickets,
                         int U, int N, int d, int heads, float alpha,
                         float drop_p, uint64_t seed, const uint64_t* seed_dev, int flags,
                         int nu_lo, int nu_n, int ni_lo, int ni_n, int gout_local, bool do_node, bool do_edges,
                         const uint32_t* needed, const int32_t* rows, int nrows, void* stream,
                         const ngacf_bwd_dx* dx = nullptr) {
    if (nseg < 0 || nsplit < 0 || nshort < 0 || U < 0 || N < U) return NGACF_E_ARG;
    if (nseg_user < 0 || nseg_user > nseg || nshort_user < 0 || nshort_user > nshort || (nshort_user % 16 != 0 && nshort_user != nshort))
        return NGACF_E_ARG;
    if (nhub < 0 || nhub_user < 0 || nhub_user > nhub) return NGACF_E_ARG;
    if (nu_lo < 0 || nu_n < 0 || ni_lo < 0 || ni_n < 0 || nu_lo + nu_n > U || ni_lo + ni_n > N - U) return NGACF_E_ARG;
    if (N == 0) return 0;
    if (!Gn || !ST || !dS) return NGACF_E_ARG;
    if (!G && !(do_edges && !gout_local && out && gout)) return NGACF_E_ARG;   // without G the edge step recomputes g from global-row gout and out
    if (do_node && (!out || !agg || !R || !gout || !S_src)) return NGACF_E_ARG;
    if (do_edges) {
        if (!col || !rowptr || !H || !S_dst || !avec || !dH) return NGACF_E_ARG;
        if ((nseg > 0 && !segs) || (nshort > 0 && !srow) || (nhub > 0 && !hrow)) return NGACF_E_ARG;
        if (nsplit > 0 && (!split || !part_acc || !part_r)) return NGACF_E_ARG;
    }
    if (!(drop_p >= 0.f && drop_p < 1.f) || !(alpha > 0.f && alpha < 1.f)) return NGACF_E_ARG;
    if ((uint64_t)N * (uint64_t)d * (dtype == NGACF_BF16 ? 2ull : 4ull) >= (1ull << 32)) return NGACF_E_SHAPE;   // 32-bit row offsets
    PropBwdArgs a;
    a.segs = reinterpret_cast<const int4*>(segs); a.nseg = nseg; a.srec = reinterpret_cast<const int4*>(srow); a.nshort = nshort; a.rowptr = rowptr;
    a.hrow = hrow; a.order.nhub = nhub; a.order.nhub_u = nhub_user;
    a.col = col; a.H = H; a.S_dst = S_dst; a.S_src = S_src; a.avec = avec; a.out = out; a.agg = agg; a.R = R; a.gout = gout;
    a.G = G; a.Gn = Gn; a.ST = reinterpret_cast<float2*>(ST); a.dH = dH; a.dS = dS; a.part_acc = part_acc; a.part_r = part_r;
    a.split = split; a.part_split = part_split; a.tickets = (part_split && nsplit > 0) ? tickets : nullptr;
    a.U = U; a.N = N; a.nu_lo = nu_lo; a.nu_n = nu_n; a.ni_lo = ni_lo; a.ni_n = ni_n; a.gout_local = gout_local;
    a.order.nseg_u = nseg_user; a.order.nshort_u = nshort_user; a.alpha = alpha;
    a.drop_thr = drop_threshold(drop_p); a.drop_scale = 1.f / (1.f - drop_p); a.seed = seed; a.seed_dev = seed_dev; a.flags = flags;
    if (needed && (nrows < 0 || (nrows > 0 && !rows))) return NGACF_E_ARG;
    a.needed = needed; a.rows = rows; a.nrows = nrows;
    if (needed && G) return NGACF_E_ARG;                       // (the G table is a node-partitioned run's; not with row flags)
    const bool drop = drop_p > 0.f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    a.fx_panel = nullptr; a.fx_dX = nullptr; a.fx_drop_thr = 0; a.fx_drop_scale = 1.f; a.fx_seed = 0;
    int fuse = 0;
    if (dx) {
        if (!dx->panel || !dx->dX) return NGACF_E_ARG;
        if (!do_edges || G || gout_local || nhub > 0) return NGACF_E_ARG;
        if (nsplit > 0 && a.tickets == nullptr) return NGACF_E_ARG;             // hub rows must be finished inside the launch
        if (dtype != NGACF_BF16 || d != 64 || dx->din != 64) return NGACF_E_SHAPE;
        if (nu_lo != 0 || nu_n != U || ni_lo != 0 || ni_n != N - U) return NGACF_E_ARG;
        if (!(dx->feat_drop_p >= 0.f && dx->feat_drop_p < 1.f)) return NGACF_E_ARG;
        if (dx->dx_dtype != NGACF_BF16 && dx->dx_dtype != NGACF_F32) return NGACF_E_SHAPE;
        a.fx_panel = dx->panel; a.fx_dX = dx->dX;
        a.fx_drop_thr = dx->feat_drop_p > 0.f ? drop_threshold(dx->feat_drop_p) : 0u;
        a.fx_drop_scale = 1.f / (1.f - dx->feat_drop_p); a.fx_seed = dx->seed;
        fuse = dx->dx_dtype == NGACF_F32 ? 1 : 3;
    }
#define NGACF_CASE(T, D, HD) return launch_bwd<T, D, HD>(a, split, nsplit, drop, st, do_node, do_edges, fuse)
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

extern "C" int ngacf_prop_bwd(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                              const int32_t* srow, int nshort, int nseg_user, int nshort_user,
                              const int32_t* hrow, int nhub, int nhub_user, const int32_t* rowptr,
                              const int32_t* col, const void* H, int dtype,
                              const float* S_dst, const float* S_src, const float* avec,
                              const void* out, const void* agg, const float* R, const void* gout,
                              float* G, void* Gn, float* ST, void* dH, float* dS,
                              float* part_acc, float* part_r, const int32_t* part_split, int32_t* tickets,
                              const uint32_t* needed, const int32_t* rows, int nrows,
                              int U, int N, int d, int heads, float alpha,
                              float drop_p, uint64_t seed, const uint64_t* seed_dev, int flags, void* stream) {
    const bool skip_node = (flags & NGACF_SKIP_NODE) != 0;
    return prop_bwd_impl(segs, nseg, split, nsplit, srow, nshort, nseg_user, nshort_user, hrow, nhub, nhub_user, rowptr, col, H, dtype, S_dst, S_src, avec, out, agg, R, gout,
                         G, Gn, ST, dH, dS, part_acc, part_r, part_split, tickets, U, N, d, heads, alpha, drop_p, seed, seed_dev, flags,
                         0, U, 0, N - U, 0, !skip_node, true, needed, rows, nrows, stream);
}

extern "C" int ngacf_prop_bwd_dx(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                                 const int32_t* srow, int nshort, int nseg_user, int nshort_user,
                                 const int32_t* hrow, int nhub, int nhub_user, const int32_t* rowptr,
                                 const int32_t* col, const void* H, int dtype,
                                 const float* S_dst, const float* S_src, const float* avec,
                                 const void* out, const void* agg, const float* R, const void* gout,
                                 float* G, void* Gn, float* ST, void* dH, float* dS,
                                 float* part_acc, float* part_r, const int32_t* part_split, int32_t* tickets,
                                 const uint32_t* needed, const int32_t* rows, int nrows,
                                 int U, int N, int d, int heads, float alpha,
                                 float drop_p, uint64_t seed, const uint64_t* seed_dev, int flags,
                                 const ngacf_bwd_dx* dx, void* stream) {
    const bool skip_node = (flags & NGACF_SKIP_NODE) != 0;
    return prop_bwd_impl(segs, nseg, split, nsplit, srow, nshort, nseg_user, nshort_user, hrow, nhub, nhub_user, rowptr, col, H, dtype, S_dst, S_src, avec, out, agg, R, gout,
                         G, Gn, ST, dH, dS, part_acc, part_r, part_split, tickets, U, N, d, heads, alpha, drop_p, seed, seed_dev, flags,
                         0, U, 0, N - U, 0, !skip_node, true, needed, rows, nrows, stream, dx);
}

extern "C" int ngacf_prop_bwd_node(const void* out, int dtype, const void* agg, const float* R, const void* gout,
                                   const float* S_src, float* G, void* Gn, float* ST, float* dS,
                                   int U, int N, int d, int heads, float alpha, int flags,
                                   int u_lo, int u_n, int i_lo, int i_n, int gout_local, void* stream) {
    return prop_bwd_impl(nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, dtype, nullptr, S_src, nullptr,
                         out, agg, R, gout, G, Gn, ST, nullptr, dS, nullptr, nullptr, nullptr, nullptr, U, N, d, heads, alpha, 0.f, 0, nullptr, flags,
                         u_lo, u_n, i_lo, i_n, gout_local, true, false, nullptr, nullptr, 0, stream);
}
