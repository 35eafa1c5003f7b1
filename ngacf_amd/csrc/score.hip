// Pair scoring (SPUIGACF.py:49-52), BPR loss (BPRLoss.py:8-9) and the dropout-RNG probe.
#include "ngacf_common.hpp"

namespace ngacf {

// LPR lanes per pair: each lane multiplies a 16-byte slice of the user row with the item row's.
template <typename T, int D>
__global__ __launch_bounds__(256) void score_fwd_kernel(const T* __restrict__ F, const int64_t* __restrict__ u,
                                                        const int64_t* __restrict__ it, float* __restrict__ score,
                                                        int* __restrict__ bad, int B, int U, int I) {
    constexpr int VEC = Slice<T>::VEC, LPR = D / VEC, PPW = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = wave * PPW + lane / LPR, q = lane % LPR;
    const bool ok = b < B;
    const int bb = ok ? b : B - 1;
    // the reference raises on an out-of-range index (SPUIGACF.py:49-50); here such a pair never becomes an address:
    // its score is NaN and it is counted in *bad (ops.raise_if_bad_indices)
    const int64_t ub = u[bb], ib = it[bb];
    const bool in = ub >= 0 && ub < U && ib >= 0 && ib < I;
    float x[VEC], y[VEC];
    Slice<T>::load(F + (size_t)(in ? ub : 0) * D + q * VEC, x);
    Slice<T>::load(F + (size_t)(in ? ib + U : 0) * D + q * VEC, y);
    float p = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) p = fmaf(x[j], y[j], p);
    p = group_sum<1, LPR>(p);
    if (ok && q == 0) {
        score[b] = in ? p : __int_as_float(0x7fc00000);
        if (!in && bad) atomicAdd(bad, 1);
    }
}

// gF[u] += g * F[i], gF[i] += g * F[u], WITHOUT atomics: bit-reproducible under a seed.
// A row of gF collects the contributions of every pair of the batch that names it (a user drawn twice, an item that is one pair's
// positive and another's negative).  Every (pair, side) is looked at by one wave: it scans the batch's ids of that side
// in ascending pair order; the FIRST pair that names a row owns it -- its wave adds the contributions of all the pairs naming the
// row, in ascending pair order, in registers, and performs the single read-modify-write of the row; the later pairs' waves find an
// earlier match and leave.  O(B^2 / 64) compares per launch (B = 2048: 131 k wave-steps, ~2 us), no sort, no order that depends on
// timing.  (Rounds 1-4 added the products with float atomics: two runs of a train step differed in the last bit.)
constexpr int SCORE_BWD_DET_MAX_B = 4096;        // (64 scan rounds: a lane keeps its matches as one 64-bit mask per row) beyond it score_bwd_atomic_kernel
constexpr int SIDES_PER_WAVE = 8;                // (pair, side) items per wave, all of ONE side: they share the scan of that side's ids
template <typename T, int D>
__global__ __launch_bounds__(256) void score_bwd_kernel(const T* __restrict__ F, const int64_t* __restrict__ u,
                                                        const int64_t* __restrict__ it, const float* __restrict__ g,
                                                        float* __restrict__ gF, int* __restrict__ bad, int B, int U, int I) {
    const int lane = threadIdx.x & 63;
    const int wpl = (B + SIDES_PER_WAVE - 1) / SIDES_PER_WAVE;        // waves per list
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= 2 * wpl) return;
    const int side = w >= wpl;
    const int b0 = (side ? w - wpl : w) * SIDES_PER_WAVE;
    const int64_t* mine = side ? it : u;                               // the ids this side's rows are named by
    const int64_t* other = side ? u : it;
    const int64_t nmine = side ? I : U, nother = side ? U : I;
    constexpr int CPL = D / 64;                  // columns per lane
    // One scan of the list (straight from the batch's id array: 16 KB that every wave of the launch reads, L2 / L1 resident) for
    // the wave's rows: per row the number of pairs naming it and the first of them.  (Almost always one pair, itself; the
    // ordered walk below runs only for a row named more than once.)
    // Ids compare as 32-bit values: an id outside the table becomes a negative sentinel that equals no row of this wave (such a pair
    // is skipped anyway) -- 64-bit compares and shifts issue at a fraction of the 32-bit rate, and this scan is pure vector work.
    int id[SIDES_PER_WAVE];
    unsigned long long mk[SIDES_PER_WAVE];       // bit i: pair lane + 64 i names row j (B <= 4096: 64 rounds)
#pragma unroll
    for (int j = 0; j < SIDES_PER_WAVE; ++j) {
        const int64_t v = b0 + j < B ? mine[b0 + j] : -1, y = b0 + j < B ? other[b0 + j] : -1;
        id[j] = (v >= 0 && v < nmine && y >= 0 && y < nother) ? (int)v : -1;
        mk[j] = 0ull;
    }
    const int rounds = (B + 63) >> 6;
    for (int i0 = 0; i0 < rounds; i0 += 16) {     // sixteen rounds' ids requested together: two round trips for B = 2048, not eight
        int v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = lane + 64 * (i0 + r);
            const int64_t x = k < B ? mine[k] : -1, y = k < B ? other[k] : -1;
            v[r] = (x >= 0 && x < nmine && y >= 0 && y < nother) ? (int)x : -2;      // (a pair out of range on either side names no row)
        }
        unsigned int blk[SIDES_PER_WAVE];
#pragma unroll
        for (int j = 0; j < SIDES_PER_WAVE; ++j) blk[j] = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < SIDES_PER_WAVE; ++j) blk[j] |= v[r] == id[j] ? (1u << r) : 0u;
#pragma unroll
        for (int j = 0; j < SIDES_PER_WAVE; ++j) mk[j] |= (unsigned long long)blk[j] << i0;
    }
    // owner: no pair in front of b names the row (one ballot); dup: some pair behind b names it too (one ballot)
    bool is_first[SIDES_PER_WAVE], has_more[SIDES_PER_WAVE];
#pragma unroll
    for (int j = 0; j < SIDES_PER_WAVE; ++j) {
        const int b = b0 + j;
        const int lim = min(64, max(0, (b - lane + 63) >> 6));            // rounds i with lane + 64 i < b
        const unsigned long long before = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
        const int own_i = b >> 6;
        const unsigned long long self = (lane == (b & 63)) ? (1ull << own_i) : 0ull;
        is_first[j] = __ballot((mk[j] & before) != 0ull) == 0ull;
        has_more[j] = __ballot((mk[j] & ~before & ~self) != 0ull) != 0ull;
    }
    // The rows this wave owns, in PHASES over its eight items -- every phase's loads are independent of each other (one round trip
    // per phase, not per item: done item by item the launch took 24 us, each item a chain of three dependent trips).
    int64_t ob[SIDES_PER_WAVE];
    float gk[SIDES_PER_WAVE];
    bool owner[SIDES_PER_WAVE];
#pragma unroll
    for (int j = 0; j < SIDES_PER_WAVE; ++j) {
        const int b = b0 + j < B ? b0 + j : B - 1;
        ob[j] = other[b];
        gk[j] = g[b];
    }
    auto other_row = [&](int64_t o) { return (size_t)(side ? o : o + U); };
    auto f_at = [&](size_t row, int col) {
        if constexpr (sizeof(T) == 4) return reinterpret_cast<const float*>(F)[row * D + col];
        else return bf16_to_f32(reinterpret_cast<const uint16_t*>(F)[row * D + col]);
    };
    float acc[SIDES_PER_WAVE][CPL];
#pragma unroll
    for (int j = 0; j < SIDES_PER_WAVE; ++j) {
        const int b = b0 + j;
        const bool in = b < B && id[j] >= 0 && ob[j] >= 0 && ob[j] < nother;
        if (b < B && !in && side == 0 && lane == 0 && bad) atomicAdd(bad, 1);      // out-of-range pair: no write anywhere, counted once
        owner[j] = in && is_first[j];            // (else an earlier pair names this row: its wave does the sum)
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[j][c] = owner[j] ? gk[j] * f_at(other_row(ob[j]), lane + 64 * c) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < SIDES_PER_WAVE; ++j) {
        if (!owner[j] || !has_more[j]) continue; // (uniform) a row named more than once: the later pairs, ascending, from the scan's masks
        const int b = b0 + j;
        for (int i = b >> 6; i < rounds; ++i) {
            unsigned long long m = __ballot((mk[j] >> i) & 1ull);
            while (m) {
                const int kk = 64 * i + __builtin_ctzll(m);
                m &= m - 1;
                if (kk == b) continue;
                const int64_t o = other[kk];
                if (o < 0 || o >= nother) continue;                     // (that pair is out of range: it contributes nothing)
                const float gq = g[kk];
#pragma unroll
                for (int c = 0; c < CPL; ++c) acc[j][c] = fmaf(gq, f_at(other_row(o), lane + 64 * c), acc[j][c]);
            }
        }
    }
    float old[SIDES_PER_WAVE][CPL];
#pragma unroll
    for (int j = 0; j < SIDES_PER_WAVE; ++j) {
        const size_t row = owner[j] ? (size_t)(side ? id[j] + U : id[j]) : 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) old[j][c] = owner[j] ? gF[row * D + lane + 64 * c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < SIDES_PER_WAVE; ++j) {
        if (!owner[j]) continue;
        const size_t row = (size_t)(side ? id[j] + U : id[j]);
#pragma unroll
        for (int c = 0; c < CPL; ++c) gF[row * D + lane + 64 * c] = old[j][c] + acc[j][c];
    }
}

// The same with float atomics (the order of the additions is not fixed): batches beyond SCORE_BWD_DET_MAX_B pairs only.
template <typename T, int D>
__global__ __launch_bounds__(256) void score_bwd_atomic_kernel(const T* __restrict__ F, const int64_t* __restrict__ u,
                                                               const int64_t* __restrict__ it, const float* __restrict__ g,
                                                               float* __restrict__ gF, int* __restrict__ bad, int B, int U, int I) {
    constexpr int PPW = 4;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int b = wave * PPW + k;
        if (b >= B) return;
        const int64_t ub = u[b], ib = it[b];
        if (ub < 0 || ub >= U || ib < 0 || ib >= I) {       // out-of-range pair: no write anywhere, counted
            if (lane == 0 && bad) atomicAdd(bad, 1);
            continue;
        }
        const size_t ur = (size_t)ub, ir = (size_t)(ib + U);
        const float gb = g[b];
#pragma unroll
        for (int c = lane; c < D; c += 64) {
            float x, y;
            if constexpr (sizeof(T) == 4) {
                x = reinterpret_cast<const float*>(F)[ur * D + c];
                y = reinterpret_cast<const float*>(F)[ir * D + c];
            } else {
                x = bf16_to_f32(reinterpret_cast<const uint16_t*>(F)[ur * D + c]);
                y = bf16_to_f32(reinterpret_cast<const uint16_t*>(F)[ir * D + c]);
            }
            atomicAdd(gF + ur * D + c, gb * y);
            atomicAdd(gF + ir * D + c, gb * x);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Node-partitioned scoring (ngacf_amd/dist.py _RowExchange): the 2B rows a batch reads of a table whose rows are spread over the
// ranks.  Every rank contributes the rows it owns (zeros elsewhere), ONE all-reduce of [2B, d] sums them (one non-zero term per
// row: exact).  These two kernels replace ~25 index / compare / gather launches of 5 us each per forward and the index_add_ of
// the backward.
//   shard_rows:  rows[b] = fp32 copy of the local row of pair side b (b < B: user u[b]; else item it[b - B]) when this rank owns
//                it, zeros otherwise; pos[b] = its local row (users first, items behind the cu user rows) or -1.
//                perm_u / perm_i: data-set id -> internal id (the degree-dealt numbering), or NULL (identity).
//   rows_scatter: d[pos[b]] = sum of g[b'] over all b' with pos[b'] == pos[b], in ascending b' (the first such b does the sum
//                and the one store: no atomics, bit-reproducible); rows nobody names are not written (the caller zero-fills).
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void shard_rows_kernel(const T* __restrict__ Tu, const T* __restrict__ Ti, const int64_t* __restrict__ u,
                                                         const int64_t* __restrict__ it, const int64_t* __restrict__ perm_u,
                                                         const int64_t* __restrict__ perm_i, float* __restrict__ rows, int* __restrict__ pos,
                                                         int* __restrict__ bad, int B, int U, int I, int u_lo, int cu, int i_lo, int ci) {
    constexpr int LPR = D / 4;                                    // lanes per row: a float4 of the output each
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = (int)(t / LPR), q = (int)(t % LPR);
    if (b >= 2 * B) return;
    const bool user = b < B;
    const int64_t id = user ? u[b] : it[b - B];
    const bool in = id >= 0 && id < (user ? U : I);
    int64_t internal = -1;
    if (in) internal = user ? (perm_u ? perm_u[id] : id) : (perm_i ? perm_i[id] : id);
    const int lo = user ? u_lo : i_lo, n = user ? cu : ci;
    const bool mine = in && internal >= lo && internal < lo + n;
    const int local = mine ? (int)(internal - lo) : 0;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mine) {
        const T* src = (user ? Tu : Ti) + (size_t)local * D + 4 * q;
        if constexpr (sizeof(T) == 4) v = *reinterpret_cast<const float4*>(src);
        else {
            const uint2 w = *reinterpret_cast<const uint2*>(src);
            v = make_float4(bf16_to_f32(w.x & 0xffffu), bf16_to_f32(w.x >> 16), bf16_to_f32(w.y & 0xffffu), bf16_to_f32(w.y >> 16));
        }
    }
    *reinterpret_cast<float4*>(rows + (size_t)b * D + 4 * q) = v;
    if (q == 0) {
        pos[b] = mine ? (user ? local : cu + local) : -1;
        if (!in && bad) atomicAdd(bad, 1);
    }
}

constexpr int ROWS_SCATTER_MAX_N = 16384;        // ids in 64 KB of LDS
// SCORED: g is not given; the n = 2B rows are an exchanged batch (rows[b] = the user row of pair b, rows[B + b] = its item row, score
// of pair b = rows[b] . rows[B + b]) and gs[b] the gradient of that score: g[k] = gs[k mod B] * rows[(k + B) mod 2B] is formed on the
// fly -- the products and the order of the sums of ngacf_score_bwd followed by the plain scatter, one launch instead of two.
template <typename T, int D, bool SCORED = false>
__global__ __launch_bounds__(256) void rows_scatter_kernel(const int* __restrict__ pos, const float* __restrict__ g, T* __restrict__ du,
                                                           T* __restrict__ di, int n, int cu, const float* __restrict__ gs = nullptr) {
#pragma clang fp contract(off)                    // (SCORED: a product is rounded on its own, as the two-launch form stores it -- never fused into the sums)
    extern __shared__ int ids[];
    const int B = n >> 1;
    auto gval = [&](int k, int col) -> float {
        if constexpr (SCORED) return gs[k < B ? k : k - B] * g[(size_t)(k < B ? k + B : k - B) * D + col];
        else return g[(size_t)k * D + col];
    };
    for (int k = threadIdx.x; k < n; k += 256) ids[k] = pos[k];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const int row = ids[b];
    if (row < 0) return;
    int cnt = 0, first = 0x7fffffff;
    for (int k = lane; k < n; k += 64) {
        const bool m = ids[k] == row;
        cnt += m ? 1 : 0;
        first = m ? min(first, k) : first;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        cnt += __shfl_xor(cnt, m, 64);
        first = min(first, __shfl_xor(first, m, 64));
    }
    if (first != b) return;                      // an earlier entry names this row: its wave does the sum
    constexpr int CPL = D / 64;
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = gval(b, lane + 64 * c);
    if (cnt > 1) {
        for (int k0 = (b + 1) & ~63; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            unsigned long long m = __ballot(k < n && k > b && ids[k] == row);
            while (m) {
                const int kk = k0 + __builtin_ctzll(m);
                m &= m - 1;
#pragma unroll
                for (int c = 0; c < CPL; ++c) acc[c] += gval(kk, lane + 64 * c);
            }
        }
    }
    T* dst = row < cu ? du + (size_t)row * D : di + (size_t)(row - cu) * D;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        if constexpr (sizeof(T) == 4) reinterpret_cast<float*>(dst)[lane + 64 * c] = acc[c];
        else reinterpret_cast<uint16_t*>(dst)[lane + 64 * c] = (uint16_t)f32_to_bf16(acc[c]);
    }
}

// bit u[b] and bit U + it[b] of `needed` set for the pairs of a batch, and the same rows as a list (rows[b] = u[b], rows[B + b] =
// U + it[b]; -1 for an id out of range: the scoring kernels report those).  u == NULL: items only (rows[b] = U + it[b]).
// gzero != NULL: rows u[b] and U + it[b] of that float [N][d] table are zeroed (where ngacf_score_bwd will add the batch's
// gradient).  d / 4 threads per pair.
__global__ __launch_bounds__(256) void mark_rows_kernel(const int64_t* __restrict__ u, const int64_t* __restrict__ it,
                                                        uint32_t* __restrict__ needed, int* __restrict__ rows, float* __restrict__ gzero,
                                                        int B, int U, int I, int d) {
    const int tpp = d >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int b = t / tpp, piece = t - b * tpp;
    if (b >= B) return;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u) {
        const int64_t ub = u[b];
        const bool ok = ub >= 0 && ub < U;
        if (piece == 0) {
            if (ok) atomicOr(needed + (ub >> 5), 1u << (ub & 31));
            rows[b] = ok ? (int)ub : -1;
        }
        if (ok && gzero) reinterpret_cast<float4*>(gzero + (size_t)ub * d)[piece] = z;
        rows += B;
    }
    const int64_t ib = it[b];
    const bool ok = ib >= 0 && ib < I;
    if (piece == 0) {
        if (ok) atomicOr(needed + ((U + ib) >> 5), 1u << ((U + ib) & 31));
        rows[b] = ok ? U + (int)ib : -1;
    }
    if (ok && gzero) reinterpret_cast<float4*>(gzero + (size_t)(U + ib) * d)[piece] = z;
}

__device__ __forceinline__ float log_sigmoid(float x) {       // log(1/(1+exp(-x))), stable for both signs
    return fminf(x, 0.f) - log1pf(__expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void bpr_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                                                      float* __restrict__ loss, int B) {
    float s = 0.f;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) s -= log_sigmoid(pos[b] - neg[b]);
    s = group_sum<1, WAVE>(s);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (ws[0] + ws[1] + ws[2] + ws[3]) / (float)B;
        if (gridDim.x == 1) *loss = v;                      // usual batch sizes: one workgroup, deterministic, no pre-zero
        else atomicAdd(loss, v);
    }
}

// loss AND its gradients (for d loss = 1) in one pass: gpos[b] = -sigmoid(neg - pos) / B = -gneg[b]
__global__ __launch_bounds__(256) void bpr_fwd_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg, float* __restrict__ loss,
                                                          float* __restrict__ gpos, float* __restrict__ gneg, int B) {
    float s = 0.f;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        const float x = pos[b] - neg[b];
        s -= log_sigmoid(x);
        const float g = -(1.f / (1.f + __expf(x))) / (float)B;             // (bpr_bwd_kernel's, for d loss = 1)
        gpos[b] = g;
        gneg[b] = -g;
    }
    s = group_sum<1, WAVE>(s);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (ws[0] + ws[1] + ws[2] + ws[3]) / (float)B;      // (bpr_fwd_kernel's expression: the same bits)
        if (gridDim.x == 1) *loss = v;
        else atomicAdd(loss, v);
    }
}

__global__ __launch_bounds__(256) void bpr_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                                                      const float* __restrict__ gloss, float* __restrict__ gpos,
                                                      float* __restrict__ gneg, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float x = pos[b] - neg[b];
    const float sg = 1.f / (1.f + __expf(x));                  // sigmoid(-x)
    const float g = -gloss[0] * sg / (float)B;
    gpos[b] = g;
    gneg[b] = -g;
}

__global__ void dropout_bits_kernel(int kind, uint64_t seed, const int* a, const int* b, const int* c, uint32_t* bits, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (kind == 0) bits[k] = feat_bits(salt_of(seed, 0u), a[k], b[k]);
    else bits[k] = edge_bits(salt_of(seed, 1u + c[k]), a[k], b[k]);
}

}  // namespace ngacf

using namespace ngacf;

extern "C" const char* ngacf_version(void) { return "ngacf-hip 0.1 (gfx950)"; }

extern "C" int ngacf_score_fwd(const void* F, int dtype, const int64_t* u, const int64_t* i,
                               float* score, int32_t* bad_index, int B, int U, int I, int d, void* stream) {
    if (B < 0 || U <= 0 || I <= 0) return NGACF_E_ARG;
    if (B == 0) return 0;
    if (!F || !u || !i || !score) return NGACF_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NGACF_CASE(T, D)                                                                                   \
    {                                                                                                      \
        constexpr int PPB = 4 * WAVE / (D / Slice<T>::VEC);                                                \
        hipLaunchKernelGGL((score_fwd_kernel<T, D>), dim3((B + PPB - 1) / PPB), dim3(256), 0, st,          \
                           static_cast<const T*>(F), u, i, score, bad_index, B, U, I);                                   \
        NGACF_CHECK_LAUNCH();                                                                              \
        return 0;                                                                                          \
    }
    if (dtype == NGACF_F32 && d == 64) NGACF_CASE(float, 64)
    if (dtype == NGACF_F32 && d == 128) NGACF_CASE(float, 128)
    if (dtype == NGACF_BF16 && d == 64) NGACF_CASE(bf16_t, 64)
    if (dtype == NGACF_BF16 && d == 128) NGACF_CASE(bf16_t, 128)
#undef NGACF_CASE
    return NGACF_E_SHAPE;
}

extern "C" int ngacf_score_bwd(const void* F, int dtype, const int64_t* u, const int64_t* i,
                               const float* gscore, float* gF, int32_t* bad_index, int B, int U, int I, int d, void* stream) {
    if (B < 0 || U <= 0 || I <= 0) return NGACF_E_ARG;
    if (B == 0) return 0;
    if (!F || !u || !i || !gscore || !gF) return NGACF_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NGACF_CASE(T, D)                                                                                   \
    {                                                                                                      \
        if (B <= SCORE_BWD_DET_MAX_B) {                                                                    \
            const int waves = 2 * ((B + SIDES_PER_WAVE - 1) / SIDES_PER_WAVE);                             \
            hipLaunchKernelGGL((score_bwd_kernel<T, D>), dim3((waves + 3) / 4), dim3(256), 0, st,          \
                               static_cast<const T*>(F), u, i, gscore, gF, bad_index, B, U, I);            \
        } else {                                                                                           \
            constexpr int PPB = 16;                                                                        \
            hipLaunchKernelGGL((score_bwd_atomic_kernel<T, D>), dim3((B + PPB - 1) / PPB), dim3(256), 0, st, \
                               static_cast<const T*>(F), u, i, gscore, gF, bad_index, B, U, I);            \
        }                                                                                                  \
        NGACF_CHECK_LAUNCH();                                                                              \
        return 0;                                                                                          \
    }
    if (dtype == NGACF_F32 && d == 64) NGACF_CASE(float, 64)
    if (dtype == NGACF_F32 && d == 128) NGACF_CASE(float, 128)
    if (dtype == NGACF_BF16 && d == 64) NGACF_CASE(bf16_t, 64)
    if (dtype == NGACF_BF16 && d == 128) NGACF_CASE(bf16_t, 128)
#undef NGACF_CASE
    return NGACF_E_SHAPE;
}

extern "C" int ngacf_mark_rows(const int64_t* u, const int64_t* i, uint32_t* needed, int32_t* rows, float* gzero, int B, int U, int I,
                               int d, int clear, void* stream) {
    if (B < 0 || U <= 0 || I <= 0 || !needed || !rows || d < 4 || d % 4 != 0) return NGACF_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (clear) {
        // (whole 64-byte units: the runtime splits a fill with a ragged tail into two kernels -- the table is allocated padded)
        const hipError_t e = hipMemsetAsync(needed, 0, 64 * ((4 * (((size_t)U + (size_t)I + 31) / 32) + 63) / 64), st);
        if (e != hipSuccess) return static_cast<int>(e);
    }
    if (B == 0) return 0;
    if (!i) return NGACF_E_ARG;
    const size_t threads = (size_t)B * (d / 4);
    hipLaunchKernelGGL(mark_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, u, i, needed, rows, gzero, B, U, I, d);
    NGACF_CHECK_LAUNCH();
    return 0;
}

extern "C" int ngacf_bpr_fwd(const float* pos, const float* neg, float* loss, int B, void* stream) {
    if (B <= 0 || !pos || !neg || !loss) return NGACF_E_ARG;
    const int nb = B <= (1 << 16) ? 1 : 64;
    if (nb > 1) {
        const hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return static_cast<int>(e);
    }
    hipLaunchKernelGGL(bpr_fwd_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), pos, neg, loss, B);
    NGACF_CHECK_LAUNCH();
    return 0;
}

extern "C" int ngacf_bpr_fwd_bwd(const float* pos, const float* neg, float* loss, float* gpos, float* gneg, int B, void* stream) {
    if (B <= 0 || !pos || !neg || !loss || !gpos || !gneg) return NGACF_E_ARG;
    const int nb = B <= (1 << 16) ? 1 : 64;
    if (nb > 1) {
        const hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return static_cast<int>(e);
    }
    hipLaunchKernelGGL(bpr_fwd_bwd_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), pos, neg, loss, gpos, gneg, B);
    NGACF_CHECK_LAUNCH();
    return 0;
}

extern "C" int ngacf_bpr_bwd(const float* pos, const float* neg, const float* gloss,
                             float* gpos, float* gneg, int B, void* stream) {
    if (B <= 0 || !pos || !neg || !gloss || !gpos || !gneg) return NGACF_E_ARG;
    hipLaunchKernelGGL(bpr_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       pos, neg, gloss, gpos, gneg, B);
    NGACF_CHECK_LAUNCH();
    return 0;
}

extern "C" int ngacf_dropout_bits(int kind, uint64_t seed, const int32_t* a, const int32_t* b, const int32_t* c,
                                  uint32_t* bits, int n, void* stream) {
    if (n < 0 || (kind != 0 && kind != 1)) return NGACF_E_ARG;
    if (n == 0) return 0;
    if (!a || !b || !bits || (kind == 1 && !c)) return NGACF_E_ARG;
    hipLaunchKernelGGL(dropout_bits_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       kind, seed, a, b, c, bits, n);
    NGACF_CHECK_LAUNCH();
    return 0;
}

extern "C" int ngacf_shard_rows(const void* Tu, const void* Ti, int dtype, const int64_t* u, const int64_t* i, const int64_t* perm_u,
                                const int64_t* perm_i, float* rows, int32_t* pos, int32_t* bad_index, int B, int U, int I,
                                int u_lo, int cu, int i_lo, int ci, int d, void* stream) {
    if (B < 0 || U <= 0 || I <= 0 || cu < 0 || ci < 0 || u_lo < 0 || i_lo < 0) return NGACF_E_ARG;
    if (B == 0) return 0;
    if (!u || !i || !rows || !pos || (cu > 0 && !Tu) || (ci > 0 && !Ti)) return NGACF_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NGACF_CASE(T, D)                                                                                                       \
    {                                                                                                                          \
        const size_t threads = (size_t)2 * B * (D / 4);                                                                        \
        hipLaunchKernelGGL((shard_rows_kernel<T, D>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,               \
                           static_cast<const T*>(Tu), static_cast<const T*>(Ti), u, i, perm_u, perm_i, rows, pos, bad_index,   \
                           B, U, I, u_lo, cu, i_lo, ci);                                                                       \
        NGACF_CHECK_LAUNCH();                                                                                                  \
        return 0;                                                                                                              \
    }
    if (dtype == NGACF_F32 && d == 64) NGACF_CASE(float, 64)
    if (dtype == NGACF_F32 && d == 128) NGACF_CASE(float, 128)
    if (dtype == NGACF_BF16 && d == 64) NGACF_CASE(bf16_t, 64)
    if (dtype == NGACF_BF16 && d == 128) NGACF_CASE(bf16_t, 128)
#undef NGACF_CASE
    return NGACF_E_SHAPE;
}

extern "C" int ngacf_rows_scatter_scored(const int32_t* pos, const float* rows, const float* gscore, void* du, void* di, int dtype,
                                         int n, int cu, int d, void* stream) {
    if (n < 0 || cu < 0 || (n & 1)) return NGACF_E_ARG;
    if (n == 0) return 0;
    if (!pos || !rows || !gscore || n > ROWS_SCATTER_MAX_N) return NGACF_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NGACF_CASE(T, D)                                                                                                       \
    {                                                                                                                          \
        hipLaunchKernelGGL((rows_scatter_kernel<T, D, true>), dim3((n + 3) / 4), dim3(256), (size_t)n * sizeof(int), st,       \
                           pos, rows, static_cast<T*>(du), static_cast<T*>(di), n, cu, gscore);                                \
        NGACF_CHECK_LAUNCH();                                                                                                  \
        return 0;                                                                                                              \
    }
    if (dtype == NGACF_F32 && d == 64) NGACF_CASE(float, 64)
    if (dtype == NGACF_F32 && d == 128) NGACF_CASE(float, 128)
    if (dtype == NGACF_BF16 && d == 64) NGACF_CASE(bf16_t, 64)
    if (dtype == NGACF_BF16 && d == 128) NGACF_CASE(bf16_t, 128)
#undef NGACF_CASE
    return NGACF_E_SHAPE;
}

extern "C" int ngacf_rows_scatter(const int32_t* pos, const float* g, void* du, void* di, int dtype, int n, int cu, int d, void* stream) {
    if (n < 0 || cu < 0) return NGACF_E_ARG;
    if (n == 0) return 0;
    if (!pos || !g || n > ROWS_SCATTER_MAX_N) return NGACF_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NGACF_CASE(T, D)                                                                                                       \
    {                                                                                                                          \
        hipLaunchKernelGGL((rows_scatter_kernel<T, D, false>), dim3((n + 3) / 4), dim3(256), (size_t)n * sizeof(int), st,      \
                           pos, g, static_cast<T*>(du), static_cast<T*>(di), n, cu, nullptr);                                  \
        NGACF_CHECK_LAUNCH();                                                                                                  \
        return 0;                                                                                                              \
    }
    if (dtype == NGACF_F32 && d == 64) NGACF_CASE(float, 64)
    if (dtype == NGACF_F32 && d == 128) NGACF_CASE(float, 128)
    if (dtype == NGACF_BF16 && d == 64) NGACF_CASE(bf16_t, 64)
    if (dtype == NGACF_BF16 && d == 128) NGACF_CASE(bf16_t, 128)
#undef NGACF_CASE
    return NGACF_E_SHAPE;
}
