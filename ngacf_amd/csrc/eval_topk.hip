// All-item ranking of the evaluation: for every evaluated user the K best-scoring
// candidate items, candidates = item pool minus the user's train positives, ties broken by the lower item id.
// Replaces train_eval_Gowalla.py:274-354 (score blocks shipped to the host) + :370-385 (heapq.nlargest per user
// in a process pool).  The [users x items] score matrix is a dense contraction -> matrix cores; it is never
// written to memory: a wave keeps the running top-K of its 32 users and only scores that beat a user's current
// K-th best leave the accumulator registers.
//
//   workgroup = 32 users x all items, NW waves (1 when the user groups alone fill the chip, else 4); wave w takes the 32-item
//               tiles t = w (mod NW)
//   tile      = S[32 users][32 items] by v_mfma_f32_32x32x2_f32 (fp32 table: exact products, fp32 accumulate)
//               or v_mfma_f32_32x32x16_bf16 (bf16 table: products of bf16 values are exact in fp32 too)
//   select    = per accumulator register (one user per 32-lane half): score > tau_user; the rare survivors
//               append THEMSELVES to the user's LDS buffer (K live entries + slack; slot by LDS atomic, no
//               wave-serial work); when a buffer overflows the wave compacts it: train positives of the user out (first 64 cached in LDS), every entry
//               ranked against the others in registers, the K best rewritten in order, tau_user raised
//   finish    = the NW waves' lists of a user are merged by rank ((score desc, item asc) is a total order)
//
// Items stream in ascending id order inside a wave, so "score > tau" (strict) is exactly the stable descending
// sort's tie rule; the cross-wave merge compares ids explicitly.
#include <stdlib.h>
#include <type_traits>
#include "ngacf_common.hpp"

namespace ngacf {

typedef __attribute__((ext_vector_type(16))) float tk_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 tk_bf16x8;

constexpr int TK_MAXK = 32;          // largest K
constexpr int TK_CAP = 64;           // buffer entries per (wave, user): K live + slack between compactions (one per lane of the compacting wave)
constexpr int TK_CACHE = 64;          // train items of a user kept in LDS
constexpr float TK_NEG = -3.402823466e38f;

// largest float below x (finite x): "score > below(t)" is "score >= t"
__device__ __forceinline__ float tk_below(float x) {
    const int b = __float_as_int(x);
    return x > 0.f ? __int_as_float(b - 1) : (x < 0.f ? __int_as_float(b + 1) : -1.401298464e-45f);
}

struct TopkArgs {
    const void* F;                 // [U+I][d] propagated embeddings
    const int64_t* users;          // [B] user ids
    const uint8_t* pool;           // [I] 1 = item is a candidate at all
    const int64_t* train_ptr;      // [U+1] CSR of the train positives
    const int32_t* train_items;    // sorted inside a row
    int32_t* top_items;            // [B][K], -1 when a user has fewer than K candidates
    float* top_scores;             // [B][K]
    int* bad;                      // optional: number of users[] entries outside [0, U) (their rows come back all -1)
    int B, U, I, K;
};

// users[ub] clamped into the table: an out-of-range id must never become an address (it is reported through a.bad)
__device__ __forceinline__ int64_t tk_user(const TopkArgs& a, int ub) {
    const int64_t u = a.users[ub];
    return u < 0 ? 0 : (u >= a.U ? (int64_t)a.U - 1 : u);
}

// NW = waves that share one group of 32 users, each taking every NW-th item tile and keeping its own lists (merged at the end).
// More waves per group = more parallelism per user, but every list pays its own warm-up (the first ~K ln(n / K) items it sees
// all pass): 4 waves insert ~860 scores per user and compact their lists ~24 times, one wave ~260 and ~8.
template <typename T, int D, int NW>
__global__ __launch_bounds__(64 * NW) void eval_topk_kernel(const TopkArgs a) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int NA = BF ? D / 16 : D / 2;        // operand registers (bf16x8 fragments / floats) per row
    // plain (not volatile) LDS arrays: volatile accesses are emitted as system-coherent flat_load/flat_store, an order
    // of magnitude slower than ds_read/ds_write.  In-wave ordering needs nothing more: LDS executes a wave's
    // operations in order and the compiler keeps may-aliasing accesses to one array in program order.
    __shared__ float l_score[NW][32][TK_CAP + 1];         // (+1: the 32 users' rows start on different banks)
    __shared__ int l_item[NW][32][TK_CAP + 1];
    __shared__ int l_cnt[NW][32];
    __shared__ int c_item[32][TK_CACHE];
    __shared__ int c_deg[32];
    // best K-th-best any of the 4 waves holds for a user, stored one ulp lower: another wave's candidate with exactly
    // that score may carry a lower item id and must still get through (ties go to the lower id)
    __shared__ float g_tau[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int u0 = blockIdx.x * 32;
    const int K = a.K;

    // ---- per-user state: empty lists, the first TK_CACHE train items ---------------------------------------
    if (lane < 32) l_cnt[wave][lane] = 0;
    if (tid < 32) g_tau[tid] = TK_NEG;
    for (int i = wave; i < 32; i += NW) {
        const int ub = u0 + i;
        int deg = 0;
        if (ub < a.B) {
            const int64_t u = tk_user(a, ub);
            const int64_t p0 = a.train_ptr[u];
            deg = (int)(a.train_ptr[u + 1] - p0);
            if (lane < deg && lane < TK_CACHE) c_item[i][lane] = a.train_items[p0 + lane];
        }
        if (lane == 0) c_deg[i] = deg;
    }

    // ---- A operand: this lane's user row (users past the batch end take the last one and never select) ------
    const int my_u = u0 + col < a.B ? u0 + col : a.B - 1;
    const T* urow = static_cast<const T*>(a.F) + (size_t)tk_user(a, my_u) * D;
    float af[BF ? 1 : NA];
    tk_bf16x8 ab[BF ? NA : 1];
    if constexpr (BF) {
#pragma unroll
        for (int s = 0; s < NA; ++s) ab[s] = *reinterpret_cast<const tk_bf16x8*>(reinterpret_cast<const uint16_t*>(urow) + 16 * s + 8 * kh);
    } else {
#pragma unroll
        for (int s = 0; s < NA; s += 4) {
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(urow) + kh * (D / 2) + s);
            af[s] = v.x; af[s + 1] = v.y; af[s + 2] = v.z; af[s + 3] = v.w;
        }
    }
    // threshold of the user behind accumulator register e in this lane's half: user (e&3) + 8(e>>2) + 4 kh
    float tau[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) tau[e] = (u0 + (e & 3) + 8 * (e >> 2) + 4 * kh) < a.B ? TK_NEG : 3.402823466e38f;
    __syncthreads();

    const T* items = static_cast<const T*>(a.F) + (size_t)a.U * D;
    const int ntiles = (a.I + 31) / 32;

    // The wave compacts user ui's buffer of n entries: train positives out, the K best kept in rank order
    // ((score desc, item asc) is a total order, so ranks are distinct).  Returns the new entry count; `own` = the
    // K-th best when K entries remain.  All arguments are wave-uniform.
    auto compact = [&](int ui, int n, float& own) -> int {
        own = TK_NEG;
        if (n == 0) return 0;
        const bool live = lane < n;
        const float s_t = live ? l_score[wave][ui][lane] : TK_NEG;
        const int i_t = live ? l_item[wave][ui][lane] : 0x7fffffff;
        const int deg = c_deg[ui];
        const int nc = deg < TK_CACHE ? deg : TK_CACHE;
        bool member = false;
        for (int c = 0; c < nc; ++c) member |= i_t == c_item[ui][c];          // broadcast reads
        if (deg > TK_CACHE && live && !member && i_t > c_item[ui][TK_CACHE - 1]) {
            const int64_t p0 = a.train_ptr[tk_user(a, u0 + ui)];
            int lo = TK_CACHE, hi = deg;                                       // per-lane binary search over the uncached tail
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (a.train_items[p0 + mid] < i_t) lo = mid + 1; else hi = mid;
            }
            member = lo < deg && a.train_items[p0 + lo] == i_t;
        }
        const bool valid = live && !member;
        const unsigned long long vm = __ballot(valid);
        // rank = entries that beat this one.  Scores first, straight-line over all buffer slots (an entry that does not count
        // carries the lowest score and outranks nobody: no branch per slot, one v_readlane + compare + add each); equal scores
        // are rare and settled by item id in a second pass only when some lane saw one.  (The first form -- a loop with a
        // branch and two v_readlane per slot -- took 2.3 us per compaction, three quarters of the selection's time.)
        const float s_v = valid ? s_t : TK_NEG;
        int rank = 0;
        bool tie = false;
#pragma unroll
        for (int t = 0; t < TK_CAP; ++t) {
            const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s_v), t));
            rank += s2 > s_v ? 1 : 0;
            tie |= s2 == s_v && t != lane;
        }
        if (__ballot(tie && valid) != 0ull) {
            for (int t = 0; t < n; ++t) {
                if (!((vm >> t) & 1ull)) continue;
                const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s_t), t));
                const int i2 = __builtin_amdgcn_readlane(i_t, t);
                rank += (s2 == s_t && i2 < i_t) ? 1 : 0;
            }
        }
        const int nvalid = __popcll(vm);
        const int ncnt = nvalid < K ? nvalid : K;
        if (valid && rank < K) {
            l_score[wave][ui][rank] = s_t;
            l_item[wave][ui][rank] = i_t;
        }
        if (ncnt == K) {
            const unsigned long long last = __ballot(valid && rank == K - 1);
            own = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s_t), __builtin_ctzll(last)));
        }
        return ncnt;
    };

    // The item rows of the wave's NEXT tile are requested before the current tile is multiplied and sifted: the loop used to
    // issue its loads, wait a full L2 round trip, and only then compute -- with two waves per SIMD (the lists' LDS) nothing hid
    // that wait, and a tile cost ~5 us of wall time against ~0.3 us of matrix-core and vector work.
    struct TileOps {
        tk_bf16x8 bb[BF ? NA : 1];
        float bfv[BF ? 1 : NA];
        int j;
        bool cand;
    };
    auto fetch = [&](int tile, TileOps& t) {
        t.j = tile * 32 + col;
        const bool jv = t.j < a.I;
        const int jj = jv ? t.j : a.I - 1;
        const T* irow = items + (size_t)jj * D;
        t.cand = jv && a.pool[jj] != 0;
        if constexpr (BF) {
#pragma unroll
            for (int s = 0; s < NA; ++s) t.bb[s] = *reinterpret_cast<const tk_bf16x8*>(reinterpret_cast<const uint16_t*>(irow) + 16 * s + 8 * kh);
        } else {
#pragma unroll
            for (int s = 0; s < NA; s += 4) {
                const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(irow) + kh * (D / 2) + s);
                t.bfv[s] = v.x; t.bfv[s + 1] = v.y; t.bfv[s + 2] = v.z; t.bfv[s + 3] = v.w;
            }
        }
    };
    // Software pipeline, two tiles deep: while tile k is sifted (vector and LDS work), tile k+1 is on the matrix cores (its
    // accumulator is not read before the next turn) and tile k+2's rows are on their way from L2.
    auto multiply = [&](const TileOps& t) -> tk_f32x16 {
        tk_f32x16 r;
#pragma unroll
        for (int e = 0; e < 16; ++e) r[e] = 0.f;
        if constexpr (BF) {
#pragma unroll
            for (int s = 0; s < NA; ++s) r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[s], t.bb[s], r, 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < NA; ++s) r = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], t.bfv[s], r, 0, 0, 0);
        }
        return r;
    };
    TileOps ops1, ops2;                                       // operands of tile k+1 (loaded) and k+2 (in flight)
    ops1.j = 0; ops1.cand = false; ops2.j = 0; ops2.cand = false;
    if (wave < ntiles) fetch(wave, ops1);
    if (wave + NW < ntiles) fetch(wave + NW, ops2);
    tk_f32x16 acc_n = multiply(ops1);
    int j_n = ops1.j;
    bool cand_n = ops1.cand;
#pragma unroll 1
    for (int tile = wave; tile < ntiles; tile += NW) {
        const tk_f32x16 acc = acc_n;
        const int j = j_n;
        const bool cand = cand_n;
        ops1 = ops2;
        if (tile + 2 * NW < ntiles) fetch(tile + 2 * NW, ops2);
        if (tile + NW < ntiles) {
            acc_n = multiply(ops1);
            j_n = ops1.j;
            cand_n = ops1.cand;
        }

        // acc[e] in lane (col, kh) = score(user (e&3) + 8(e>>2) + 4kh, item tile*32 + col)
        if (NW > 1 && (((tile - wave) / NW) & 7) == 0) {   // every 8th tile of this wave: what the other waves found
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; e < 16; ++e) tau[e] = fmaxf(tau[e], g_tau[(e & 3) + 8 * (e >> 2) + 4 * kh]);
        }
        // Which of this lane's 16 scores beat their user's threshold?  One bit per register; a tile in which nobody's does (most
        // tiles, once the lists have filled) is dropped with ONE ballot.
        unsigned int todo = 0u;
#pragma unroll
        for (int e = 0; e < 16; ++e) todo |= (acc[e] > tau[e] ? 1u : 0u) << e;
        if (!cand) todo = 0u;
        // Survivors append themselves to their user's buffer: one LDS atomic for the slot.  The wave makes as many rounds as its
        // busiest lane has survivors (one or two), not sixteen: every lane takes its lowest pending register per round.
        unsigned int pend = 0u;                            // bit e: this lane's score of register e found the buffer full
        while (__ballot(todo != 0u) != 0ull) {
            if (todo != 0u) {
                const int e = __builtin_ctz(todo);
                todo &= todo - 1u;
                float v = acc[0];
#pragma unroll
                for (int k = 1; k < 16; ++k) v = e == k ? acc[k] : v;
                const int ui = (e & 3) + 8 * (e >> 2) + 4 * kh;
                const int pos = atomicAdd(&l_cnt[wave][ui], 1);
                if (pos < TK_CAP) {
                    l_score[wave][ui][pos] = v;
                    l_item[wave][ui][pos] = j;
                } else {
                    pend |= 1u << e;
                }
            }
        }
        while (__ballot(pend != 0u) != 0ull) {             // rare: some buffer overflowed -> compact it, raise tau, retry
            const int c = lane < 32 ? l_cnt[wave][lane] : 0;
            unsigned int over = (unsigned int)__ballot(lane < 32 && c >= TK_CAP);
            while (over) {
                const int ui = __builtin_ctz(over);
                over &= over - 1u;
                float own;
                const int n = compact(ui, TK_CAP, own);
                if (lane == 0) l_cnt[wave][ui] = n;
                if (n == K) {
                    if (lane == 0 && tk_below(own) > g_tau[ui]) g_tau[ui] = tk_below(own);   // racy on purpose (see g_tau)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if ((e & 3) + 8 * (e >> 2) + 4 * kh == ui) tau[e] = fmaxf(tau[e], own);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (!(pend & (1u << e))) continue;
                const float v = acc[e];                    // (cand holds: the bit was set)
                pend &= ~(1u << e);
                if (v > tau[e]) {
                    const int ui = (e & 3) + 8 * (e >> 2) + 4 * kh;
                    const int pos = atomicAdd(&l_cnt[wave][ui], 1);
                    if (pos < TK_CAP) {
                        l_score[wave][ui][pos] = v;
                        l_item[wave][ui][pos] = j;
                    } else {
                        pend |= 1u << e;
                    }
                }
            }
        }
    }
    // leave every list compacted: valid entries only, best first, at most K
    for (int ui = 0; ui < 32; ++ui) {
        float own;
        const int have = l_cnt[wave][ui];
        const int cnt = compact(ui, have < TK_CAP ? have : TK_CAP, own);
        if (lane == 0) l_cnt[wave][ui] = cnt;
    }
    __syncthreads();

    // ---- merge the 4 waves' lists of each user by rank ----------------------------------------------------------
    for (int ui = wave; ui < 32; ui += NW) {
        const int ub = u0 + ui;
        if (ub >= a.B) continue;
        const int64_t uraw = a.users[ub];
        if (uraw < 0 || uraw >= a.U) {                       // not a user: no ranking, flagged
            if (lane < K) {
                a.top_items[(size_t)ub * K + lane] = -1;
                a.top_scores[(size_t)ub * K + lane] = TK_NEG;
            }
            if (lane == 0 && a.bad) atomicAdd(a.bad, 1);
            continue;
        }
        int total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) total += l_cnt[w][ui];
#pragma unroll 1
        for (int c = lane; c < NW * TK_MAXK; c += 64) {
            const int w = c / TK_MAXK, p = c % TK_MAXK;     // compacted lists hold at most K <= TK_MAXK entries
            if (p >= l_cnt[w][ui]) continue;
            const float s = l_score[w][ui][p];
            const int it = l_item[w][ui][p];
            int rank = 0;
            for (int w2 = 0; w2 < NW; ++w2) {
                const int n2 = l_cnt[w2][ui];
                for (int p2 = 0; p2 < n2; ++p2) {
                    const float s2 = l_score[w2][ui][p2];
                    const int it2 = l_item[w2][ui][p2];
                    rank += (s2 > s || (s2 == s && it2 < it)) ? 1 : 0;
                }
            }
            if (rank < K) {
                a.top_items[(size_t)ub * K + rank] = it;
                a.top_scores[(size_t)ub * K + rank] = s;
            }
        }
        const int filled = total < K ? total : K;
        if (lane >= filled && lane < K) {
            a.top_items[(size_t)ub * K + lane] = -1;
            a.top_scores[(size_t)ub * K + lane] = TK_NEG;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Second form (round 5; which call takes which form: ngacf_eval_topk below): 16 * NW users per workgroup, the ITEM rows staged
// once per workgroup in LDS and read by all of its NW consumer waves (the form above streams the whole item table through L2 once
// per 32 users: 4.9 GB per Gowalla-scale evaluation, 0.66 ms at the L2's rate before any selection), and the score tile TRANSPOSED:
//
//   wave      = 16 users x all items on v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32 (exact products, fp32 accumulate),
//               A operand = 16 item rows from LDS, B operand = the wave's 16 user rows (registers, loaded once)
//   lane      = (user c = lane & 15, row group rg = lane >> 4): accumulator register e holds score(user c, item 4 rg + e) --
//               a lane's scores all belong to ONE user, so that user's threshold is ONE register (the first form keeps 16),
//               its list's entry count is a register (no LDS atomic, no returned slot to wait for) and a survivor is ONE
//               ds_write2_b32; the steady state of a step is v_max3 over the lane's eight scores, one compare, one branch
//   list      = ONE per user (64 entries of (score, item) in LDS, K live + slack) in four quarters, one per lane of the user, each
//               with its own fill count in a register; a user with a full quarter is compacted by its wave (train positives out,
//               the K best by quickselect on 64-bit (score, ~item) keys, dealt round the quarters, threshold raised); scores of
//               the SAME 32-item step that were held back are retried with ">=" (an equal score of the same step may carry a
//               lower item id than the list's K-th entry; a later step's never does: strict ">" is the stable sort's tie rule)
//   stage     = 16 KB of item rows (8 KB with 4 or 12 consumer waves), a ring of TL_NB of them filled by the workgroup's LAST wave,
//               the producer: LDS-DMA for rows and pool flags, a `full` word per buffer published when vmcnt says the stage landed,
//               a `done` word per buffer that every consumer bumps behind its last fragment read -- no barrier after the first
//               (details at the producer's loop); a stage in which every item is a candidate (the usual case) takes the sift
//               without the per-score flag test
// ---------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float tk_f32x4;
constexpr int TL_NB = 4;             // stage buffers in the ring
__host__ __device__ constexpr int tl_stage_bytes(int nw) { return nw == 8 ? 16384 : 8192; }
constexpr int TL_SUB = 16;           // entries per quarter list (a user's list: four quarters, one per lane of the user)

template <typename T, int D, int NW>
__global__ __launch_bounds__(64 * (NW + 1)) void eval_topk_lds_kernel(const TopkArgs a, const int tc) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int RB = D * (int)sizeof(T);          // bytes of a table row
    // Staged rows are NOT padded: their 16-byte chunks are swizzled, chunk k of stage row r at position k ^ sw(r), sw(r) =
    // (r >> 1) & 7 for 128-byte rows, r & 15 for longer ones.  ds_read_b128 is served in four groups of 16 lanes -- lanes
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- each of which must touch 64 distinct banks; with lane =
    // (c, rg) a group holds all 16 rows c, half of them with chunk 4 f + rg and half with 4 f + rg + 1, and no row pitch
    // separates those (144 bytes: 39 % extra LDS cycles by SQ_LDS_BANK_CONFLICT); these two XOR patterns do.
    constexpr int STAGE = tl_stage_bytes(NW);       // bytes of a stage
    constexpr int SI = STAGE / RB;                  // items per stage
    constexpr int STEPS = SI / 32;                  // a step = 32 items = two 16 x 16 score tiles per wave
    constexpr int NB = TL_NB, FR = TL_NB + 4;       // stage buffers; slots of the flag ring (a stage's flags outlive its rows: the sift runs two steps behind the reads)
    constexpr int CPR = RB / 16;                    // 16-byte chunks per row
    constexpr int PIECES = STAGE / 1024;            // 1 KB LDS-DMA pieces per stage
    constexpr int FL = (SI + 63) / 64;              // flag requests per stage (64 items each)
    constexpr int FSLOT = FL * 64;                  // dwords of a stage's flag slot (one dword per item: what a byte-sized LDS-DMA writes)
    constexpr int UW = 16 * NW;                     // users per workgroup
    constexpr int NF = RB / 64;                     // 16-byte operand fragments per row and lane
    static_assert(SI >= 32 && SI % 32 == 0, "a stage holds whole steps");
    extern __shared__ __attribute__((aligned(16))) char tl_lds[];
    char* const stage0 = tl_lds;                                                          // [NB][SI][RB], chunks swizzled
    unsigned int* const pflag = reinterpret_cast<unsigned int*>(tl_lds + NB * STAGE);      // [FR][FSLOT]: the pool's byte of an item, one dword each (stage s in slot s % FR)
    int* const ctr = reinterpret_cast<int*>(pflag + FR * FSLOT);                           // full[NB]: times buffer b was filled; done[NB]: waves that left it, summed
    uint2* const lists = reinterpret_cast<uint2*>(ctr + 2 * NB);                           // [UW][4][TL_SUB + 1] (score bits, item)
    int* const c_item = reinterpret_cast<int*>(lists + UW * 4 * (TL_SUB + 1));             // [UW][tc] first train items of a user
    int* const c_deg = c_item + UW * tc;                                                   // [UW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, rg = lane >> 4;
    const int u0 = blockIdx.x * UW;
    const int us = wave * 16 + c;                    // this lane's user slot in the workgroup
    const int K = a.K;
    const int nst = (a.I + SI - 1) / SI;
    const char* const items = static_cast<const char*>(a.F) + (size_t)a.U * RB;
    // LDS byte addresses of the counters (polled / bumped with explicit ds_ instructions: a plain load could be kept in a register)
    const unsigned int ctr_at = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(ctr);
    auto lds_peek = [&](unsigned int at) -> int {
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(at) : "memory");
        return __builtin_amdgcn_readfirstlane(v);
    };
    // A wait that never ends would hang the GPU, not fail a launch.  Every cycle of waits in this kernel runs through the
    // PRODUCER (consumers wait only for it, it waits only for them), so its poll loop carries the guard: after ~2^22 sleeps (seconds;
    // a stage takes microseconds) it traps and the launch ends in an error the host sees.  (The same guard in the consumers' loop
    // cost 2.5 % of the kernel by its code placement alone.)
    if (tid < 2 * NB) ctr[tid] = 0;
    __syncthreads();                                  // (the only barrier: from here on the waves run on their own)

    // ---- the producer wave: item rows global -> LDS by LDS-DMA, NB - 1 stages ahead of the slowest consumer ---------------------
    // buffer b = s % NB is refilled once all NW consumer waves have bumped done[b] for its previous stage (a wave bumps it right
    // behind the LAST fragment reads of that stage: LDS executes a wave's instructions in order); a stage is published (full[b] =
    // its round) once this wave's vmcnt says its pieces and flags have landed.  The consumers never meet at a barrier: one of
    // them compacting a list (~2 000 cycles) used to hold the other seven at the next stage's barrier -- 0.4 ms of the kernel.
    if (wave == NW) {
        // One stage's turn: request stage s -- rows AND pool flags by LDS-DMA, so that this wave holds no loaded register and
        // the compiler has no reason to put a vmcnt(0) of its own anywhere in the loop (it did, in front of every use of a
        // flag byte: the producer then ran one stage at a time) -- then, vmcnt counting in order: at most this stage's
        // PIECES + FL requests outstanding = the previous stage has landed -> its round number goes up.
        // (A flag byte lands as one zero-extended dword per lane; rows and flags past the table's end are clamped copies of
        // its last item and are never believed: the consumers check j < I wherever a stage reaches beyond it.)
#pragma unroll 1
        for (int s = 0; s <= nst; ++s) {
            if (s < nst) {
                const int b = s % NB;
                if (s >= NB) {
                    const int want = NW * (s / NB);
                    int spins = 0;
                    while (lds_peek(ctr_at + 4 * (NB + b)) < want) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1 << 22)) __builtin_trap();
                    }
                }
                const int j0 = s * SI;
#pragma unroll
                for (int p = 0; p < PIECES; ++p) {
                    const int r = p * (1024 / RB) + lane / CPR;                     // stage row of this lane's chunk
                    const int pc = lane % CPR;                                      // its place in the row as stored
                    const int k = RB == 128 ? pc ^ ((r >> 1) & 7) : pc ^ (r & 15);  // the chunk that belongs there
                    const int j = j0 + r < a.I ? j0 + r : a.I - 1;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(items + (size_t)j * RB + 16 * k),
                                                     (__attribute__((address_space(3))) void*)(stage0 + b * STAGE + p * 1024), 16, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < FL; ++k) {
                    const int j = j0 + lane + 64 * k;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.pool + (j < a.I ? j : a.I - 1)),
                                                     (__attribute__((address_space(3))) void*)(pflag + (s % FR) * FSLOT + 64 * k), 1, 0, 0);
                }
            }
            if (s > 0) {
                if (s < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES + FL) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned int at = ctr_at + 4 * ((s - 1) % NB);
                const int rnd = (s - 1) / NB + 1;
                asm volatile("ds_write_b32 %0, %1" :: "v"(at), "v"(rnd) : "memory");
            }
        }
        return;
    }

    // ---- per-user state: the first tc train items (wave-private: a wave fills and reads only its own users') ------------------
    {
        long long p0_l = 0;                                      // lane i < 16: row of the wave's user i (one round of loads for all 16)
        int deg_l = 0;
        if (lane < 16 && u0 + wave * 16 + lane < a.B) {
            const int64_t u = tk_user(a, u0 + wave * 16 + lane);
            p0_l = a.train_ptr[u];
            deg_l = (int)(a.train_ptr[u + 1] - p0_l);
        }
        if (lane < 16) c_deg[wave * 16 + lane] = deg_l;
        int got[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {                           // 16 independent loads in flight, then 16 LDS writes
            const int deg = __builtin_amdgcn_readlane(deg_l, i);
            const long long p0 = (long long)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(p0_l >> 32), i) << 32) |
                                             (unsigned int)__builtin_amdgcn_readlane((int)p0_l, i));
            got[i] = (lane < deg && lane < tc) ? a.train_items[p0 + lane] : 0x7fffffff;      // (padding: beyond every item id)
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (lane < tc) c_item[(wave * 16 + i) * tc + lane] = got[i];
    }
    // ---- B operand: this lane's user row (users past the batch end take the last one and never select) -------------------------
    uint4 ufrag[NF];
    {
        const int my_u = u0 + us < a.B ? u0 + us : a.B - 1;
        const char* urow = static_cast<const char*>(a.F) + (size_t)tk_user(a, my_u) * RB + 16 * rg;
#pragma unroll
        for (int f = 0; f < NF; ++f) ufrag[f] = *reinterpret_cast<const uint4*>(urow + 64 * f);
    }
    // (The user rows must have ARRIVED before the loop: left pending, the wait the compiler puts in front of their first use
    // would sit inside it.)
#pragma unroll
    for (int f = 0; f < NF; ++f) asm volatile("" :: "v"(ufrag[f].x), "v"(ufrag[f].y), "v"(ufrag[f].z), "v"(ufrag[f].w));
    float tau = u0 + us < a.B ? TK_NEG : 3.402823466e38f;       // the user's threshold, the same in its four lanes
    int cnt = 0;                                                  // appends to THIS LANE's quarter of the user's list (> TL_SUB: some were held back)
    unsigned int pend = 0u;                                       // bit (4 t + e): this lane's score of that register was held back
    // LDS byte address of this lane's quarter (entry TL_SUB: where a held-back score is dumped).  Appends go out as ds_write2_b32
    // with the score and the item number in two UNRELATED registers: a 64-bit store wants them in a register pair, and to have
    // one the compiler copied half of every accumulator right behind its MFMAs -- four moves and a full matrix-core drain per turn.
    const unsigned int mysub_at = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(lists + (us * 4 + rg) * (TL_SUB + 1));

    // The wave compacts the list of its user ui (0..15; n entries), one entry per lane.  All arguments wave-uniform.
    //   * train positives out: the user's first tc train items sit in LDS, ascending, padded -- a 4-ary search (three round trips
    //     for 64 entries; the first form's linear scan cost ~4 000 cycles per compaction); beyond them a binary search in memory
    //   * (score desc, item asc) as ONE unsigned 64-bit key per entry (order-preserving score bits, ~item): no ties
    //   * SORTED = false (in the stream): only the K best are wanted, in any order -> the wave finds the K-th largest key by
    //     quickselect (pivot = a candidate lane's key, v_cmp_gt_u64 + ballot + popcount per round, ~8 rounds) and the keepers
    //     move to the front by their position in the keep mask; SORTED = true (the end): every entry ranked against all others
    // Returns the new entry count; `own` = the K-th best score when K entries remain.
    auto compact = [&](int ui, float& own, auto sorted) -> int {
        constexpr bool SORTED = decltype(sorted)::value;
        own = TK_NEG;
        const int usl = wave * 16 + ui;
        uint2* const L = lists + usl * 4 * (TL_SUB + 1);
        // lane i looks at entry (i & 15) of quarter (i >> 4); the quarters' counts sit in the user's four lanes
        const int n0 = __builtin_amdgcn_readlane(cnt, ui), n1 = __builtin_amdgcn_readlane(cnt, ui + 16),
                  n2 = __builtin_amdgcn_readlane(cnt, ui + 32), n3 = __builtin_amdgcn_readlane(cnt, ui + 48);
        const int nq = rg == 0 ? n0 : (rg == 1 ? n1 : (rg == 2 ? n2 : n3));
        const bool live = c < (nq < TL_SUB ? nq : TL_SUB);
        const uint2 en = live ? L[rg * (TL_SUB + 1) + c] : make_uint2(__float_as_uint(TK_NEG), 0x7fffffffu);
        const int i_t = (int)en.y;
        const int* const ci = c_item + usl * tc;
        int lo = 0;
        if (tc == 32) lo = ci[15] < i_t ? 16 : 0;
        for (int q = tc == 64 ? 4 : 1; q >= 1; q >>= 2) {            // a span of 16 q entries -> 4 q
            const int p1 = ci[lo + 4 * q - 1], p2 = ci[lo + 8 * q - 1], p3 = ci[lo + 12 * q - 1];
            lo += 4 * q * ((p1 < i_t ? 1 : 0) + (p2 < i_t ? 1 : 0) + (p3 < i_t ? 1 : 0));
        }
        const int4 fin = *reinterpret_cast<const int4*>(ci + lo);
        bool member = fin.x == i_t || fin.y == i_t || fin.z == i_t || fin.w == i_t;
        const int deg = c_deg[usl];
        if (deg > tc && live && !member && i_t > ci[tc - 1]) {
            const int64_t p0 = a.train_ptr[tk_user(a, u0 + usl)];
            int l2 = tc, h2 = deg;                                   // per-lane binary search over the uncached tail
            while (l2 < h2) {
                const int mid = (l2 + h2) >> 1;
                if (a.train_items[p0 + mid] < i_t) l2 = mid + 1; else h2 = mid;
            }
            member = l2 < deg && a.train_items[p0 + l2] == i_t;
        }
        const bool valid = live && !member;
        const unsigned long long vm = __ballot(valid);
        const int nvalid = __popcll(vm);
        const int ncnt = nvalid < K ? nvalid : K;
        const unsigned int ob = en.x ^ ((unsigned int)((int)en.x >> 31) | 0x80000000u);       // float order -> unsigned order
        const unsigned long long key = valid ? (((unsigned long long)ob << 32) | (unsigned int)~i_t) : 0ull;
        if constexpr (SORTED) {
            int rank = 0;
#pragma unroll
            for (int t = 0; t < TK_CAP; ++t) {
                const unsigned long long k2 = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(key >> 32), t) << 32) |
                                              (unsigned int)__builtin_amdgcn_readlane((int)key, t);
                rank += k2 > key ? 1 : 0;
            }
            if (valid && rank < K) L[(rank & 3) * (TL_SUB + 1) + (rank >> 2)] = en;      // entry r of the ranking: quarter r & 3, place r >> 2
            if (ncnt == K) {
                const unsigned long long last = __ballot(valid && rank == K - 1);
                own = __int_as_float(__builtin_amdgcn_readlane((int)en.x, __builtin_ctzll(last)));
            }
        } else {
            unsigned long long keep = vm;
            if (nvalid > K) {
                unsigned long long cand = vm;
                int need = K;
                unsigned long long pk;
                for (;;) {                                               // cand shrinks every round: at most 64
                    const int p = __builtin_ctzll(cand);
                    pk = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(key >> 32), p) << 32) |
                         (unsigned int)__builtin_amdgcn_readlane((int)key, p);
                    const unsigned long long G = __ballot(key > pk) & cand;
                    const int g = __popcll(G);
                    if (g == need - 1) break;
                    if (g >= need) cand = G;
                    else { need -= g + 1; cand &= ~G & ~(1ull << p); }
                }
                keep = __ballot(key >= pk) & vm;                        // exactly K lanes
                own = __uint_as_float((unsigned int)(pk >> 32) ^ (((unsigned int)(pk >> 32) >> 31) ? 0x80000000u : 0xffffffffu));
            } else if (nvalid == K) {
                float sm = valid ? __uint_as_float(en.x) : 3.402823466e38f;     // rare: exactly K left -> their smallest score
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) sm = fminf(sm, __shfl_xor(sm, o, 64));
                own = sm;
            }
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned int)(keep >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)keep, 0u));
            if ((keep >> lane) & 1ull) L[(pos & 3) * (TL_SUB + 1) + (pos >> 2)] = en;    // dealt round the four quarters
        }
        return ncnt;
    };

    // One accumulator register of the wave: the lanes whose score passed append it to their user's list.
    // (Each of a user's four lanes appends to its OWN quarter of the list: the first build kept one count per user and found a
    // lane's slot from the ballot mask -- 15 vector + 11 scalar instructions per survivor, three quarters of the kernel's
    // issue slots at ~310 survivors per user.)
    auto push = [&](bool hit, float sc, int j, unsigned int bit) {
        if (__ballot(hit) == 0ull) return;                        // nobody's did
        if (hit) {
            const unsigned int at = mysub_at + 8u * (unsigned int)(cnt < TL_SUB ? cnt : TL_SUB);
            asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(at), "v"(sc), "v"(j) : "memory");
            pend |= cnt >= TL_SUB ? bit : 0u;
            ++cnt;
        }
    };

    // Three steps in flight per wave (a step = 32 items = two 16-item tiles): while step g is sifted, step g + 1 is on the
    // matrix cores and the operand fragments of step g + 2 are on their way from LDS.  The steps run on across the stages; a
    // stage is written, and the barrier passed, just before its first fragments are read -- by then every wave has the
    // fragments of the stage two back in registers (its multiply precedes the barrier), so two buffers suffice.
    // The loop is ISSUE-bound (SQ counters of the first build: 52 vector + 23 scalar + 19 branch instructions per wave and
    // step, 0.62 ms with nothing ever selected), so the steady state is pared down: the turn is unrolled by two (the two
    // accumulator sets swap roles by name, no copies), the eight scores of a lane are folded to their maximum (v_max3) and
    // ONE compare + ONE never-taken branch decides whether anybody in the wave has anything to do (the per-register
    // compares, item numbers, pool flags live behind it), addresses are a per-lane constant + a scalar.
    uint4 frag[2][NF];                 // step g + 2
    int foff[NF];                      // this lane's byte offsets into a stage row block: row c, chunk (4 f + rg) ^ sw(c)
#pragma unroll
    for (int f = 0; f < NF; ++f) foff[f] = c * RB + 16 * ((4 * f + rg) ^ (RB == 128 ? (c >> 1) & 7 : c));
    unsigned int okbits = 0u;          // bit (s & 31): every item of stage s is a candidate inside the table (wave-uniform)
    // The wave moves on to stage st: the previous stage's buffer is released (every fragment read of it has been ISSUED, and LDS
    // executes this wave's instructions in order), then the wave waits for the producer's word on st.
    auto enter_stage = [&](int st) {
        if (st > 0 && lane == 0) {
            const unsigned int at = ctr_at + 4 * (NB + (st - 1) % NB);
            const int one = 1;
            asm volatile("ds_add_u32 %0, %1" :: "v"(at), "v"(one) : "memory");
        }
        const int want = st / NB + 1;
        while (lds_peek(ctr_at + 4 * (st % NB)) < want) __builtin_amdgcn_s_sleep(1);
        bool zero = false;
#pragma unroll
        for (int k = 0; k < FL; ++k) zero |= lane + 64 * k < SI && pflag[(st % FR) * FSLOT + lane + 64 * k] == 0u;
        const bool ok = __ballot(zero) == 0ull && (st + 1) * SI <= a.I;
        okbits = (okbits & ~(1u << (st & 31))) | ((ok ? 1u : 0u) << (st & 31));
    };
    // (what the loop keeps per turn is counted in scalar instructions: the ring offset of the step being read and the two
    // steps-into-their-stage counters run along instead of being divided out of g)
    int rd_off = 0;                    // byte offset in the ring of the step that is read next
    int rd_in = 0, sf_in = 0, sf_stage = 0;   // steps into its stage of the step read next / sifted next; the sifted step's stage
    auto read_step = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int f = 0; f < NF; ++f) frag[t][f] = *reinterpret_cast<const uint4*>(stage0 + rd_off + foff[f] + t * 16 * RB);
        rd_off += 32 * RB;
        if (rd_off == NB * STAGE) rd_off = 0;
        if (++rd_in == STEPS) rd_in = 0;
    };
    // r[t][e] in lane (c, rg) = score(user c, item 32 gs + 16 t + 4 rg + e)
    auto multiply = [&](tk_f32x4 (&r)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            r[t] = tk_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const uint4 v = frag[t][f];
                if constexpr (BF) {
                    r[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tk_bf16x8, v), __builtin_bit_cast(tk_bf16x8, ufrag[f]), r[t], 0, 0, 0);
                } else {
                    r[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v.x), __uint_as_float(ufrag[f].x), r[t], 0, 0, 0);
                    r[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v.y), __uint_as_float(ufrag[f].y), r[t], 0, 0, 0);
                    r[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v.z), __uint_as_float(ufrag[f].z), r[t], 0, 0, 0);
                    r[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v.w), __uint_as_float(ufrag[f].w), r[t], 0, 0, 0);
                }
            }
        }
    };
    // Somebody in the wave has a score above its threshold (or the stage holds non-candidates): the registers one by one.
    auto sift_slow = [&](const tk_f32x4 (&acc)[2], int g, bool sok) {
        int jb = g * 32 + 4 * rg;
        asm volatile("" : "+v"(jb));                                    // (item numbers are computed HERE, not in the steady state)
        if (sok) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) push(acc[t][e] > tau, acc[t][e], jb + 16 * t + e, 1u << (4 * t + e));
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint4 fw = *reinterpret_cast<const uint4*>(pflag + (sf_stage % FR) * FSLOT + sf_in * 32 + 16 * t + 4 * rg);
                const unsigned int fwe[4] = {fw.x, fw.y, fw.z, fw.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    push(acc[t][e] > tau && fwe[e] != 0u && jb + 16 * t + e < a.I, acc[t][e], jb + 16 * t + e, 1u << (4 * t + e));
            }
        }
        while (__ballot(pend != 0u) != 0ull) {                          // some list overflowed: compact it, raise tau, retry what was held back
            const unsigned long long om = __ballot(cnt > TL_SUB);
            unsigned int over = (unsigned int)(om | (om >> 16) | (om >> 32) | (om >> 48)) & 0xffffu;      // users with a full quarter
            while (over) {
                const int ui = __builtin_ctz(over);
                over &= over - 1u;
                float own;
                const int n = compact(ui, own, std::false_type{});
                if (c == ui) {
                    cnt = (n + 3 - rg) >> 2;
                    if (n == K) tau = fmaxf(tau, own);
                }
            }
            const unsigned int p = pend;
            pend = 0u;
            const float tb = tk_below(tau);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    push(((p >> (4 * t + e)) & 1u) && acc[t][e] > tb, acc[t][e], jb + 16 * t + e, 1u << (4 * t + e));
        }
    };
    const int total = nst * STEPS;
    bool sok_cur = false;
    float taue = TK_NEG;
    // one turn: `cur` holds step g (sifted now), `nxt` receives step g + 1, the fragments of step g + 2 are requested.
    // MUL / RD / SIFT: which of the three the turn has (all of them between the loop's head and its last two turns)
    auto turn = [&](auto mul, auto rd, auto sift, tk_f32x4 (&cur)[2], tk_f32x4 (&nxt)[2], int g) {
        if constexpr (decltype(mul)::value) multiply(nxt);              // its fragments were read a turn ago
        if constexpr (decltype(rd)::value) {
            if (rd_in == 0) enter_stage((g + 2) / STEPS);               // step g + 2 opens a stage
            read_step();
        }
        if constexpr (decltype(sift)::value) {
            if (sf_in == 0) {                                           // the sift enters a stage: may its scores skip the flag test?
                sok_cur = (okbits >> (sf_stage & 31)) & 1u;
                taue = sok_cur ? tau : TK_NEG;                          // (a stage with non-candidates always takes the slow path)
            }
            const float m = fmaxf(fmaxf(fmaxf(fmaxf(cur[0][0], cur[0][1]), cur[0][2]), fmaxf(fmaxf(cur[0][3], cur[1][0]), cur[1][1])),
                                  fmaxf(fmaxf(cur[1][2], cur[1][3]), TK_NEG));
            if (__builtin_expect(__ballot(m > taue) != 0ull, 0)) {
                sift_slow(cur, g, sok_cur);
                taue = sok_cur ? tau : TK_NEG;
            }
            if (++sf_in == STEPS) { sf_in = 0; ++sf_stage; }
        }
    };
    constexpr std::true_type Y{};
    constexpr std::false_type N_{};

    enter_stage(0);
    read_step();                                                        // step 0
    tk_f32x4 accA[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accB[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // g = -1: multiply step 0 into B, request step 1 (when there is one); then turns 0 .. total - 1 with A / B swapping roles:
    // even g sifts B and fills A.  The last two turns have nothing left to request, the last one nothing to multiply.
    multiply(accB);
    if (total > 1) {
        if (rd_in == 0) enter_stage(1 / STEPS);
        read_step();
    }
    int g = 0;
#pragma unroll 1
    for (; g + 3 < total; g += 2) {                                      // both turns complete: g + 1 + 2 < total
        turn(Y, Y, Y, accB, accA, g);
        turn(Y, Y, Y, accA, accB, g + 1);
    }
    for (; g < total; ++g) {                                            // at most three turns, each with what is left
        const bool even = (g & 1) == 0;
        if (g + 2 < total) { if (even) turn(Y, Y, Y, accB, accA, g); else turn(Y, Y, Y, accA, accB, g); }
        else if (g + 1 < total) { if (even) turn(Y, N_, Y, accB, accA, g); else turn(Y, N_, Y, accA, accB, g); }
        else { if (even) turn(N_, N_, Y, accB, accA, g); else turn(N_, N_, Y, accA, accB, g); }
    }

    // ---- every list compacted (valid entries only, best first, at most K) and written out ------------------------------------------
    for (int ui = 0; ui < 16; ++ui) {
        float own;
        const int n = compact(ui, own, std::true_type{});
        const int usl = wave * 16 + ui, ub = u0 + usl;
        if (ub >= a.B) continue;
        const int64_t uraw = a.users[ub];
        const bool isuser = uraw >= 0 && uraw < a.U;
        if (!isuser && lane == 0 && a.bad) atomicAdd(a.bad, 1);
        if (lane < K) {
            const uint2 en = (isuser && lane < n) ? lists[(usl * 4 + (lane & 3)) * (TL_SUB + 1) + (lane >> 2)] : make_uint2(__float_as_uint(TK_NEG), 0xffffffffu);
            a.top_items[(size_t)ub * K + lane] = (int)en.y;
            a.top_scores[(size_t)ub * K + lane] = __uint_as_float(en.x);
        }
    }
}

template <typename T, int D, int NW>
static int launch_topk_lds(const TopkArgs& a, hipStream_t st) {
    constexpr int RB = D * (int)sizeof(T), STAGE = tl_stage_bytes(NW), SI = STAGE / RB, UW = 16 * NW;
    if constexpr (SI < 32) {
        return NGACF_E_SHAPE;                                   // (512-byte rows take the 16 KB stages of the 8-wave form)
    } else {
        const int tc = NW == 12 ? 16 : 32;                      // train items of a user cached in LDS (what the ring and the lists leave)
        const size_t lds = (size_t)TL_NB * STAGE + (size_t)(TL_NB + 4) * ((SI + 63) / 64) * 64 * 4 + 2 * TL_NB * 4 + (size_t)UW * 4 * (TL_SUB + 1) * 8 +
                           (size_t)UW * tc * 4 + (size_t)UW * 4;
        auto k = eval_topk_lds_kernel<T, D, NW>;
        if (lds > 64 * 1024) {                                 // (every call: the attribute belongs to the current device)
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return static_cast<int>(e);
        }
        hipLaunchKernelGGL(k, dim3((a.B + UW - 1) / UW), dim3(64 * (NW + 1)), lds, st, a, tc);
        NGACF_CHECK_LAUNCH();
        return 0;
    }
}

}  // namespace ngacf

using namespace ngacf;

extern "C" int ngacf_eval_topk(const void* F, int dtype, const int64_t* users, int B, int U, int I, int d,
                               const uint8_t* pool_mask, const int64_t* train_ptr, const int32_t* train_items,
                               int K, int32_t* top_items, float* top_scores, int32_t* bad_index, void* stream) {
    if (B < 0 || U <= 0 || I <= 0 || K <= 0 || K > TK_MAXK) return NGACF_E_ARG;
    if (B == 0) return 0;
    if (!F || !users || !pool_mask || !train_ptr || !train_items || !top_items || !top_scores) return NGACF_E_ARG;
    TopkArgs a;
    a.F = F; a.users = users; a.pool = pool_mask; a.train_ptr = train_ptr; a.train_items = train_items;
    a.top_items = top_items; a.top_scores = top_scores; a.bad = bad_index; a.B = B; a.U = U; a.I = I; a.K = K;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // Which form, how many consumer waves (measured, 40 981 items, iid table, ms for 64 / 1 024 / 16 384 / 29 858 users): 128-byte rows
    // (bf16, d = 64) -- wave-private 1.00 / 1.02 / 1.14 / 1.91, LDS-shared with 4 waves 0.80 / 0.84 / 0.84 / 0.93, with 8 waves
    // 0.89 (2 048 users) / 0.89 / 0.90: the LDS-shared form always, 64 users per workgroup while those fit one per CU, 128 while
    // THOSE fit the chip in one round, 192 while those do (a second, partly filled round doubles the time: Yelp's 45 919 users
    // 1.16 ms against 1.89), 128 beyond.  Longer rows (fp32, d = 128: the matrix cores bound the kernel; fp32 d = 64 at 64 / 1 024
    // users: wave-private 1.26, LDS-shared 1.44 / 1.49; at 29 858: 3.05 against 1.96): the wave-private form below 12 288 users.
    // NGACF_TOPK_FORM=lds|wave and NGACF_TOPK_LDS_WAVES=4|8|12 force a choice (tests, A/B runs); read per call.
    {
        const char* form = getenv("NGACF_TOPK_FORM");
        const char* lw = getenv("NGACF_TOPK_LDS_WAVES");
        const int rb = d * (dtype == NGACF_F32 ? 4 : 2);
        const bool use_lds = form ? (form[0] == 'l') : (rb == 128 || B >= 96 * 128);
        if (use_lds) {
            int nwl = lw ? atoi(lw) : (rb == 128 && B <= 64 * 256 ? 4 : ((B + 127) / 128 <= 256 ? 8 : ((B + 191) / 192 <= 256 ? 12 : 8)));
            if (nwl != 4 && nwl != 8 && nwl != 12) nwl = 8;
            if (d * (dtype == NGACF_F32 ? 4 : 2) >= 512) nwl = 8;
#define NGACF_TKL(T, D) return nwl == 4 ? launch_topk_lds<T, D, 4>(a, st) : (nwl == 8 ? launch_topk_lds<T, D, 8>(a, st) : launch_topk_lds<T, D, 12>(a, st))
            if (dtype == NGACF_F32 && d == 64) NGACF_TKL(float, 64);
            if (dtype == NGACF_F32 && d == 128) NGACF_TKL(float, 128);
            if (dtype == NGACF_BF16 && d == 64) NGACF_TKL(bf16_t, 64);
            if (dtype == NGACF_BF16 && d == 128) NGACF_TKL(bf16_t, 128);
#undef NGACF_TKL
            return NGACF_E_SHAPE;
        }
    }
    const int nw_env = getenv("NGACF_TOPK_WAVES") ? atoi(getenv("NGACF_TOPK_WAVES")) : 0;
    // waves per group of 32 users: one list per user (1) as long as the groups alone fill the chip's 1024 SIMDs; small user
    // batches split every group's items over 2 or 4 waves instead
    const int groups = (B + 31) / 32;
    // (measured, 29 858 users x 40 981 items, bf16 / fp32: 4 waves 2.18 / 3.33 ms, 2 waves 2.63 / 4.83 ms, 1 wave 1.85 / 2.99 ms)
    const int nw = nw_env == 1 || nw_env == 2 || nw_env == 4 ? nw_env : (groups >= 768 ? 1 : 4);
    const dim3 grid(groups), block(64 * nw);
#define NGACF_TK(T, D)                                                                                        \
    do {                                                                                                      \
        if (nw == 1) hipLaunchKernelGGL((eval_topk_kernel<T, D, 1>), grid, block, 0, st, a);                  \
        else if (nw == 2) hipLaunchKernelGGL((eval_topk_kernel<T, D, 2>), grid, block, 0, st, a);             \
        else hipLaunchKernelGGL((eval_topk_kernel<T, D, 4>), grid, block, 0, st, a);                          \
    } while (0)
    if (dtype == NGACF_F32 && d == 64) NGACF_TK(float, 64);
    else if (dtype == NGACF_F32 && d == 128) NGACF_TK(float, 128);
    else if (dtype == NGACF_BF16 && d == 64) NGACF_TK(bf16_t, 64);
    else if (dtype == NGACF_BF16 && d == 128) NGACF_TK(bf16_t, 128);
    else return NGACF_E_SHAPE;
#undef NGACF_TK
    NGACF_CHECK_LAUNCH();
    return 0;
}
