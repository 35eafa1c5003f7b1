// BPR pair sampler: for every row of a batch, one positive drawn uniformly from the
// user's train positives and one negative drawn uniformly from (item pool - the user's train positives).
// Replaces data/loadGowalla.py:63-77 (per-user Python sets of negatives, 60-100 GB of host memory, random.choice per
// row) with rejection sampling against the sorted CSR row of the user: one thread per batch row, counter-hash
// randomness (same lowbias32 family as the dropout masks; host restatement in ngacf_amd/rng.py), binary search for
// membership.  The batch's user ids are read at users[offset + row], offset = host value + optional device counter,
// so a captured hipGraph walks through the epoch's sequential batches (train_eval_Gowalla.py:109-115) by itself.
#include "ngacf_common.hpp"

namespace ngacf {

constexpr int SAMPLER_MAX_ROUNDS = 64;

__global__ __launch_bounds__(256) void sample_pairs_kernel(const int64_t* __restrict__ users, long long n_rows, long long offset,
                                                           const int64_t* __restrict__ offset_dev,
                                                           const int64_t* __restrict__ train_ptr, int n_users,
                                                           const int32_t* __restrict__ train_items,
                                                           const int32_t* __restrict__ pool_items, int P,
                                                           uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                           int64_t* __restrict__ u_out, int64_t* __restrict__ pos_out,
                                                           int64_t* __restrict__ neg_out, int* __restrict__ unresolved, int B) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= B) return;
    const uint32_t salt = salt_of(seed + (seed_dev ? *seed_dev : 0ull), 0x5A17u);
    const long long off = offset + (offset_dev ? *offset_dev : 0ll);
    // a row past the end of the interaction list (or a user id outside the CSR) is never dereferenced: it samples
    // for the last valid row / user 0 and is counted in *unresolved, which the epoch loop checks
    long long at = off + r;
    bool oob = at < 0 || at >= n_rows;
    if (oob) at = n_rows - 1;
    int64_t u = users[at];
    if (u < 0 || u >= n_users) { oob = true; u = 0; }
    if (oob) atomicAdd(unresolved, 1);
    const int64_t p0 = train_ptr[u];
    const int deg = (int)(train_ptr[u + 1] - p0);
    // positive: index floor(bits * deg / 2^32)
    const uint32_t bp = lowbias32(((uint32_t)r * RNG_K1) ^ salt);
    const int pi = (int)(((uint64_t)bp * (uint64_t)(deg > 0 ? deg : 1)) >> 32);
    int neg = -1;
    for (int round = 0; round < SAMPLER_MAX_ROUNDS; ++round) {
        const uint32_t bn = lowbias32(((uint32_t)r * RNG_K1) ^ ((uint32_t)(round + 1) * RNG_K2 + salt));
        const int cand = pool_items[(int)(((uint64_t)bn * (uint64_t)P) >> 32)];
        int lo = 0, hi = deg;                              // is cand one of the user's train positives?
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (train_items[p0 + mid] < cand) lo = mid + 1; else hi = mid;
        }
        if (lo >= deg || train_items[p0 + lo] != cand) { neg = cand; break; }
    }
    if (neg < 0) { atomicAdd(unresolved, 1); neg = pool_items[0]; }
    u_out[r] = u;
    pos_out[r] = deg > 0 ? (int64_t)train_items[p0 + pi] : 0;
    neg_out[r] = (int64_t)neg;
}

}  // namespace ngacf

using namespace ngacf;

extern "C" int ngacf_sample_pairs(const int64_t* users, long long n_rows, long long offset, const int64_t* offset_dev,
                                  const int64_t* train_ptr, int n_users, const int32_t* train_items,
                                  const int32_t* pool_items, int P, uint64_t seed, const uint64_t* seed_dev,
                                  int64_t* u_out, int64_t* pos_out, int64_t* neg_out, int* unresolved, int B, void* stream) {
    if (B < 0 || P <= 0 || offset < 0 || n_rows < 0 || n_users <= 0) return NGACF_E_ARG;
    if (B > 0 && n_rows == 0) return NGACF_E_ARG;
    if (B == 0) return 0;
    if (!users || !train_ptr || !train_items || !pool_items || !u_out || !pos_out || !neg_out || !unresolved) return NGACF_E_ARG;
    hipLaunchKernelGGL(sample_pairs_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       users, n_rows, offset, offset_dev, train_ptr, n_users, train_items, pool_items, P, seed, seed_dev,
                       u_out, pos_out, neg_out, unresolved, B);
    NGACF_CHECK_LAUNCH();
    return 0;
}
