// Dense (optionally masked) multi-head attention between node sets, forward and backward.
// Replaces graphattention/modules.py:32-46 `attention` as called by MultiHeadAttention.forward (:92-107):
//
//     scores = q k^T / sqrt(d_k);  scores.masked_fill(mask == 0, -1e9);  softmax over keys;  dropout;  @ v
//
// for h heads of width d_k = dim / h over Nq queries x Nk keys (ATTLayer: users x items and items x users; ATTLayer_mask:
// all N nodes against all N nodes with the adjacency as mask).  The reference materialises the [h, Nq, Nk] score tensor
// twice (scores and probabilities: 2 x 220 MB at ml100k, N = 2625) plus the autograd copies; here it never exists:
// flash-style, one workgroup per (head, 32-query block), keys streamed in stages through LDS with a running max / sum.
//
// Q, K, V are read straight from the [N, dim] outputs of the linear layers (head h = columns h*d_k ..): the
// view/transpose/contiguous of :96-98 and :104 costs nothing.  Exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32: parity
// with the fp32 reference); the kernels below describe their tiling.  A lane owns queries (forward, dQ pass) or keys (dK / dV
// pass), so the softmax reductions stay in registers and no atomics are needed anywhere.
// The work is exp- and VALU-bound, not MFMA-bound (d_k = 8: 32 flop per score against one v_exp_f32): DESIGN.md.
// (Rounds 2-3 also carried a form with bf16 matrix-core operands on 32x32x16 tiles: slower than this one in both directions --
// 80 against 67 us forward, 513 against 174 us backward at ml100k -- and less accurate, so it is gone: VERDICT r3 item 7.)
#include "ngacf_common.hpp"

namespace ngacf {

typedef float mha_f32x16 __attribute__((ext_vector_type(16)));

struct MhaArgs {
    const float* Q;        // [Nq][dim]
    const float* K;        // [Nk][dim]
    const float* V;        // [Nk][dim]
    const uint32_t* mask;  // [Nq][ceil(Nk/32)] bit (k & 31) of word k / 32: 0 = masked (score -1e9); NULL = no mask
    float* O;              // [Nq][dim]
    float* LSE;            // [Nq][heads][2]  running max m and sum l = sum_k exp(s - m) of the (scaled, masked) scores, kept apart:
                           // a fully masked row has m = -1e9, where m + log(l) would lose log(l) to fp32 rounding
    const float* dO;       // backward
    float* dQ;
    float* dK;
    float* dV;
    float* delta;          // [Nq][heads]  dO . O per head
    int Nq, Nk, dim, heads;
    float scale;
    uint32_t drop_thr;
    float drop_scale;
    uint64_t seed;
    const uint64_t* seed_dev;
};

constexpr float MHA_MASKED = -1e9f;                 // modules.py:36

// keep-multiplier of attention probability (query q, key k, head salt): the edge-mask hash of the sparse path
__device__ __forceinline__ float drop_mul(const MhaArgs& a, uint32_t salt_h, int q, int k) {
    return edge_bits(salt_h, (uint32_t)q, (uint32_t)k) >= a.drop_thr ? a.drop_scale : 0.f;
}


// ---------------------------------------------------------------------------------------------------------
// forward, exact fp32 matrix cores, second form (round 3): v_mfma_f32_16x16x4_f32 and K / V tiles staged through LDS.
//   * workgroup = NW waves per (head, 32-query block); a STAGE = 32 * NW keys whose K and V rows every thread fetches once
//     (16-byte pieces, coalesced per row) into registers while the previous stage is computed, then parks in LDS (K transposed:
//     the score product's A operand becomes a conflict-free ds_read_b32) -- one barrier per stage, double buffered; the first
//     form issued 16 dependent dword loads of V per wave and tile straight from global.
//   * S^T block (16 keys x 16 queries) = K_blk . Q_blk^T, d_k / 4 MFMAs; lane (n = l & 15, g = l >> 4) then holds the scores of
//     query n with keys 4g .. 4g+3 -- and register r of all lanes IS the B operand (keys {r, 4+r, 8+r, 12+r} x queries) of the
//     next product O^T += V^T . P^T, as in the 32x32 form: no LDS round trip for P.  With 16-row output blocks a d_k = 8 head
//     wastes half of the PV matrix-core rows instead of three quarters (768 instead of 1280 matrix-pipe cycles per 32 x 32 tile).
//   * a lane owns TWO queries (n and 16 + n) and sees 8 of a tile's 32 keys for each: max / sum are 8 registers + two exchanges
//     (lanes 16 and 32 apart); scale, mask and tail tests run only where they can matter.
// (Same semantics and outputs as the 32x32 form of round 2 it replaced.)
// ---------------------------------------------------------------------------------------------------------
typedef float mha_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lane_xor16(float v, int g) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((g & 1) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v, int g) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((g & 2) ? r[0] : r[1]);
}

template <int DK, int NW, bool DROP, bool MASK>
__global__ __launch_bounds__(NW * 64) void mha_fwd16_kernel(const MhaArgs a) {
    constexpr int SK = 32 * NW;                 // keys per stage
    constexpr int NT = 64 * NW;
    constexpr int FB = (DK + 15) / 16;          // 16-feature blocks of the output
    constexpr int KQ = DK / 4;                  // k-steps of the score product
    constexpr int PC = DK / 8;                  // 16-byte pieces per thread, stage and matrix  (SK * DK / 4 / NT)
    extern __shared__ __attribute__((aligned(16))) float mha_lds[];
    float* const Kt = mha_lds;                  // [2][DK][SK]   transposed key tile
    float* const Vs = mha_lds + 2 * DK * SK;    // [2][SK][DK]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int qblocks = (a.Nq + 31) / 32;
    const int h = blockIdx.x / qblocks, q0 = (blockIdx.x % qblocks) * 32;
    const uint32_t salt_h = DROP ? salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 1u + h) : 0u;
    const int mwords = (a.Nk + 31) >> 5;
    int qi[2];
    float qreg[2][KQ];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qi[qb] = min(q0 + 16 * qb + n, a.Nq - 1);
#pragma unroll
        for (int t = 0; t < KQ; ++t) qreg[qb][t] = a.Q[(size_t)qi[qb] * a.dim + h * DK + 4 * t + g];
    }
    mha_f32x4 o[FB][2];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[fb][qb] = mha_f32x4{0.f, 0.f, 0.f, 0.f};
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};

    float4 kst[PC], vst[PC];
    auto fetch = [&](int s) {                   // stage s: this thread's pieces of K and V (rows beyond Nk: the last row, masked later)
#pragma unroll
        for (int i = 0; i < PC; ++i) {
            const int p = tid + i * NT, key = p / (DK / 4), c = p % (DK / 4);
            const size_t row = (size_t)min(s * SK + key, a.Nk - 1) * a.dim + h * DK + 4 * c;
            kst[i] = *reinterpret_cast<const float4*>(a.K + row);
            vst[i] = *reinterpret_cast<const float4*>(a.V + row);
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PC; ++i) {
            const int p = tid + i * NT, key = p / (DK / 4), c = p % (DK / 4);
            float* kt = Kt + buf * DK * SK + (4 * c) * SK + key;
            kt[0] = kst[i].x; kt[SK] = kst[i].y; kt[2 * SK] = kst[i].z; kt[3 * SK] = kst[i].w;
            *reinterpret_cast<float4*>(Vs + buf * SK * DK + key * DK + 4 * c) = vst[i];
        }
    };
    const int nstage = (a.Nk + SK - 1) / SK;
    fetch(0);
    park(0);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) fetch(s + 1);       // in flight under this stage's arithmetic
        const int k0 = s * SK + 32 * wv;        // this wave's 32 keys
        if (k0 < a.Nk) {
            const float* kt = Kt + buf * DK * SK + 32 * wv;
            const float* vs = Vs + buf * SK * DK + 32 * wv * DK;
            mha_f32x4 sc[2][2];                 // [kb][qb]: keys 16 kb + 4 g + r, query 16 qb + n
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float ka[KQ];
#pragma unroll
                for (int t = 0; t < KQ; ++t) ka[t] = kt[(4 * t + g) * SK + 16 * kb + n];
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    mha_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < KQ; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[t], qreg[qb][t], acc, 0, 0, 0);
                    sc[kb][qb] = acc;
                }
            }
            const bool tail = k0 + 32 > a.Nk;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                uint32_t mw = 0xffffffffu;
                if (MASK) mw = a.mask[(size_t)qi[qb] * mwords + (k0 >> 5)];
                float tmax = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kk = 16 * kb + 4 * g + r;
                        float v = sc[kb][qb][r] * a.scale;
                        if (MASK) v = ((mw >> kk) & 1u) ? v : MHA_MASKED;             // masked_fill(mask == 0, -1e9)
                        if (tail) v = k0 + kk < a.Nk ? v : -INFINITY;
                        sc[kb][qb][r] = v;
                        tmax = fmaxf(tmax, v);
                    }
                tmax = fmaxf(tmax, lane_xor16(tmax, g));
                tmax = fmaxf(tmax, lane_xor32(tmax, g));
                const float mnew = fmaxf(m[qb], tmax);
                const float corr = __expf(m[qb] - mnew);                              // exp(-inf) = 0 on the first tile
                float psum = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = __expf(sc[kb][qb][r] - mnew);
                        psum += p;
                        if (DROP) p *= drop_mul(a, salt_h, qi[qb], k0 + 16 * kb + 4 * g + r);
                        sc[kb][qb][r] = p;
                    }
                psum += lane_xor16(psum, g);
                psum += lane_xor32(psum, g);
                l[qb] = l[qb] * corr + psum;
                m[qb] = mnew;
#pragma unroll
                for (int fb = 0; fb < FB; ++fb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[fb][qb][r] *= corr;
            }
            // O^T += V^T . P^T: for register r of block kb the contraction runs over the keys 16 kb + 4 k + r, k = 0..3
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int fb = 0; fb < FB; ++fb) {
                        const float va = (16 * fb + n < DK) ? vs[(16 * kb + 4 * g + r) * DK + 16 * fb + n] : 0.f;
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb)
                            o[fb][qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(va, sc[kb][qb][r], o[fb][qb], 0, 0, 0);
                    }
        }
        if (s + 1 < nstage) park(buf ^ 1);      // (every wave left that buffer at the previous barrier)
        __syncthreads();
    }
    // merge the NW waves' (max, sum, O) of every query in wave order (deterministic); the stage buffers are free now
    float* const mlb = mha_lds;                                   // [2][NW][2][16]
    float* const ob = mha_lds + 2 * NW * 32;                      // [NW][FB * 2 * 4][64]
    if (g == 0) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            mlb[(0 * NW + wv) * 32 + 16 * qb + n] = m[qb];
            mlb[(1 * NW + wv) * 32 + 16 * qb + n] = l[qb];
        }
    }
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[(wv * FB * 8 + (fb * 2 + qb) * 4 + r) * 64 + lane] = o[fb][qb][r];
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float M = -INFINITY, L = 0.f, f[NW];
#pragma unroll
        for (int x = 0; x < NW; ++x) M = fmaxf(M, mlb[(0 * NW + x) * 32 + 16 * qb + n]);
#pragma unroll
        for (int x = 0; x < NW; ++x) {
            f[x] = __expf(mlb[(0 * NW + x) * 32 + 16 * qb + n] - M);       // a wave without keys: m = -inf, factor 0
            L = fmaf(mlb[(1 * NW + x) * 32 + 16 * qb + n], f[x], L);
        }
        const int q = q0 + 16 * qb + n;
        const float inv = 1.f / L;
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int x = 0; x < NW; ++x) acc = fmaf(ob[(x * FB * 8 + (fb * 2 + qb) * 4 + r) * 64 + lane], f[x], acc);
                v[r] = acc * inv;
            }
            if (q < a.Nq && 16 * fb + 4 * g < DK)
                *reinterpret_cast<float4*>(a.O + (size_t)q * a.dim + h * DK + 16 * fb + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (q < a.Nq && g == 0) *reinterpret_cast<float2*>(a.LSE + 2 * ((size_t)q * a.heads + h)) = make_float2(M, inv);
    }
}


// ---------------------------------------------------------------------------------------------------------
// backward, exact fp32 matrix cores, second form (round 3): the two passes above on v_mfma_f32_16x16x4_f32 with the streamed
// side staged through LDS, as in mha_fwd16_kernel.
//   pass Q   workgroup = NW waves per (head, 32-query block); a stage = 32 * NW keys: K and V transposed (A operands of the
//            S^T = K Q^T and dP^T = V dO^T blocks, conflict-free ds_read_b32) and K row-major (A operand of dQ^T += K^T dS^T, whose
//            B operand IS register r of the dS^T block: no LDS round trip).  A lane owns two queries (n, 16 + n): max, 1 / sum
//            and delta are lane scalars.  No running maximum in a backward pass: nothing is rescaled.
//   pass KV  workgroup per (head, 32-key block); a stage = 32 * NW queries: Q and dO transposed (A operands of S = Q K^T and
//            dP = dO V^T) and row-major (A operands of dV^T += dO^T Pd, dK^T += Q^T dS), plus the queries' (max, 1 / sum, delta,
//            mask word of this key block).  A lane owns two keys; the rows of a block are queries 4 g + r.
// Waves add their partial sums through LDS in wave order (deterministic); no atomics.  Same semantics as the 32x32 form.
// ---------------------------------------------------------------------------------------------------------
template <int DK, int NW, bool DROP, bool MASK>
__global__ __launch_bounds__(NW * 64) void mha_bwd16_q_kernel(const MhaArgs a) {
    constexpr int SK = 32 * NW, NT = 64 * NW, FB = (DK + 15) / 16, KQ = DK / 4, PC = DK / 8;
    constexpr int BUF = 3 * DK * SK;            // floats per stage buffer: Kt [DK][SK] | Vt [DK][SK] | Ks [SK][DK]
    extern __shared__ __attribute__((aligned(16))) float mha_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int qblocks = (a.Nq + 31) / 32;
    const int h = blockIdx.x / qblocks, q0 = (blockIdx.x % qblocks) * 32;
    const uint32_t salt_h = DROP ? salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 1u + h) : 0u;
    const int mwords = (a.Nk + 31) >> 5;
    int qi[2];
    float qreg[2][KQ], doreg[2][KQ], mx[2], il[2], dl[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qi[qb] = min(q0 + 16 * qb + n, a.Nq - 1);
        const size_t row = (size_t)qi[qb] * a.dim + h * DK;
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < KQ; ++t) {
            qreg[qb][t] = a.Q[row + 4 * t + g];
            doreg[qb][t] = a.dO[row + 4 * t + g];
            part = fmaf(doreg[qb][t], a.O[row + 4 * t + g], part);
        }
        part += lane_xor16(part, g);
        part += lane_xor32(part, g);
        dl[qb] = part;                                                            // delta = dO . O over the head's features
        const float2 ml = *reinterpret_cast<const float2*>(a.LSE + 2 * ((size_t)qi[qb] * a.heads + h));   // (max, 1 / sum)
        mx[qb] = ml.x; il[qb] = ml.y;
    }
    mha_f32x4 dq[FB][2];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) dq[fb][qb] = mha_f32x4{0.f, 0.f, 0.f, 0.f};

    float4 kst[PC], vst[PC];
    auto fetch = [&](int s) {
#pragma unroll
        for (int i = 0; i < PC; ++i) {
            const int p = tid + i * NT, key = p / (DK / 4), c = p % (DK / 4);
            const size_t row = (size_t)min(s * SK + key, a.Nk - 1) * a.dim + h * DK + 4 * c;
            kst[i] = *reinterpret_cast<const float4*>(a.K + row);
            vst[i] = *reinterpret_cast<const float4*>(a.V + row);
        }
    };
    auto park = [&](int buf) {
        float* const B = mha_lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < PC; ++i) {
            const int p = tid + i * NT, key = p / (DK / 4), c = p % (DK / 4);
            float* kt = B + (4 * c) * SK + key;
            kt[0] = kst[i].x; kt[SK] = kst[i].y; kt[2 * SK] = kst[i].z; kt[3 * SK] = kst[i].w;
            float* vt = B + DK * SK + (4 * c) * SK + key;
            vt[0] = vst[i].x; vt[SK] = vst[i].y; vt[2 * SK] = vst[i].z; vt[3 * SK] = vst[i].w;
            *reinterpret_cast<float4*>(B + 2 * DK * SK + key * DK + 4 * c) = kst[i];
        }
    };
    const int nstage = (a.Nk + SK - 1) / SK;
    fetch(0);
    park(0);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) fetch(s + 1);
        const int k0 = s * SK + 32 * wv;
        if (k0 < a.Nk) {
            const float* kt = mha_lds + buf * BUF + 32 * wv;
            const float* vt = kt + DK * SK;
            const float* ks = mha_lds + buf * BUF + 2 * DK * SK + 32 * wv * DK;
            mha_f32x4 sc[2][2], dp[2][2];       // [kb][qb]: keys 16 kb + 4 g + r, query 16 qb + n
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float ka[KQ], va[KQ];
#pragma unroll
                for (int t = 0; t < KQ; ++t) {
                    ka[t] = kt[(4 * t + g) * SK + 16 * kb + n];
                    va[t] = vt[(4 * t + g) * SK + 16 * kb + n];
                }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    mha_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < KQ; ++t) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[t], qreg[qb][t], acc, 0, 0, 0);
                        acd = __builtin_amdgcn_mfma_f32_16x16x4f32(va[t], doreg[qb][t], acd, 0, 0, 0);
                    }
                    sc[kb][qb] = acc;
                    dp[kb][qb] = acd;
                }
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                uint32_t mw = 0xffffffffu;
                if (MASK) mw = a.mask[(size_t)qi[qb] * mwords + (k0 >> 5)];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kk = 16 * kb + 4 * g + r, key = k0 + kk;
                        const bool inr = key < a.Nk;
                        const bool open = (mw >> kk) & 1u;
                        const float sv = open ? sc[kb][qb][r] * a.scale : MHA_MASKED;
                        const float p = inr ? __expf(sv - mx[qb]) * il[qb] : 0.f;
                        float dpe = dp[kb][qb][r];
                        if (DROP) dpe *= drop_mul(a, salt_h, qi[qb], min(key, a.Nk - 1));
                        sc[kb][qb][r] = (open && inr) ? p * (dpe - dl[qb]) * a.scale : 0.f;
                    }
            }
            // dQ^T += K^T . dS^T: register r of block kb contracts over the keys 16 kb + 4 k + r, k = 0..3
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int fb = 0; fb < FB; ++fb) {
                        const float kav = (16 * fb + n < DK) ? ks[(16 * kb + 4 * g + r) * DK + 16 * fb + n] : 0.f;
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb)
                            dq[fb][qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kav, sc[kb][qb][r], dq[fb][qb], 0, 0, 0);
                    }
        }
        if (s + 1 < nstage) park(buf ^ 1);
        __syncthreads();
    }
    float* const ob = mha_lds;                                    // [NW][FB * 2 * 4][64]
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[(wv * FB * 8 + (fb * 2 + qb) * 4 + r) * 64 + lane] = dq[fb][qb][r];
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + 16 * qb + n;
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int x = 0; x < NW; ++x) acc += ob[(x * FB * 8 + (fb * 2 + qb) * 4 + r) * 64 + lane];
                v[r] = acc;
            }
            if (q < a.Nq && 16 * fb + 4 * g < DK)
                *reinterpret_cast<float4*>(a.dQ + (size_t)q * a.dim + h * DK + 16 * fb + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (q < a.Nq && g == 0) a.delta[(size_t)q * a.heads + h] = dl[qb];
    }
}

template <int DK, int NW, bool DROP, bool MASK>
__global__ __launch_bounds__(NW * 64) void mha_bwd16_kv_kernel(const MhaArgs a) {
    constexpr int SQ = 32 * NW, NT = 64 * NW, FB = (DK + 15) / 16, KQ = DK / 4, PC = DK / 8;
    constexpr int BUF = 4 * DK * SQ + 4 * SQ;   // floats per stage buffer: Qt | dOt [DK][SQ] | Qs | dOs [SQ][DK] | max | 1/sum | delta | mask word
    extern __shared__ __attribute__((aligned(16))) float mha_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int kblocks = (a.Nk + 31) / 32;
    const int h = blockIdx.x / kblocks, k0 = (blockIdx.x % kblocks) * 32;
    const uint32_t salt_h = DROP ? salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 1u + h) : 0u;
    const int mwords = (a.Nk + 31) >> 5;
    int ki[2];
    float kreg[2][KQ], vreg[2][KQ];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        ki[kb] = min(k0 + 16 * kb + n, a.Nk - 1);
        const size_t row = (size_t)ki[kb] * a.dim + h * DK;
#pragma unroll
        for (int t = 0; t < KQ; ++t) {
            kreg[kb][t] = a.K[row + 4 * t + g];
            vreg[kb][t] = a.V[row + 4 * t + g];
        }
    }
    mha_f32x4 dk[FB][2], dv[FB][2];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { dk[fb][kb] = mha_f32x4{0.f, 0.f, 0.f, 0.f}; dv[fb][kb] = mha_f32x4{0.f, 0.f, 0.f, 0.f}; }

    float4 qst[PC], dst[PC];
    float s_m = 0.f, s_i = 0.f, s_d = 0.f;
    uint32_t s_w = 0xffffffffu;
    auto fetch = [&](int s) {
#pragma unroll
        for (int i = 0; i < PC; ++i) {
            const int p = tid + i * NT, qq = p / (DK / 4), c = p % (DK / 4);
            const size_t row = (size_t)min(s * SQ + qq, a.Nq - 1) * a.dim + h * DK + 4 * c;
            qst[i] = *reinterpret_cast<const float4*>(a.Q + row);
            dst[i] = *reinterpret_cast<const float4*>(a.dO + row);
        }
        if (tid < SQ) {
            const int q = min(s * SQ + tid, a.Nq - 1);
            const float2 ml = *reinterpret_cast<const float2*>(a.LSE + 2 * ((size_t)q * a.heads + h));
            s_m = ml.x; s_i = ml.y;
            s_d = a.delta[(size_t)q * a.heads + h];
            if (MASK) s_w = a.mask[(size_t)q * mwords + (k0 >> 5)];
        }
    };
    auto park = [&](int buf) {
        float* const B = mha_lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < PC; ++i) {
            const int p = tid + i * NT, qq = p / (DK / 4), c = p % (DK / 4);
            float* qt = B + (4 * c) * SQ + qq;
            qt[0] = qst[i].x; qt[SQ] = qst[i].y; qt[2 * SQ] = qst[i].z; qt[3 * SQ] = qst[i].w;
            float* dt = B + DK * SQ + (4 * c) * SQ + qq;
            dt[0] = dst[i].x; dt[SQ] = dst[i].y; dt[2 * SQ] = dst[i].z; dt[3 * SQ] = dst[i].w;
            *reinterpret_cast<float4*>(B + 2 * DK * SQ + qq * DK + 4 * c) = qst[i];
            *reinterpret_cast<float4*>(B + 3 * DK * SQ + qq * DK + 4 * c) = dst[i];
        }
        if (tid < SQ) {
            float* sc4 = B + 4 * DK * SQ;
            sc4[tid] = s_m; sc4[SQ + tid] = s_i; sc4[2 * SQ + tid] = s_d; sc4[3 * SQ + tid] = __uint_as_float(s_w);
        }
    };
    const int nstage = (a.Nq + SQ - 1) / SQ;
    fetch(0);
    park(0);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) fetch(s + 1);
        const int qw = 32 * wv, q0 = s * SQ + qw;                  // this wave's 32 queries (index in the stage, global)
        if (q0 < a.Nq) {
            const float* B = mha_lds + buf * BUF;
            const float* qt = B + qw;
            const float* dt = B + DK * SQ + qw;
            const float* qs = B + 2 * DK * SQ + qw * DK;
            const float* ds_ = B + 3 * DK * SQ + qw * DK;
            const float* sc4 = B + 4 * DK * SQ + qw;
            mha_f32x4 sc[2][2], dp[2][2];       // [qb][kb]: queries 16 qb + 4 g + r, key 16 kb + n
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float qa[KQ], da[KQ];
#pragma unroll
                for (int t = 0; t < KQ; ++t) {
                    qa[t] = qt[(4 * t + g) * SQ + 16 * qb + n];
                    da[t] = dt[(4 * t + g) * SQ + 16 * qb + n];
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    mha_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < KQ; ++t) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[t], kreg[kb][t], acc, 0, 0, 0);
                        acd = __builtin_amdgcn_mfma_f32_16x16x4f32(da[t], vreg[kb][t], acd, 0, 0, 0);
                    }
                    sc[qb][kb] = acc;
                    dp[qb][kb] = acd;
                }
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = 16 * qb + 4 * g + r, q = q0 + qq;
                    const bool inq = q < a.Nq;
                    const int qc = inq ? q : a.Nq - 1;
                    const float M = sc4[qq], inv = sc4[SQ + qq], dlt = sc4[2 * SQ + qq];
                    const uint32_t mw = __float_as_uint(sc4[3 * SQ + qq]);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const int kk = 16 * kb + n;
                        const bool inr = inq && k0 + kk < a.Nk;
                        const bool open = !MASK || ((mw >> kk) & 1u);
                        const float sv = open ? sc[qb][kb][r] * a.scale : MHA_MASKED;
                        const float p = inr ? __expf(sv - M) * inv : 0.f;
                        const float mul = DROP ? drop_mul(a, salt_h, qc, ki[kb]) : 1.f;
                        sc[qb][kb][r] = (open && inr) ? p * (dp[qb][kb][r] * mul - dlt) * a.scale : 0.f;
                        dp[qb][kb][r] = p * mul;
                    }
                }
            // dV^T += dO^T . Pd, dK^T += Q^T . dS: register r of block qb contracts over the queries 16 qb + 4 k + r, k = 0..3
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int fb = 0; fb < FB; ++fb) {
                        const bool live = 16 * fb + n < DK;
                        const float dov = live ? ds_[(16 * qb + 4 * g + r) * DK + 16 * fb + n] : 0.f;
                        const float qv = live ? qs[(16 * qb + 4 * g + r) * DK + 16 * fb + n] : 0.f;
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            dv[fb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(dov, dp[qb][kb][r], dv[fb][kb], 0, 0, 0);
                            dk[fb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv, sc[qb][kb][r], dk[fb][kb], 0, 0, 0);
                        }
                    }
        }
        if (s + 1 < nstage) park(buf ^ 1);
        __syncthreads();
    }
    float* const ob = mha_lds;                                    // [2][NW][FB * 2 * 4][64]
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ob[(wv * FB * 8 + (fb * 2 + kb) * 4 + r) * 64 + lane] = dk[fb][kb][r];
                ob[((NW + wv) * FB * 8 + (fb * 2 + kb) * 4 + r) * 64 + lane] = dv[fb][kb][r];
            }
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = k0 + 16 * kb + n;
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
            float vk[4], vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float ak = 0.f, av = 0.f;
#pragma unroll
                for (int x = 0; x < NW; ++x) {
                    ak += ob[(x * FB * 8 + (fb * 2 + kb) * 4 + r) * 64 + lane];
                    av += ob[((NW + x) * FB * 8 + (fb * 2 + kb) * 4 + r) * 64 + lane];
                }
                vk[r] = ak; vv[r] = av;
            }
            if (key < a.Nk && 16 * fb + 4 * g < DK) {
                *reinterpret_cast<float4*>(a.dK + (size_t)key * a.dim + h * DK + 16 * fb + 4 * g) = make_float4(vk[0], vk[1], vk[2], vk[3]);
                *reinterpret_cast<float4*>(a.dV + (size_t)key * a.dim + h * DK + 16 * fb + 4 * g) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            }
        }
    }
}

template <int DK>
static int launch_mha(const MhaArgs& a, bool backward, bool drop, hipStream_t st) {
    const int qw = ((a.Nq + 31) / 32) * a.heads, kw = ((a.Nk + 31) / 32) * a.heads;
    const dim3 gq(qw), gk(kw);                                       // one workgroup per 32-row block and head
    constexpr int NW = DK == 32 ? 4 : 8;
    const bool mk = a.mask != nullptr;
    if (!backward) {                                                 // exact fp32: the 16x16x4 form with LDS-staged K / V
        constexpr size_t lds = (size_t)4 * DK * 32 * NW * sizeof(float);
#define NGACF_MHA16(DR, MK) hipLaunchKernelGGL((mha_fwd16_kernel<DK, NW, DR, MK>), gq, dim3(64 * NW), lds, st, a)
        if (drop) { if (mk) NGACF_MHA16(true, true); else NGACF_MHA16(true, false); }
        else      { if (mk) NGACF_MHA16(false, true); else NGACF_MHA16(false, false); }
#undef NGACF_MHA16
        NGACF_CHECK_LAUNCH();
        return 0;
    }
    constexpr size_t lq = (size_t)2 * 3 * DK * 32 * NW * sizeof(float);
    constexpr size_t lkv = (size_t)2 * (4 * DK * 32 * NW + 4 * 32 * NW) * sizeof(float);
#define NGACF_MHAB(KERNEL, GRID, LDS, DR, MK)                                                                             \
    do {                                                                                                              \
        static bool big_lds_ok = false;   /* > 64 KB of dynamic LDS must be asked for once per kernel */              \
        if ((LDS) > 65536 && !big_lds_ok) {                                                                           \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL<DK, NW, DR, MK>),          \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS));         \
            if (e != hipSuccess) return static_cast<int>(e);                                                          \
            big_lds_ok = true;                                                                                        \
        }                                                                                                             \
        hipLaunchKernelGGL((KERNEL<DK, NW, DR, MK>), GRID, dim3(64 * NW), LDS, st, a);                                \
    } while (0)
    if (drop) { if (mk) NGACF_MHAB(mha_bwd16_q_kernel, gq, lq, true, true); else NGACF_MHAB(mha_bwd16_q_kernel, gq, lq, true, false); }
    else      { if (mk) NGACF_MHAB(mha_bwd16_q_kernel, gq, lq, false, true); else NGACF_MHAB(mha_bwd16_q_kernel, gq, lq, false, false); }
    NGACF_CHECK_LAUNCH();
    if (drop) { if (mk) NGACF_MHAB(mha_bwd16_kv_kernel, gk, lkv, true, true); else NGACF_MHAB(mha_bwd16_kv_kernel, gk, lkv, true, false); }
    else      { if (mk) NGACF_MHAB(mha_bwd16_kv_kernel, gk, lkv, false, true); else NGACF_MHAB(mha_bwd16_kv_kernel, gk, lkv, false, false); }
#undef NGACF_MHAB
    NGACF_CHECK_LAUNCH();
    return 0;
}

}  // namespace ngacf

using namespace ngacf;

static int mha_dispatch(MhaArgs& a, bool backward, float drop_p, void* stream) {
    if (a.Nq <= 0 || a.Nk <= 0 || a.heads <= 0 || a.dim <= 0 || a.dim % a.heads != 0) return NGACF_E_ARG;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return NGACF_E_ARG;
    const int dk = a.dim / a.heads;
    a.scale = 1.f / sqrtf((float)dk);                              // modules.py:33
    a.drop_thr = drop_threshold(drop_p);
    a.drop_scale = 1.f / (1.f - drop_p);
    const bool drop = drop_p > 0.f;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NGACF_CASE(DKV)                                                                                  \
    if (dk == DKV) return launch_mha<DKV>(a, backward, drop, st)
    NGACF_CASE(8);
    NGACF_CASE(16);
    NGACF_CASE(32);
#undef NGACF_CASE
    return NGACF_E_SHAPE;
}

extern "C" int ngacf_mha_fwd(const float* Q, const float* K, const float* V, const uint32_t* mask, float* O, float* LSE,
                             int Nq, int Nk, int dim, int heads, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                             void* stream) {
    if (!Q || !K || !V || !O || !LSE) return NGACF_E_ARG;
    MhaArgs a{};
    a.Q = Q; a.K = K; a.V = V; a.mask = mask; a.O = O; a.LSE = LSE; a.Nq = Nq; a.Nk = Nk; a.dim = dim; a.heads = heads;
    a.seed = seed; a.seed_dev = seed_dev;
    return mha_dispatch(a, false, drop_p, stream);
}

extern "C" int ngacf_mha_bwd(const float* Q, const float* K, const float* V, const uint32_t* mask, const float* O,
                             const float* LSE, const float* dO, float* dQ, float* dK, float* dV, float* delta,
                             int Nq, int Nk, int dim, int heads, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                             void* stream) {
    if (!Q || !K || !V || !O || !LSE || !dO || !dQ || !dK || !dV || !delta) return NGACF_E_ARG;
    MhaArgs a{};
    a.Q = Q; a.K = K; a.V = V; a.mask = mask; a.O = const_cast<float*>(O); a.LSE = const_cast<float*>(LSE); a.dO = dO;
    a.dQ = dQ; a.dK = dK; a.dV = dV; a.delta = delta; a.Nq = Nq; a.Nk = Nk; a.dim = dim; a.heads = heads;
    a.seed = seed; a.seed_dev = seed_dev;
    return mha_dispatch(a, true, drop_p, stream);
}
