// bf16-storage form of the row transform and of its dX backward on the matrix cores.
//
//   forward  H[n, :] = X[n, :] @ W[type]          (contraction over din)      + scores S = H . a
//   dX       dX[n, :] = dH[n, :] @ W[type]^T      (contraction over dout)
//
// Both are D^T = A . B^T with A taken from the 64..128-wide weight panel and B from node rows, so one
// kernel serves both: v_mfma_f32_32x32x16_bf16 with the NODE on the lane (B operand) and the output
// column on the accumulator registers.  A lane then owns 4 consecutive output columns of its node per
// register group -> 8-byte (bf16) / 16-byte (fp32) row-contiguous stores, and the B fragment of a node is
// 16 contiguous bytes of its row: no LDS transpose anywhere.  The weight panel is converted to bf16
// A-fragments once per workgroup (LDS, 8..32 KB).  fp32 accumulate; operands rounded to bf16 (this path
// runs only when H is stored as bf16 -- the fp32 path keeps the exact FMA kernels of transform.hip).
#include "ngacf_common.hpp"
#include "mfma_rows.hpp"

namespace ngacf {

constexpr int MF_MAX_BLOCKS = 4096;  // ~16 workgroups per CU: beyond that a wave takes several 32-row tiles, so the
                                    // per-workgroup weight-panel prologue stays amortised on 10^7-row tables

struct MfmaArgs {
    const void* B;                 // node rows: X (forward) or dH (dX)
    const float* M;                // W [2][DIN][DOUT]
    const float* avec;             // [2][DOUT]  (forward)
    void* out;                     // H (forward, global rows) or dX
    float* S;                      // [N][heads] (forward)
    int U, I;
    int u_lo, u_n, i_lo, i_n;      // row windows
    int b_local;                   // B holds only the window rows (forward: X local)
    int out_local;                 // out holds only the window rows (dX local)
    uint32_t drop_thr;
    float drop_scale;
    uint64_t seed;
    const uint64_t* seed_dev;
    int tpw;                       // 32-row tiles per wave
    // side jobs (forward only): workgroups nb_main .. nb_main + 2*npack - 1 pack LATER stages' weight panels into the bf16
    // A fragments of v_mfma_f32_16x16x32_bf16 that the fused traversal epilogue reads (propagate.hip, NextArgs)
    int nb_main, npack;
    const float* pack_W[NGACF_MAX_PACK];
    void* pack_F[NGACF_MAX_PACK];
    int pack_din[NGACF_MAX_PACK], pack_dout[NGACF_MAX_PACK], pack_kind[NGACF_MAX_PACK];   // kind 1: the transposed (dX) fragments
};

__device__ __forceinline__ bf16x8 pack8(const float (&f)[8]) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = static_cast<__bf16>(f[j]);
    return r;
}

// TB: type of the node-row operand (bf16_t or float); TO: output element type; FWD: forward (else dX)
template <typename TB, typename TO, int DIN, int DOUT, bool FWD, int HEADS, bool DROP, bool STAGE_OUT = false>
__global__ __launch_bounds__(256) void transform_mfma_kernel(const MfmaArgs a) {
    constexpr int KD = FWD ? DIN : DOUT;           // contraction width
    constexpr int OD = FWD ? DOUT : DIN;           // output width
    constexpr int KS = KD / 16, OT = OD / 32;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    bf16x8* Af = reinterpret_cast<bf16x8*>(lds_raw);           // [KS][OT][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (FWD) {
        if ((int)blockIdx.x >= a.nb_main) {                  // a side job: one (stage, node type) weight panel
            const int job = blockIdx.x - a.nb_main;
            if (a.pack_kind[job >> 1]) pack_panel_job_t(a.pack_W[job >> 1], a.pack_F[job >> 1], a.pack_din[job >> 1], a.pack_dout[job >> 1], job & 1, tid);
            else pack_panel_job(a.pack_W[job >> 1], a.pack_F[job >> 1], a.pack_din[job >> 1], a.pack_dout[job >> 1], job & 1, tid);
            return;
        }
    }
    const int tiles_u = (a.u_n + 31) / 32, tiles_i = (a.i_n + 31) / 32;
    const int nbu = (tiles_u + 4 * a.tpw - 1) / (4 * a.tpw);
    const bool users = (int)blockIdx.x < nbu;
    const int type = users ? 0 : 1;
    const int bl = users ? blockIdx.x : blockIdx.x - nbu;
    const int win_lo = users ? a.u_lo : a.U + a.i_lo;
    const int win_end = win_lo + (users ? a.u_n : a.i_n);
    const int ntiles = users ? tiles_u : tiles_i;
    const int loc_off = users ? a.u_lo : a.U + a.i_lo - a.u_n;  // global row - loc_off = local row
    const uint32_t salt = DROP ? salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u) : 0u;

    const int r = lane & 31, h = lane >> 5;
    // Node rows of one 32-row tile -> bf16 B fragments (16 contiguous bytes of the node's row per k-step), THROUGH LDS: the tile
    // is 32 consecutive rows = one contiguous span of 32 * RB bytes, fetched as RB / 32 fully coalesced 1 KB wave loads (every
    // cache line requested once), converted (+ feature dropout) and laid down as bf16 rows in the wave's own LDS tile; the
    // fragment of lane (r, h) for k-step s is then one 16-byte LDS read.  (Fetching the fragments straight from global memory
    // made every wave instruction touch 32 lines for 1 KB -- each line requested 4-8 times: the launch ran at a third of the
    // copy rate.)  The tile is wave-private: LDS executes a wave's instructions in order, no barrier.
    constexpr int RB = KD * (int)sizeof(TB);                 // bytes of a node row: 128, 256 or 512
    constexpr int NLD = RB / 32;                             // 16-byte loads per lane and tile
    constexpr int TROW = KD / 8 + 1;                         // LDS tile row in uint4: the bf16 row + 16 B (fragment reads: no bank conflict)
    constexpr int TSZ = 32 * ((KD > OD ? KD : OD) / 8 + 1);  // (the forward's epilogue sends H' through the same tile)
    uint4* tile = reinterpret_cast<uint4*>(lds_raw + (size_t)KS * OT * 64 * 16) + wave * TSZ;
    float* const avs = reinterpret_cast<float*>(reinterpret_cast<uint4*>(lds_raw + (size_t)KS * OT * 64 * 16) + 4 * TSZ);   // the type's attention vector (forward)
    const char* Bg = static_cast<const char*>(a.B);
    uint4 raw[NLD];
    int base = 0, rows = 0;                                  // the tile in `raw`: first node, number of valid rows
    auto issue = [&](int it) {
        const int ti = (bl * 4 + wave) * a.tpw + it;
        base = win_lo + ti * 32;
        rows = ti < ntiles ? min(32, win_end - base) : 0;
        const char* src = Bg + (size_t)(a.b_local ? base - loc_off : base) * RB;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int o = j * 1024 + lane * 16;
            raw[j] = o < rows * RB ? *reinterpret_cast<const uint4*>(src + o) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stage = [&]() {                                     // raw -> the LDS tile
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int o = j * 1024 + lane * 16;
            const int tr = o / RB, cb = o % RB;              // tile row, byte offset inside the node's row
            if constexpr (sizeof(TB) == 2) {
                uint4 v = raw[j];
                if constexpr (FWD && DROP) {
                    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = __uint_as_float(w[k] << 16), hi = __uint_as_float(w[k] & 0xffff0000u);
                        const int c = cb / 2 + 2 * k;
                        const float l2 = feat_bits(salt, base + tr, c) >= a.drop_thr ? lo * a.drop_scale : 0.f;
                        const float h2 = feat_bits(salt, base + tr, c + 1) >= a.drop_thr ? hi * a.drop_scale : 0.f;
                        w[k] = f32_to_bf16(l2) | (f32_to_bf16(h2) << 16);
                    }
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
                tile[tr * TROW + cb / 16] = v;
            } else {
                float f[4] = {__uint_as_float(raw[j].x), __uint_as_float(raw[j].y), __uint_as_float(raw[j].z), __uint_as_float(raw[j].w)};
                if constexpr (FWD && DROP) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) f[k] = feat_bits(salt, base + tr, cb / 4 + k) >= a.drop_thr ? f[k] * a.drop_scale : 0.f;
                }
                reinterpret_cast<uint2*>(tile)[(tr * TROW * 16 + cb / 2) / 8] =
                    make_uint2(f32_to_bf16(f[0]) | (f32_to_bf16(f[1]) << 16), f32_to_bf16(f[2]) | (f32_to_bf16(f[3]) << 16));
            }
        }
    };
    issue(0);                            // in flight while the workgroup builds the weight panel

    {   // weight panel -> bf16 A fragments: A[i][k] = W[k][i] (forward) or W[i][k] (dX)
        const float* M = a.M + (size_t)type * DIN * DOUT;
        for (int slot = tid; slot < KS * OT * 64; slot += 256) {
            const int l = slot & 63, t = (slot >> 6) % OT, s = (slot >> 6) / OT;
            const int i = 32 * t + (l & 31), k0 = 16 * s + 8 * (l >> 5);
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = FWD ? M[(size_t)(k0 + j) * DOUT + i] : M[(size_t)i * DOUT + k0 + j];
            Af[slot] = pack8(f);
        }
        if constexpr (FWD) {
            if (tid < DOUT / 4) reinterpret_cast<float4*>(avs)[tid] = reinterpret_cast<const float4*>(a.avec + type * DOUT)[tid];
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int it = 0; it < a.tpw; ++it) {
        if ((bl * 4 + wave) * a.tpw + it >= ntiles) break;
        stage();
        const bool ok_cur = r < rows;
        const int nd_cur = ok_cur ? base + r : win_end - 1;
        bf16x8 bf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) bf[s] = __builtin_bit_cast(bf16x8, tile[r * TROW + 2 * s + h]);
        const int base_done = base, rows_done = rows;
        if (it + 1 < a.tpw) issue(it + 1);                   // next tile's rows fly under this tile's MFMAs and epilogue
        f32x16 acc[OT];
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < OT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[(s * OT + t) * 64 + lane], bf[s], acc[t], 0, 0, 0);

        // lane: node r (+ half h); acc[t][4g + j] = output column 32t + 8g + 4h + j
        if constexpr (FWD) {
            const int base_cur = base_done, rows_cur = rows_done;
            mfma_fwd_epilogue<OT, HEADS, DOUT, STAGE_OUT>(acc, nd_cur, ok_cur, h, avs, a.out, a.S, tile, r,
                                               [&](int tr) { return tr < rows_cur ? base_cur + tr : -1; });
        } else {
            const size_t orow = (size_t)(a.out_local ? nd_cur - loc_off : nd_cur);
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    float v[2][4];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int c = 32 * t + 8 * (gp + gg) + 4 * h;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[gg][j] = acc[t][4 * (gp + gg) + j];
                            if (DROP) v[gg][j] = feat_bits(salt, nd_cur, c + j) >= a.drop_thr ? v[gg][j] * a.drop_scale : 0.f;
                        }
                    }
                    if constexpr (sizeof(TO) == 4) {
                        if (!ok_cur) continue;
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg)
                            *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + orow * DIN + 32 * t + 8 * (gp + gg) + 4 * h) =
                                make_float4(v[gg][0], v[gg][1], v[gg][2], v[gg][3]);
                    } else {    // bf16 rows: the half-waves exchange one group each -> one 16-byte store per lane (see the forward)
                        uint2 pk[2];
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg)
                            pk[gg] = make_uint2(f32_to_bf16(v[gg][0]) | (f32_to_bf16(v[gg][1]) << 16), f32_to_bf16(v[gg][2]) | (f32_to_bf16(v[gg][3]) << 16));
                        const auto rx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                        const auto ry = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                        if (ok_cur)
                            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.out) + orow * DIN * 2 + 64 * t + 16 * gp + 16 * h) =
                                make_uint4(rx[0], ry[0], rx[1], ry[1]);
                    }
                }
        }
    }
}

template <typename TB, typename TO, int DIN, int DOUT, bool FWD, int HEADS>
static int launch_mfma(MfmaArgs a, bool drop, hipStream_t st) {
    constexpr int KD = FWD ? DIN : DOUT, OD = FWD ? DOUT : DIN;
    const int tiles_u = (a.u_n + 31) / 32, tiles_i = (a.i_n + 31) / 32;
    a.tpw = (tiles_u + tiles_i + 4 * MF_MAX_BLOCKS - 1) / (4 * MF_MAX_BLOCKS);
    if (a.tpw < 1) a.tpw = 1;
    a.nb_main = (tiles_u + 4 * a.tpw - 1) / (4 * a.tpw) + (tiles_i + 4 * a.tpw - 1) / (4 * a.tpw);
    if (!FWD) a.npack = 0;
    const bool stage_out = FWD && stage_out_pays((uint64_t)a.U + a.I, DOUT);    // H' through the LDS tile as whole lines
    const int nb = a.nb_main + 2 * a.npack;
    if (nb == 0) return 0;
    constexpr size_t lds = (size_t)(KD / 16) * (OD / 32) * 64 * 16 + (size_t)4 * 32 * ((KD > OD ? KD : OD) / 8 + 1) * 16 + (size_t)OD * 4;   // panel + the four waves' row tiles + the attention vector
    if (!FWD) {
        if (drop) hipLaunchKernelGGL((transform_mfma_kernel<TB, TO, DIN, DOUT, FWD, HEADS, true>), dim3(nb), dim3(256), lds, st, a);
        else      hipLaunchKernelGGL((transform_mfma_kernel<TB, TO, DIN, DOUT, FWD, HEADS, false>), dim3(nb), dim3(256), lds, st, a);
    } else if (stage_out) {
        if (drop) launch_k(NGACF_TAG_TRANSFORM_FWD, transform_mfma_kernel<TB, TO, DIN, DOUT, FWD, HEADS, true, FWD>, dim3(nb), dim3(256), lds, st, a);
        else      launch_k(NGACF_TAG_TRANSFORM_FWD, transform_mfma_kernel<TB, TO, DIN, DOUT, FWD, HEADS, false, FWD>, dim3(nb), dim3(256), lds, st, a);
    } else {
        if (drop) launch_k(NGACF_TAG_TRANSFORM_FWD, transform_mfma_kernel<TB, TO, DIN, DOUT, FWD, HEADS, true>, dim3(nb), dim3(256), lds, st, a);
        else      launch_k(NGACF_TAG_TRANSFORM_FWD, transform_mfma_kernel<TB, TO, DIN, DOUT, FWD, HEADS, false>, dim3(nb), dim3(256), lds, st, a);
    }
    NGACF_CHECK_LAUNCH();
    return 0;
}

template <typename TB, typename TO, bool FWD>
static int dispatch_mfma(const MfmaArgs& a, int din, int dout, int heads, bool drop, hipStream_t st) {
    if (din == 64 && dout == 64 && heads == 8) return launch_mfma<TB, TO, 64, 64, FWD, 8>(a, drop, st);
    if (din == 64 && dout == 64 && heads == 1) return launch_mfma<TB, TO, 64, 64, FWD, 1>(a, drop, st);
    if (din == 128 && dout == 64 && heads == 8) return launch_mfma<TB, TO, 128, 64, FWD, 8>(a, drop, st);
    if (din == 128 && dout == 64 && heads == 1) return launch_mfma<TB, TO, 128, 64, FWD, 1>(a, drop, st);
    if (din == 64 && dout == 128 && heads == 1) return launch_mfma<TB, TO, 64, 128, FWD, 1>(a, drop, st);
    if (din == 64 && dout == 128 && heads == 16) return launch_mfma<TB, TO, 64, 128, FWD, 16>(a, drop, st);
    if (din == 128 && dout == 128 && heads == 1) return launch_mfma<TB, TO, 128, 128, FWD, 1>(a, drop, st);
    if (din == 128 && dout == 128 && heads == 16) return launch_mfma<TB, TO, 128, 128, FWD, 16>(a, drop, st);
    return NGACF_E_SHAPE;
}

// entry points used by transform.hip / transform_bwd.hip when H is stored as bf16
int transform_fwd_mfma(const void* X, int x_dtype, const float* W, const float* avec, void* H, float* S,
                       int U, int I, int din, int dout, int heads, int u_lo, int u_n, int i_lo, int i_n, int x_local,
                       float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t st,
                       int n_pack, const float* const* pack_W, void* const* pack_panel, const int* pack_din, const int* pack_dout,
                       const int* pack_kind) {
    MfmaArgs a;
    a.npack = n_pack;
    for (int k = 0; k < n_pack; ++k) {
        a.pack_W[k] = pack_W[k]; a.pack_F[k] = pack_panel[k]; a.pack_din[k] = pack_din[k]; a.pack_dout[k] = pack_dout[k];
        a.pack_kind[k] = pack_kind ? pack_kind[k] : 0;
    }
    a.B = X; a.M = W; a.avec = avec; a.out = H; a.S = S; a.U = U; a.I = I;
    a.u_lo = u_lo; a.u_n = u_n; a.i_lo = i_lo; a.i_n = i_n; a.b_local = x_local; a.out_local = 0;
    a.drop_thr = drop_threshold(feat_drop_p); a.drop_scale = 1.f / (1.f - feat_drop_p); a.seed = seed; a.seed_dev = seed_dev; a.tpw = 1;
    const bool drop = feat_drop_p > 0.f;
    if (x_dtype == NGACF_BF16) return dispatch_mfma<bf16_t, bf16_t, true>(a, din, dout, heads, drop, st);
    return dispatch_mfma<float, bf16_t, true>(a, din, dout, heads, drop, st);
}

int transform_dx_mfma(const void* dH, int dh_dtype, const float* W, void* dX, int dx_dtype, int U, int I, int din, int dout, int heads,
                      int u_lo, int u_n, int i_lo, int i_n, int x_local,
                      float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t st) {
    MfmaArgs a;
    a.npack = 0; a.nb_main = 0;
    a.B = dH; a.M = W; a.avec = nullptr; a.out = dX; a.S = nullptr; a.U = U; a.I = I;
    a.u_lo = u_lo; a.u_n = u_n; a.i_lo = i_lo; a.i_n = i_n; a.b_local = 0; a.out_local = x_local;
    a.drop_thr = drop_threshold(feat_drop_p); a.drop_scale = 1.f / (1.f - feat_drop_p); a.seed = seed; a.seed_dev = seed_dev; a.tpw = 1;
    const bool drop = feat_drop_p > 0.f;
    if (dh_dtype == NGACF_BF16) {
        if (dx_dtype == NGACF_BF16) return dispatch_mfma<bf16_t, bf16_t, false>(a, din, dout, heads, drop, st);
        return dispatch_mfma<bf16_t, float, false>(a, din, dout, heads, drop, st);
    }
    if (dx_dtype == NGACF_BF16) return dispatch_mfma<float, bf16_t, false>(a, din, dout, heads, drop, st);
    return dispatch_mfma<float, float, false>(a, din, dout, heads, drop, st);
}

}  // namespace ngacf

using namespace ngacf;

extern "C" int ngacf_transform_fwd_pack(const void* X, int x_dtype, const float* W, const float* avec, void* H, int h_dtype, float* S,
                                        int U, int I, int din, int dout, int heads,
                                        float feat_drop_p, uint64_t seed, const uint64_t* seed_dev,
                                        int n_pack, const float* const* pack_W, void* const* pack_panel, const int* pack_din,
                                        const int* pack_dout, const int* pack_kind, void* stream) {
    if (U < 0 || I < 0 || n_pack < 0 || n_pack > NGACF_MAX_PACK) return NGACF_E_ARG;
    if (U + I > 0 && (!X || !W || !avec || !H || !S)) return NGACF_E_ARG;     // (U = I = 0: the panels alone, nothing is transformed)
    if (U + I == 0) { din = 64; dout = 64; heads = 8; }
    if (!(feat_drop_p >= 0.f && feat_drop_p < 1.f)) return NGACF_E_ARG;
    if (h_dtype != NGACF_BF16 || (x_dtype != NGACF_F32 && x_dtype != NGACF_BF16)) return NGACF_E_SHAPE;   // the matrix-core path only
    if (n_pack > 0 && (!pack_W || !pack_panel || !pack_din || !pack_dout)) return NGACF_E_ARG;
    for (int k = 0; k < n_pack; ++k) {
        if (!pack_W[k] || !pack_panel[k]) return NGACF_E_ARG;
        if (pack_din[k] != 64 || (pack_dout[k] != 64 && pack_dout[k] != 128)) return NGACF_E_SHAPE;      // what ngacf_prop_fwd_next takes
        if (pack_kind && pack_kind[k] != 0 && (pack_kind[k] != 1 || pack_dout[k] != 64)) return NGACF_E_SHAPE;   // what ngacf_prop_bwd_dx takes
    }
    return transform_fwd_mfma(X, x_dtype, W, avec, H, S, U, I, din, dout, heads, 0, U, 0, I, 0, feat_drop_p, seed, seed_dev,
                              static_cast<hipStream_t>(stream), n_pack, pack_W, pack_panel, pack_din, pack_dout, pack_kind);
}
