// Row transform  H = dropout(X) @ W[type]  with the per-node attention scores
// S[n,h] = sum_{c in head h} H[n,c] * avec[type][c]  fused into the epilogue.
// Replaces graphattention/SPUIGACF.py:356-357 (+ the node halves of :361, the F.dropout of
// :208/:213) for all heads of a stage at once.
//
// [N,64]x[64,64] fp32: 16 flop per byte of X/H traffic -- HBM/issue balanced, not MFMA shaped
// (the fp32 MFMA runs at the vector rate on gfx950).  One workgroup per tile of 16*RPT rows with
// the 16-32 KB weight panel in LDS; RPT is picked per launch so that the grid fits one residency
// round (4 workgroups per CU) whenever the graph allows: a persistent 2-per-CU variant measured
// slower (too few loads in flight per CU), see DESIGN.md.
#include "ngacf_common.hpp"
#include "mfma_rows.hpp"

namespace ngacf {

struct TransArgs {
    const void* X;
    const float* W;
    const float* avec;
    void* H;
    float* S;
    int U, I;
    int u_lo, u_n, i_lo, i_n;      // row windows: users [u_lo, u_lo+u_n), items [i_lo, i_lo+i_n) (global row = U + i)
    int x_local;                   // X holds only the window rows, users first (node-partitioned runs)
    uint32_t drop_thr;
    float drop_scale;
    uint64_t seed;
    const uint64_t* seed_dev;      // optional device-side addend (hipGraph replays advance it)
};

// thread (ty, tx): rows RPT*ty..RPT*ty+RPT-1 of the tile, columns 4*tx..+3 (and 64+4*tx..+3 when DOUT = 128)
template <typename TX, typename TH, int DIN, int DOUT, int HEADS, bool DROP, int RPT>
__global__ __launch_bounds__(256) void transform_fwd_kernel(const TransArgs a) {
    constexpr int TM = 16 * RPT;                   // rows per workgroup tile
    constexpr int XS = DIN + 4;                    // LDS row stride of the X tile (floats), keeps 16-B alignment
    constexpr int CG = DOUT / 64;                  // column groups per thread
    constexpr int HW = DOUT / HEADS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                               // [TM][XS]
    float* Ws = lds + TM * XS;                     // [DIN][DOUT]

    const int tid = threadIdx.x;
    const uint32_t salt = DROP ? salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u) : 0u;
    const int nbu = (a.u_n + TM - 1) / TM;
    const bool users = (int)blockIdx.x < nbu;
    const int type = users ? 0 : 1;
    const int row0 = users ? a.u_lo + blockIdx.x * TM : a.U + a.i_lo + (blockIdx.x - nbu) * TM;
    const int row_end = users ? a.u_lo + a.u_n : a.U + a.i_lo + a.i_n;
    const int xoff = a.x_local ? (users ? a.u_lo : a.U + a.i_lo - a.u_n) : 0;     // X row = global row - xoff

    // weight panel -> LDS (coalesced float4)
    {
        const float4* Wg = reinterpret_cast<const float4*>(a.W + (size_t)type * DIN * DOUT);
        float4* Wl = reinterpret_cast<float4*>(Ws);
        for (int i = tid; i < DIN * DOUT / 4; i += 256) Wl[i] = Wg[i];
    }
    // X tile -> LDS, dropout applied on the way
    {
        constexpr int XV = Slice<TX>::VEC;
        constexpr int PER_ROW = DIN / XV;
        const TX* Xg = static_cast<const TX*>(a.X);
        for (int i = tid; i < TM * PER_ROW; i += 256) {
            const int r = i / PER_ROW, cv = i % PER_ROW;
            float f[XV];
            if (row0 + r < row_end) {
                Slice<TX>::load(Xg + (size_t)(row0 + r - xoff) * DIN + cv * XV, f);
                if (DROP) {
#pragma unroll
                    for (int j = 0; j < XV; ++j)
                        f[j] = feat_bits(salt, row0 + r, cv * XV + j) >= a.drop_thr ? f[j] * a.drop_scale : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < XV; ++j) f[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < XV; j += 4)
                *reinterpret_cast<float4*>(Xs + r * XS + cv * XV + j) = make_float4(f[j], f[j + 1], f[j + 2], f[j + 3]);
        }
    }
    __syncthreads();

    const int ty = tid >> 4, tx = tid & 15;
    float acc[RPT][4 * CG];
#pragma unroll
    for (int i = 0; i < RPT; ++i)
#pragma unroll
        for (int j = 0; j < 4 * CG; ++j) acc[i][j] = 0.f;

#pragma unroll 4                   // full unrolling measured 2x slower (registers / instruction cache)
    for (int k4 = 0; k4 < DIN / 4; ++k4) {
        float4 xa[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) xa[i] = *reinterpret_cast<const float4*>(Xs + (RPT * ty + i) * XS + 4 * k4);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                const float4 wb = *reinterpret_cast<const float4*>(Ws + (4 * k4 + kk) * DOUT + 64 * g + 4 * tx);
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const float x = kk == 0 ? xa[i].x : kk == 1 ? xa[i].y : kk == 2 ? xa[i].z : xa[i].w;
                    acc[i][4 * g + 0] = fmaf(x, wb.x, acc[i][4 * g + 0]);
                    acc[i][4 * g + 1] = fmaf(x, wb.y, acc[i][4 * g + 1]);
                    acc[i][4 * g + 2] = fmaf(x, wb.z, acc[i][4 * g + 2]);
                    acc[i][4 * g + 3] = fmaf(x, wb.w, acc[i][4 * g + 3]);
                }
            }
        }
    }

    // epilogue: round to the storage type, scores from the rounded values, store
    float av[4 * CG];
#pragma unroll
    for (int g = 0; g < CG; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) av[4 * g + j] = a.avec[type * DOUT + 64 * g + 4 * tx + j];

    TH* Hg = static_cast<TH*>(a.H);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = row0 + RPT * ty + i;
        float part[CG];
#pragma unroll
        for (int g = 0; g < CG; ++g) {
            part[g] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][4 * g + j] = Slice<TH>::round(acc[i][4 * g + j]);
                part[g] = fmaf(acc[i][4 * g + j], av[4 * g + j], part[g]);
            }
        }
        if (HEADS == 1) {
            float p = part[0];
#pragma unroll
            for (int g = 1; g < CG; ++g) p += part[g];
            p = lanes_sum<16>(p);
            if (tx == 0 && row < row_end) a.S[row] = p;
        } else {
            static_assert(HEADS == 1 || HW == 8, "multi-head stages use 8-wide heads");
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                const float p = lanes_sum<2>(part[g]);
                if ((tx & 1) == 0 && row < row_end) a.S[(size_t)row * HEADS + 8 * g + (tx >> 1)] = p;
            }
        }
        if (row < row_end) {
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                TH* p = Hg + (size_t)row * DOUT + 64 * g + 4 * tx;
                if constexpr (sizeof(TH) == 4) {
                    *reinterpret_cast<float4*>(p) = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
                } else {
                    const uint32_t lo = f32_to_bf16(acc[i][4 * g]) | (f32_to_bf16(acc[i][4 * g + 1]) << 16);
                    const uint32_t hi = f32_to_bf16(acc[i][4 * g + 2]) | (f32_to_bf16(acc[i][4 * g + 3]) << 16);
                    *reinterpret_cast<uint2*>(p) = make_uint2(lo, hi);
                }
            }
        }
    }
}

template <typename TX, typename TH, int DIN, int DOUT, int HEADS, int RPT>
static int launch_transform_rpt(const TransArgs& a, bool drop, hipStream_t st) {
    constexpr int TM = 16 * RPT;
    const int nb = (a.u_n + TM - 1) / TM + (a.i_n + TM - 1) / TM;
    if (nb == 0) return 0;
    constexpr size_t lds = (size_t)(TM * (DIN + 4) + DIN * DOUT) * sizeof(float);
    if (drop) {
        auto k = transform_fwd_kernel<TX, TH, DIN, DOUT, HEADS, true, RPT>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        launch_k(NGACF_TAG_TRANSFORM_FWD, k, dim3(nb), dim3(256), lds, st, a);
    } else {
        auto k = transform_fwd_kernel<TX, TH, DIN, DOUT, HEADS, false, RPT>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        launch_k(NGACF_TAG_TRANSFORM_FWD, k, dim3(nb), dim3(256), lds, st, a);
    }
    NGACF_CHECK_LAUNCH();
    return 0;
}

// rows per thread: the smallest tile whose grid fits one residency round (1024 workgroups); 64-row
// tiles when the graph needs many rounds anyway
template <typename TX, typename TH, int DIN, int DOUT, int HEADS>
static int launch_transform(const TransArgs& a, bool drop, hipStream_t st) {
    auto blocks = [&](int tm) { return (a.u_n + tm - 1) / tm + (a.i_n + tm - 1) / tm; };
    if (blocks(64) <= 1024 || blocks(96) > 1024) return launch_transform_rpt<TX, TH, DIN, DOUT, HEADS, 4>(a, drop, st);
    if (blocks(80) <= 1024) return launch_transform_rpt<TX, TH, DIN, DOUT, HEADS, 5>(a, drop, st);
    return launch_transform_rpt<TX, TH, DIN, DOUT, HEADS, 6>(a, drop, st);
}

// fp32 rows of 64 columns: the exact-fp32 matrix-core form (mfma_rows.hpp: f32_tile_step, the step the traversal epilogue runs:
// same fmaf chains, same bits as transform_fwd_kernel above).  A workgroup keeps W'[type] in LDS (LDS-DMA, 16 / 32 KB) and walks
// `tpb` consecutive 16-row tiles: 256 threads fetch a tile as ONE coalesced 4 KB read (thread = 16 bytes), the next tile's read is
// in flight while this one's 16 MFMAs per wave run.  15.2 -> 10-11 us at Gowalla size (transform_fwd_kernel stays for the other
// shapes: 128 input columns, bf16 input rows).
template <int NEXT, bool NONE, bool DROP>
__global__ __launch_bounds__(256) void transform_f32_tiles_kernel(const TransArgs a, int tpb) {
    __shared__ __attribute__((aligned(16))) float panel[64 * NEXT];
    __shared__ __attribute__((aligned(16))) float xtile[16 * XROW_F32];
    __shared__ int xmeta[16];
    __shared__ __attribute__((aligned(16))) float avs[NEXT];
    __shared__ float sx[64];
    const int tid = threadIdx.x, row = tid >> 4, q = tid & 15;
    const int tiles_u = (a.u_n + 15) / 16, tiles_i = (a.i_n + 15) / 16;
    const int nbu = (tiles_u + tpb - 1) / tpb;
    const bool users = (int)blockIdx.x < nbu;
    const int type = users ? 0 : 1;
    const int bl = users ? blockIdx.x : blockIdx.x - nbu;
    const int win_lo = users ? a.u_lo : a.U + a.i_lo;
    const int win_end = win_lo + (users ? a.u_n : a.i_n);
    const int ntiles = users ? tiles_u : tiles_i;
    const int loc_off = users ? a.u_lo : a.U + a.i_lo - a.u_n;  // global row - loc_off = local row
    const uint32_t salt = DROP ? salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u) : 0u;
    load_panel_f32<NEXT>(panel, a.W + (size_t)type * 64 * NEXT);
    if (tid < NEXT / 4) reinterpret_cast<float4*>(avs)[tid] = reinterpret_cast<const float4*>(a.avec + type * NEXT)[tid];
    const float* X = static_cast<const float*>(a.X);
    auto fetch = [&](int it, float4& v, int& node) {
        const int tile = bl * tpb + it;
        const int nd = win_lo + tile * 16 + row;
        const bool ok = tile < ntiles && nd < win_end;
        node = ok ? nd : -1;
        v = ok ? *reinterpret_cast<const float4*>(X + (size_t)(a.x_local ? nd - loc_off : nd) * 64 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 v;
    int node;
    fetch(0, v, node);
    for (int it = 0; it < tpb; ++it) {
        if (bl * tpb + it >= ntiles) break;                  // (uniform)
        float x[4] = {v.x, v.y, v.z, v.w};
        if (DROP && node >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = feat_bits(salt, node, 4 * q + j) >= a.drop_thr ? x[j] * a.drop_scale : 0.f;
        }
        *reinterpret_cast<float4*>(xtile + row * XROW_F32 + 4 * q) = make_float4(x[0], x[1], x[2], x[3]);
        if (q == 0) xmeta[row] = node;
        if (it + 1 < tpb) fetch(it + 1, v, node);            // the next tile's rows fly under this tile's step
        __syncthreads();                                     // tile (and, the first time, panel + avec) in LDS
        f32_tile_step<NEXT, NONE>(xtile, xmeta, panel, avs, sx, a.H, a.S);
        __syncthreads();                                     // the tile may be overwritten
    }
}

template <int NEXT, bool NONE>
static int launch_transform_f32_tiles(const TransArgs& a, bool drop, hipStream_t st) {
    const int tiles = (a.u_n + 15) / 16 + (a.i_n + 15) / 16;
    int tpb = (tiles + 2047) / 2048;                         // ~2 k workgroups: the 16-32 KB panel is read once per 16 * tpb rows
    if (tpb < 3) tpb = 3;
    const int nb = ((a.u_n + 15) / 16 + tpb - 1) / tpb + ((a.i_n + 15) / 16 + tpb - 1) / tpb;
    if (nb == 0) return 0;
    if (drop) launch_k(NGACF_TAG_TRANSFORM_FWD, transform_f32_tiles_kernel<NEXT, NONE, true>, dim3(nb), dim3(256), 0, st, a, tpb);
    else      launch_k(NGACF_TAG_TRANSFORM_FWD, transform_f32_tiles_kernel<NEXT, NONE, false>, dim3(nb), dim3(256), 0, st, a, tpb);
    NGACF_CHECK_LAUNCH();
    return 0;
}

template <typename TX, typename TH>
static int dispatch_transform(const TransArgs& a, int din, int dout, int heads, bool drop, hipStream_t st) {
    if constexpr (sizeof(TX) == 4 && sizeof(TH) == 4) {       // fp32 rows in, fp32 rows out, 64 columns in: the matrix-core form
        if (din == 64 && dout == 64 && heads == 8) return launch_transform_f32_tiles<64, false>(a, drop, st);
        if (din == 64 && dout == 64 && heads == 1) return launch_transform_f32_tiles<64, true>(a, drop, st);
        if (din == 64 && dout == 128 && heads == 16) return launch_transform_f32_tiles<128, false>(a, drop, st);
        if (din == 64 && dout == 128 && heads == 1) return launch_transform_f32_tiles<128, true>(a, drop, st);
    }
    if (din == 64 && dout == 64 && heads == 8) return launch_transform<TX, TH, 64, 64, 8>(a, drop, st);
    if (din == 64 && dout == 64 && heads == 1) return launch_transform<TX, TH, 64, 64, 1>(a, drop, st);
    if (din == 128 && dout == 64 && heads == 8) return launch_transform<TX, TH, 128, 64, 8>(a, drop, st);
    if (din == 128 && dout == 64 && heads == 1) return launch_transform<TX, TH, 128, 64, 1>(a, drop, st);
    if (din == 64 && dout == 128 && heads == 1) return launch_transform<TX, TH, 64, 128, 1>(a, drop, st);
    if (din == 64 && dout == 128 && heads == 16) return launch_transform<TX, TH, 64, 128, 16>(a, drop, st);
    if (din == 128 && dout == 128 && heads == 1) return launch_transform<TX, TH, 128, 128, 1>(a, drop, st);
    if (din == 128 && dout == 128 && heads == 16) return launch_transform<TX, TH, 128, 128, 16>(a, drop, st);
    return NGACF_E_SHAPE;
}



// transform_mfma.hip: the bf16-storage path on the matrix cores
int transform_fwd_mfma(const void* X, int x_dtype, const float* W, const float* avec, void* H, float* S,
                       int U, int I, int din, int dout, int heads, int u_lo, int u_n, int i_lo, int i_n, int x_local,
                       float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t st,
                       int n_pack, const float* const* pack_W, void* const* pack_panel, const int* pack_din, const int* pack_dout,
                       const int* pack_kind);

}  // namespace ngacf

using namespace ngacf;

extern "C" int ngacf_transform_fwd_rows(const void* X, int x_dtype, const float* W, const float* avec,
                                        void* H, int h_dtype, float* S,
                                        int U, int I, int din, int dout, int heads,
                                        int u_lo, int u_n, int i_lo, int i_n, int x_local,
                                        float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream) {
    if (U < 0 || I < 0 || u_lo < 0 || u_n < 0 || i_lo < 0 || i_n < 0 || u_lo + u_n > U || i_lo + i_n > I) return NGACF_E_ARG;
    if (u_n + i_n == 0) return 0;
    if (!X || !W || !avec || !H || !S) return NGACF_E_ARG;
    if (!(feat_drop_p >= 0.f && feat_drop_p < 1.f)) return NGACF_E_ARG;
    TransArgs a;
    a.X = X; a.W = W; a.avec = avec; a.H = H; a.S = S; a.U = U; a.I = I;
    a.u_lo = u_lo; a.u_n = u_n; a.i_lo = i_lo; a.i_n = i_n; a.x_local = x_local;
    a.drop_thr = drop_threshold(feat_drop_p); a.drop_scale = 1.f / (1.f - feat_drop_p);
    a.seed = seed; a.seed_dev = seed_dev;
    const bool drop = feat_drop_p > 0.f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((x_dtype != NGACF_F32 && x_dtype != NGACF_BF16) || (h_dtype != NGACF_F32 && h_dtype != NGACF_BF16)) return NGACF_E_SHAPE;
    if (h_dtype == NGACF_BF16)                     // bf16 storage: bf16 MFMA, fp32 accumulate
        return transform_fwd_mfma(X, x_dtype, W, avec, H, S, U, I, din, dout, heads, u_lo, u_n, i_lo, i_n, x_local,
                                  feat_drop_p, seed, seed_dev, st, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (x_dtype == NGACF_F32) return dispatch_transform<float, float>(a, din, dout, heads, drop, st);
    return dispatch_transform<bf16_t, float>(a, din, dout, heads, drop, st);
}

extern "C" int ngacf_transform_fwd(const void* X, int x_dtype, const float* W, const float* avec,
                                   void* H, int h_dtype, float* S,
                                   int U, int I, int din, int dout, int heads,
                                   float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream) {
    return ngacf_transform_fwd_rows(X, x_dtype, W, avec, H, h_dtype, S, U, I, din, dout, heads, 0, U, 0, I, 0,
                                    feat_drop_p, seed, seed_dev, stream);
}
