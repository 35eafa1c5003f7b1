// Backward of the row transform (transform.hip): dX = mask * (dH @ W^T), dW = (mask*X)^T @ dH,
// davec = sum_n dS[n,head(c)] H[n,c].  Autograd of graphattention/SPUIGACF.py:356-357,:361.
#include "ngacf_common.hpp"

namespace ngacf {

constexpr int TM = 64;     // rows per workgroup tile

// ===========================================================================
// backward
// ===========================================================================
struct TransBwdArgs {
    const void* X;
    const float* W;
    const void* H;
    const void* dH;                // [N][dout] in H's storage type
    const float* dS;
    void* dX;                      // [rows][din] of X's type
    float* work;
    int U, I;
    int u_lo, u_n, i_lo, i_n;      // row windows (see transform.hip)
    int x_local;                   // X and dX hold only the window rows, users first
    uint32_t drop_thr;
    float drop_scale;
    float drop_p;
    uint64_t seed;
    const uint64_t* seed_dev;
    int nbu, nbi;          // workgroups of the dW kernel given to user rows / item rows
    // ngacf_transform_dx_node (fp32 storage): step 1 of the PREVIOUS stage's backward on the dX rows, in the dX kernel's epilogue
    const float* pv_out;   // [N][din]
    const float* pv_agg;   // [2][N][din]
    const float* pv_R;     // [2][N][din/8]
    const float* pv_S;     // [N][din/8]
    float* pv_Gn;          // [N][din]
    float2* pv_ST;         // [N][din/8]
    float* pv_dS;          // [N][din/8]
    float pv_alpha;
    int pv_flags;
};

// dX = mask * (dH @ W[type]^T): the forward tile kernel with the panel transposed on its way into LDS.
// NODE (ngacf_transform_dx_node; fp32 rows, the previous stage in 8-column heads): the thread that holds four columns of a dX row
// also runs that row's share of step 1 of the previous stage's backward -- prop_bwd_node_kernel<float, 64, 8>'s arithmetic in its
// own order (a head = two neighbouring lanes, folded by the same DPP add): the same bits, without dX's trip through memory.
template <typename TX, int DIN, int DOUT, bool DROP, bool NODE = false>
__global__ __launch_bounds__(256) void transform_bwd_dx_kernel(const TransBwdArgs a) {
    constexpr int GS = DOUT + 4;                   // LDS row stride of the dH tile
    constexpr int WS = DIN + 4;                    // LDS row stride of W^T  [DOUT][DIN]
    constexpr int CG = DIN / 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Gs = lds;                               // [TM][GS]
    float* Wt = lds + TM * GS;                     // [DOUT][WS]
    const int tid = threadIdx.x;
    const uint32_t salt = DROP ? salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u) : 0u;
    const int nbu = (a.u_n + TM - 1) / TM;
    const bool users = (int)blockIdx.x < nbu;
    const int type = users ? 0 : 1;
    const int row0 = users ? a.u_lo + blockIdx.x * TM : a.U + a.i_lo + (blockIdx.x - nbu) * TM;
    const int row_end = users ? a.u_lo + a.u_n : a.U + a.i_lo + a.i_n;
    const int xoff = a.x_local ? (users ? a.u_lo : a.U + a.i_lo - a.u_n) : 0;
    {
        const float* Wg = a.W + (size_t)type * DIN * DOUT;
        for (int i = tid; i < DIN * DOUT / 4; i += 256) {
            const int k = i % DIN, c4 = i / DIN;   // consecutive lanes -> consecutive k: conflict-free LDS stores
            const float4 v = *reinterpret_cast<const float4*>(Wg + (size_t)k * DOUT + 4 * c4);
            Wt[(4 * c4 + 0) * WS + k] = v.x;
            Wt[(4 * c4 + 1) * WS + k] = v.y;
            Wt[(4 * c4 + 2) * WS + k] = v.z;
            Wt[(4 * c4 + 3) * WS + k] = v.w;
        }
        constexpr int PER_ROW = DOUT / 4;
        for (int i = tid; i < TM * PER_ROW; i += 256) {
            const int r = i / PER_ROW, cv = i % PER_ROW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < row_end) v = *reinterpret_cast<const float4*>(static_cast<const float*>(a.dH) + (size_t)(row0 + r) * DOUT + 4 * cv);
            *reinterpret_cast<float4*>(Gs + r * GS + 4 * cv) = v;
        }
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    // NODE: the previous stage's rows of this thread's four dX rows are requested HERE, in front of the contraction, which hides
    // their trip (loaded in the epilogue they cost the launch as much as the node step's own kernel: 54 against 24 + 28 us)
    float4 pv_o[NODE ? 4 : 1], pv_a[NODE ? 4 : 1], pv_n[NODE ? 4 : 1];
    float pv_r[NODE ? 4 : 1], pv_rn[NODE ? 4 : 1], pv_s[NODE ? 4 : 1];
    if constexpr (NODE) {
        const size_t Nn = (size_t)a.U + a.I;
        const int head = tx >> 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * ty + i;
            const size_t n = row < row_end ? (size_t)row : (size_t)(row_end - 1);
            pv_o[i] = *reinterpret_cast<const float4*>(a.pv_out + n * DIN + 4 * tx);
            pv_a[i] = *reinterpret_cast<const float4*>(a.pv_agg + n * DIN + 4 * tx);
            pv_n[i] = *reinterpret_cast<const float4*>(a.pv_agg + (Nn + n) * DIN + 4 * tx);
            pv_r[i] = a.pv_R[n * 8 + head];
            pv_rn[i] = a.pv_R[(Nn + n) * 8 + head];
            pv_s[i] = a.pv_S[n * 8 + head];
        }
    }
    float acc[4][4 * CG];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4 * CG; ++j) acc[i][j] = 0.f;
#pragma unroll 4
    for (int c4 = 0; c4 < DOUT / 4; ++c4) {
        float4 ga[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ga[i] = *reinterpret_cast<const float4*>(Gs + (4 * ty + i) * GS + 4 * c4);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                const float4 wb = *reinterpret_cast<const float4*>(Wt + (4 * c4 + cc) * WS + 64 * g + 4 * tx);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = cc == 0 ? ga[i].x : cc == 1 ? ga[i].y : cc == 2 ? ga[i].z : ga[i].w;
                    acc[i][4 * g + 0] = fmaf(x, wb.x, acc[i][4 * g + 0]);
                    acc[i][4 * g + 1] = fmaf(x, wb.y, acc[i][4 * g + 1]);
                    acc[i][4 * g + 2] = fmaf(x, wb.z, acc[i][4 * g + 2]);
                    acc[i][4 * g + 3] = fmaf(x, wb.w, acc[i][4 * g + 3]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * ty + i;
        if (row >= row_end) continue;
#pragma unroll
        for (int g = 0; g < CG; ++g) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = acc[i][4 * g + j];
                if (DROP) o[j] = feat_bits(salt, row, 64 * g + 4 * tx + j) >= a.drop_thr ? o[j] * a.drop_scale : 0.f;
            }
            if constexpr (NODE) {
                static_assert(!NODE || (sizeof(TX) == 4 && DIN == 64), "dX + node step: fp32 rows of 64 columns");
                const size_t n = (size_t)row;
                const int head = tx >> 1;
                const float4 ov = pv_o[i], av = pv_a[i], nv = pv_n[i];
                const float r = pv_r[i], rn = pv_rn[i];
                const float outv[4] = {ov.x, ov.y, ov.z, ov.w}, ag[4] = {av.x, av.y, av.z, av.w}, an[4] = {nv.x, nv.y, nv.z, nv.w};
                float gg[4], gn[4], p = 0.f, pn = 0.f;
                const float inv = r > 0.f ? 1.f / r : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gg[j] = o[j];
                    if (a.pv_flags & NGACF_ELU) gg[j] *= outv[j] > 0.f ? 1.f : outv[j] + 1.f;
                    gn[j] = gg[j] * inv;
                    p = fmaf(gn[j], ag[j], p);
                    pn = fmaf(gn[j], an[j], pn);
                }
                p = lanes_sum<2>(p);
                pn = lanes_sum<2>(pn);
                *reinterpret_cast<float4*>(a.pv_Gn + n * DIN + 4 * tx) = make_float4(gn[0], gn[1], gn[2], gn[3]);
                if ((tx & 1) == 0) {
                    a.pv_ST[n * 8 + head] = make_float2(pv_s[i], p);
                    a.pv_dS[n * 8 + head] = own_term(a.pv_alpha, r, rn, p, pn);
                }
            }
            if constexpr (sizeof(TX) == 4) {
                *reinterpret_cast<float4*>(static_cast<float*>(a.dX) + (size_t)(row - xoff) * DIN + 64 * g + 4 * tx) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                const uint32_t lo = f32_to_bf16(o[0]) | (f32_to_bf16(o[1]) << 16), hi = f32_to_bf16(o[2]) | (f32_to_bf16(o[3]) << 16);
                *reinterpret_cast<uint2*>(static_cast<uint16_t*>(a.dX) + (size_t)(row - xoff) * DIN + 64 * g + 4 * tx) = make_uint2(lo, hi);
            }
        }
    }
}

// Partial dW = Xmasked^T @ dH and davec = sum_n dS[n,head(c)] H[n,c] over a strided set of 64-row tiles;
// workgroup b leaves its partial in work[b][DIN*DOUT + DOUT].
template <typename TX, typename TH, int DIN, int DOUT, int HEADS, bool DROP>
__global__ __launch_bounds__(256) void transform_bwd_dw_kernel(const TransBwdArgs a) {
    constexpr int XS = DIN + 4, GS = DOUT + 4;
    constexpr int RI = DIN / 16, RJ = DOUT / 16;   // per-thread tile of dW
    constexpr int HW = DOUT / HEADS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                               // [TM][XS]
    float* Gs = lds + TM * XS;                     // [TM][GS]
    const int tid = threadIdx.x;
    const uint32_t salt = DROP ? salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u) : 0u;
    const bool users = (int)blockIdx.x < a.nbu;
    const int nb = users ? a.nbu : a.nbi;
    const int bl = users ? blockIdx.x : blockIdx.x - a.nbu;
    const int rbeg = users ? a.u_lo : a.U + a.i_lo, rend = rbeg + (users ? a.u_n : a.i_n);
    const int xoff = a.x_local ? (users ? a.u_lo : a.U + a.i_lo - a.u_n) : 0;
    const int ntiles = (rend - rbeg + TM - 1) / TM;
    const int ty = tid >> 4, tx = tid & 15;
    float acc[RI][RJ];
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < RJ; ++j) acc[i][j] = 0.f;
    float da = 0.f;
    const TX* Xg = static_cast<const TX*>(a.X);
    const TH* Hg = static_cast<const TH*>(a.H);

    for (int tile = bl; tile < ntiles; tile += nb) {
        const int row0 = rbeg + tile * TM;
        __syncthreads();
        {
            constexpr int XV = Slice<TX>::VEC, PER_ROW = DIN / XV;
            for (int i = tid; i < TM * PER_ROW; i += 256) {
                const int r = i / PER_ROW, cv = i % PER_ROW;
                float f[XV];
                if (row0 + r < rend) {
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
            constexpr int GPR = DOUT / 4;
            for (int i = tid; i < TM * GPR; i += 256) {
                const int r = i / GPR, cv = i % GPR;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row0 + r < rend) v = *reinterpret_cast<const float4*>(static_cast<const float*>(a.dH) + (size_t)(row0 + r) * DOUT + 4 * cv);
                *reinterpret_cast<float4*>(Gs + r * GS + 4 * cv) = v;
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int r = 0; r < TM; ++r) {
            float xa[RI], gb[RJ];
#pragma unroll
            for (int i = 0; i < RI; i += 4) {
                const float4 v = *reinterpret_cast<const float4*>(Xs + r * XS + RI * ty + i);
                xa[i] = v.x; xa[i + 1] = v.y; xa[i + 2] = v.z; xa[i + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < RJ; j += 4) {
                const float4 v = *reinterpret_cast<const float4*>(Gs + r * GS + RJ * tx + j);
                gb[j] = v.x; gb[j + 1] = v.y; gb[j + 2] = v.z; gb[j + 3] = v.w;
            }
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int j = 0; j < RJ; ++j) acc[i][j] = fmaf(xa[i], gb[j], acc[i][j]);
        }
        {   // davec: thread (rg, c) walks rows rg, rg+RG, ... of the tile; loads are independent
            constexpr int RG = 256 / DOUT;
            const int c = tid % DOUT, rg = tid / DOUT;
#pragma unroll 8
            for (int r = rg; r < TM; r += RG) {
                const int row = row0 + r;
                const bool ok = row < rend;
                const size_t n = (size_t)(ok ? row : rend - 1);
                float h;
                if constexpr (sizeof(TH) == 4) h = reinterpret_cast<const float*>(Hg)[n * DOUT + c];
                else h = bf16_to_f32(reinterpret_cast<const uint16_t*>(Hg)[n * DOUT + c]);
                const float s = a.dS[n * HEADS + c / HW];
                da = fmaf(ok ? s : 0.f, h, da);
            }
        }
    }
    {   // fold the row groups of davec through LDS
        constexpr int RG = 256 / DOUT;
        __syncthreads();
        lds[tid] = da;
        __syncthreads();
        if (tid < DOUT) {
            da = 0.f;
#pragma unroll
            for (int g = 0; g < RG; ++g) da += lds[g * DOUT + tid];
        }
    }
    float* wk = a.work + (size_t)blockIdx.x * (DIN * DOUT + DOUT);
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < RJ; j += 4)
            *reinterpret_cast<float4*>(wk + (size_t)(RI * ty + i) * DOUT + RJ * tx + j) =
                make_float4(acc[i][j], acc[i][j + 1], acc[i][j + 2], acc[i][j + 3]);
    if (tid < DOUT) wk[DIN * DOUT + tid] = da;
}


// ---------------------------------------------------------------------------------------------------------
// bf16-storage form of the dW / davec partial on the matrix cores:  dW[k][c] = sum_n Xm[n][k] dH[n][c].
// The contraction runs over NODES, so both MFMA operands want "8 consecutive nodes of one column" per lane:
// a 64-node tile of X and of dH (rounded to bf16) is staged row-major in LDS with coalesced 16-byte stores
// and consumed column-major with ds_read_b64_tr_b16 (the hardware transpose read of gfx950).  The 4 waves
// own the 4 quadrants of the [DIN][DOUT] result, so there is no cross-wave reduction; the next tile's global
// loads are in flight while the current one is multiplied.  davec rides along on the VALU.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 dw_bf16x8;
typedef __attribute__((ext_vector_type(4))) short dw_short4;
typedef __attribute__((ext_vector_type(8))) short dw_short8;
typedef __attribute__((ext_vector_type(16))) float dw_f32x16;

__device__ __forceinline__ dw_bf16x8 tr_frag(const char* lds_base, int off_lo, int off_hi) {
    typedef __attribute__((address_space(3))) dw_short4 lds_short4;
    const dw_short4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(lds_base + off_lo));
    const dw_short4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(lds_base + off_hi));
    const dw_short8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(dw_bf16x8, v);
}

__device__ __forceinline__ uint4 pack_bf16x8(const float (&f)[8]) {
    return make_uint4(f32_to_bf16(f[0]) | (f32_to_bf16(f[1]) << 16), f32_to_bf16(f[2]) | (f32_to_bf16(f[3]) << 16),
                      f32_to_bf16(f[4]) | (f32_to_bf16(f[5]) << 16), f32_to_bf16(f[6]) | (f32_to_bf16(f[7]) << 16));
}

// WITH_DX: the same workgroup also produces dX = mask * (dH @ W^T) for its tile -- the dH tile is already in LDS as bf16
// (the separate dX kernel reads and rounds the same values), node on the lane as in transform_mfma.hip: the wave takes
// the 32-node half (wave & 1) and the output column tiles of (wave >> 1).
template <typename TX, int DIN, int DOUT, int HEADS, bool DROP, bool WITH_DX>
__global__ __launch_bounds__(256) void transform_dw_mfma_kernel(const TransBwdArgs a) {
    constexpr int TN = 64;                               // nodes per tile
    constexpr int SX = DIN * 2 + 64, SG = DOUT * 2 + 64;  // LDS row strides (bytes): 4 consecutive rows on disjoint 16-bank windows
    constexpr int KW = DIN / 64, CW = DOUT / 64;          // 32-blocks of k / of c per wave (2 x 2 wave grid)
    constexpr int HW = DOUT / HEADS;
    constexpr int XI = TN * (DIN / 8) / 256, GI = TN * (DOUT / 8) / 256;   // 8-element items per thread per tile
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    char* Xs = lds_raw;                                  // [TN][SX]
    char* Gs = lds_raw + TN * SX;                        // [TN][SG]
    constexpr int DKS = DOUT / 16, DOT = DIN / 32;       // dX: k-steps over dout, 32-column tiles of din
    dw_bf16x8* Af = reinterpret_cast<dw_bf16x8*>(lds_raw + TN * SX + TN * SG);   // [DKS][DOT][64] W^T fragments (WITH_DX)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wc = wave & 1;
    const uint32_t salt = DROP ? salt_of(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u) : 0u;
    const bool users = (int)blockIdx.x < a.nbu;
    const int nb = users ? a.nbu : a.nbi;
    const int bl = users ? blockIdx.x : blockIdx.x - a.nbu;
    const int rbeg = users ? a.u_lo : a.U + a.i_lo, rend = rbeg + (users ? a.u_n : a.i_n);
    const int xoff = a.x_local ? (users ? a.u_lo : a.U + a.i_lo - a.u_n) : 0;
    const int ntiles = (rend - rbeg + TN - 1) / TN;
    const TX* Xg = static_cast<const TX*>(a.X);
    const uint16_t* Hg = static_cast<const uint16_t*>(a.H);

    dw_f32x16 acc[KW][CW];
#pragma unroll
    for (int i = 0; i < KW; ++i)
#pragma unroll
        for (int j = 0; j < CW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float da[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) da[e] = 0.f;

    // staged registers of one tile
    uint4 xr[XI], gr[GI];
    auto fetch = [&](int tile) {
        const int row0 = rbeg + tile * TN;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int i = tid + 256 * j, r = i / (DIN / 8), q = i % (DIN / 8);
            const int row = row0 + r;
            float f[8];
            if (row < rend) {
                if constexpr (sizeof(TX) == 2) {
                    if constexpr (!DROP) {
                        xr[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(Xg) + (size_t)(row - xoff) * DIN + 8 * q);
                        continue;
                    }
                    Slice<bf16_t>::load(reinterpret_cast<const bf16_t*>(Xg) + (size_t)(row - xoff) * DIN + 8 * q, f);
                } else {
                    const float4 v0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Xg) + (size_t)(row - xoff) * DIN + 8 * q);
                    const float4 v1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Xg) + (size_t)(row - xoff) * DIN + 8 * q + 4);
                    f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
                }
                if (DROP) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = feat_bits(salt, row, 8 * q + e) >= a.drop_thr ? f[e] * a.drop_scale : 0.f;
                }
                xr[j] = pack_bf16x8(f);
            } else {
                xr[j] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
#pragma unroll
        for (int j = 0; j < GI; ++j) {
            const int i = tid + 256 * j, r = i / (DOUT / 8), q = i % (DOUT / 8);
            const int row = row0 + r;
            if (row < rend) {
                gr[j] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(a.dH) + (size_t)row * DOUT + 8 * q);   // bf16 as stored
                // davec: this thread's 8 columns of row `row` (the column group q is the same for every item)
                const uint4 hv = *reinterpret_cast<const uint4*>(Hg + (size_t)row * DOUT + 8 * q);
                const float s = a.dS[(size_t)row * HEADS + (8 * q) / HW];
                const uint32_t w[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    da[2 * e] = fmaf(s, bf16_to_f32(w[e] & 0xffffu), da[2 * e]);
                    da[2 * e + 1] = fmaf(s, bf16_to_f32(w[e] >> 16), da[2 * e + 1]);
                }
            } else {
                gr[j] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int i = tid + 256 * j, r = i / (DIN / 8), q = i % (DIN / 8);
            *reinterpret_cast<uint4*>(Xs + r * SX + 16 * q) = xr[j];
        }
#pragma unroll
        for (int j = 0; j < GI; ++j) {
            const int i = tid + 256 * j, r = i / (DOUT / 8), q = i % (DOUT / 8);
            *reinterpret_cast<uint4*>(Gs + r * SG + 16 * q) = gr[j];
        }
    };

    // transposed-read addressing: 16-lane group g = lane>>4 takes nodes 8*(g>>1).. of columns 16*(g&1)..; lane 4q+p of
    // the group points at node q, columns 4p..4p+3
    const int g4 = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int node_in_step = 8 * (g4 >> 1) + qq;
    const int col_in_blk = 16 * (g4 & 1) + 4 * pp;

    int tile = bl;
    if (tile < ntiles) fetch(tile);
    if constexpr (WITH_DX) {     // A[i][k] = W[type][i][k], i over din, k over dout (built under the first tile's loads)
        const float* M = a.W + (size_t)(users ? 0 : 1) * DIN * DOUT;
        for (int slot = tid; slot < DKS * DOT * 64; slot += 256) {
            const int l = slot & 63, t = (slot >> 6) % DOT, sk = (slot >> 6) / DOT;
            const int i = 32 * t + (l & 31), k0 = 16 * sk + 8 * (l >> 5);
            dw_bf16x8 fr;
#pragma unroll
            for (int j = 0; j < 8; ++j) fr[j] = static_cast<__bf16>(M[(size_t)i * DOUT + k0 + j]);
            Af[slot] = fr;
        }
    }
    for (; tile < ntiles; tile += nb) {
        __syncthreads();                                  // everyone done reading the previous tile
        stage();
        __syncthreads();
        if (tile + nb < ntiles) fetch(tile + nb);         // next tile's loads fly under the MFMAs
#pragma unroll
        for (int s = 0; s < TN / 16; ++s) {
            dw_bf16x8 af[KW], bf[CW];
#pragma unroll
            for (int i = 0; i < KW; ++i) {
                const int off = (16 * s + node_in_step) * SX + (32 * (wk * KW + i) + col_in_blk) * 2;
                af[i] = tr_frag(Xs, off, off + 4 * SX);
            }
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                const int off = (16 * s + node_in_step) * SG + (32 * (wc * CW + j) + col_in_blk) * 2;
                bf[j] = tr_frag(Gs, off, off + 4 * SG);
            }
#pragma unroll
            for (int i = 0; i < KW; ++i)
#pragma unroll
                for (int j = 0; j < CW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (WITH_DX) {
            const int sub = wave & 1, r = lane & 31, hh = lane >> 5;
            const int node = rbeg + tile * TN + 32 * sub + r;
            const bool okn = node < rend;
            dw_bf16x8 gb[DKS];
#pragma unroll
            for (int sk = 0; sk < DKS; ++sk)
                gb[sk] = *reinterpret_cast<const dw_bf16x8*>(Gs + (32 * sub + r) * SG + (16 * sk + 8 * hh) * 2);
            constexpr int TPW = DOT / 2;                     // column tiles of this wave
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) {
                const int t = (wave >> 1) * TPW + tt;
                dw_f32x16 dacc;
#pragma unroll
                for (int e = 0; e < 16; ++e) dacc[e] = 0.f;
#pragma unroll
                for (int sk = 0; sk < DKS; ++sk)
                    dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[(sk * DOT + t) * 64 + lane], gb[sk], dacc, 0, 0, 0);
                // lane: node r (+ half hh); dacc[4g + j] = dX column 32t + 8g + 4hh + j
                const size_t orow = (size_t)(node - xoff);
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    float v[2][4];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[gg][j] = dacc[4 * (gp + gg) + j];
                            if (DROP) v[gg][j] = feat_bits(salt, node, 32 * t + 8 * (gp + gg) + 4 * hh + j) >= a.drop_thr ? v[gg][j] * a.drop_scale : 0.f;
                        }
                    if constexpr (sizeof(TX) == 4) {
                        if (okn) {
#pragma unroll
                            for (int gg = 0; gg < 2; ++gg)
                                *reinterpret_cast<float4*>(static_cast<float*>(a.dX) + orow * DIN + 32 * t + 8 * (gp + gg) + 4 * hh) =
                                    make_float4(v[gg][0], v[gg][1], v[gg][2], v[gg][3]);
                        }
                    } else {
                        uint2 pk[2];
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg)
                            pk[gg] = make_uint2(f32_to_bf16(v[gg][0]) | (f32_to_bf16(v[gg][1]) << 16), f32_to_bf16(v[gg][2]) | (f32_to_bf16(v[gg][3]) << 16));
                        const auto rx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                        const auto ry = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                        if (okn)
                            *reinterpret_cast<uint4*>(static_cast<char*>(a.dX) + orow * DIN * 2 + 64 * t + 16 * gp + 16 * hh) =
                                make_uint4(rx[0], ry[0], rx[1], ry[1]);
                    }
                }
            }
        }
    }

    // partial of this workgroup: acc[i][j][reg] = dW[32(wk KW + i) + (reg&3) + 8(reg>>2) + 4(lane>>5)][32(wc CW + j) + (lane&31)]
    float* wkp = a.work + (size_t)blockIdx.x * (DIN * DOUT + DOUT);
#pragma unroll
    for (int i = 0; i < KW; ++i)
#pragma unroll
        for (int j = 0; j < CW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = 32 * (wk * KW + i) + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                wkp[(size_t)k * DOUT + 32 * (wc * CW + j) + (lane & 31)] = acc[i][j][e];
            }
    {   // davec: fold the threads that share a column group
        constexpr int QN = DOUT / 8, TG = 256 / QN;
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds_raw);   // [256][8]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = da[e];
        __syncthreads();
        if (tid < DOUT) {
            const int q = tid >> 3, e = tid & 7;
            float s = 0.f;
#pragma unroll 8
            for (int m = 0; m < TG; ++m) s += red[(q + QN * m) * 8 + e];
            wkp[DIN * DOUT + tid] = s;
        }
    }
}

// dW[type], davec[type] = sum of the partials of that type's workgroups, in workgroup order
__global__ __launch_bounds__(256) void transform_bwd_reduce_kernel(const float* __restrict__ work, float* __restrict__ dW,
                                                                   float* __restrict__ davec, int nbu, int nbi, int wsz, int dwsz) {
    // 64 outputs per workgroup; the 4 waves take interleaved quarters of the partials, then fold in LDS
    __shared__ float red[256];
    const int o = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;                  // over [2][wsz]
    const bool ok = idx < 2 * wsz;
    const int type = ok ? idx / wsz : 0, k = ok ? idx % wsz : 0;
    const int b0 = type == 0 ? 0 : nbu, b1 = type == 0 ? nbu : nbu + nbi;
    float s = 0.f;
#pragma unroll 8
    for (int b = b0 + pg; b < b1; b += 4) s += work[(size_t)b * wsz + k];
    red[threadIdx.x] = s;
    __syncthreads();
    if (pg != 0 || !ok) return;
    s = (red[o] + red[64 + o]) + (red[128 + o] + red[192 + o]);
    if (k < dwsz) dW[(size_t)type * dwsz + k] = s;
    else davec[(size_t)type * (wsz - dwsz) + (k - dwsz)] = s;
}

// ---------------------------------------------------------------------------------------------------------
// ngacf_transform_dw_batch: the dW / davec partials of SEVERAL stages' row transforms in ONE launch (64 -> 64 stages, bf16 dH
// and H), and one launch that reduces them.  A train step whose edge backward also produces dX (ngacf_prop_bwd_dx) needs the
// weights' gradients only at the optimizer: its three per-stage dW launches -- each latency-bound at 18 MB, each in front of the
// next stage's edge backward -- become one launch behind the chain's last edge backward.  Per job the arithmetic is
// transform_dw_mfma_kernel<WITH_DX = false>'s (same tiles, same fragments, same order within a workgroup); X's type, the head
// count and the dropout threshold are run-time values of the job.
// ---------------------------------------------------------------------------------------------------------
struct DwBatchArgs {
    ngacf_dw_job job[NGACF_DW_BATCH_MAX];
    int njobs, nb;                 // workgroups per job
    int nbu, nbi;                  // ... of which user rows / item rows
    float* work;                   // [njobs][nb][64 * 64 + 64]
    int U, I;
    const uint64_t* seed_dev;
};

__global__ __launch_bounds__(256) void transform_dw_batch_kernel(const DwBatchArgs a) {
    constexpr int DIN = 64, DOUT = 64, TN = 64;
    constexpr int SX = DIN * 2 + 64, SG = DOUT * 2 + 64;
    constexpr int XI = TN * (DIN / 8) / 256, GI = TN * (DOUT / 8) / 256;
    __shared__ __attribute__((aligned(16))) char lds_raw[TN * SX + TN * SG];
    char* Xs = lds_raw;
    char* Gs = lds_raw + TN * SX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wc = wave & 1;
    const int jb = blockIdx.x / a.nb, bj = blockIdx.x % a.nb;
    const ngacf_dw_job& J = a.job[jb];
    const bool xf32 = J.x_dtype == NGACF_F32;
    const int HW = DOUT / J.heads;
    const uint32_t drop_thr = J.feat_drop_p > 0.f ? drop_threshold(J.feat_drop_p) : 0u;
    const float drop_scale = 1.f / (1.f - J.feat_drop_p);
    const bool drop = J.feat_drop_p > 0.f;
    const uint32_t salt = drop ? salt_of(J.seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u) : 0u;
    const bool users = bj < a.nbu;
    const int nb = users ? a.nbu : a.nbi;
    const int bl = users ? bj : bj - a.nbu;
    const int rbeg = users ? 0 : a.U, rend = users ? a.U : a.U + a.I;
    const int ntiles = (rend - rbeg + TN - 1) / TN;
    const uint16_t* Hg = static_cast<const uint16_t*>(J.H);

    dw_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float da[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) da[e] = 0.f;
    uint4 xr[XI], gr[GI];
    auto fetch = [&](int tile) {
        const int row0 = rbeg + tile * TN;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int i = tid + 256 * j, r = i / (DIN / 8), q = i % (DIN / 8);
            const int row = row0 + r;
            float f[8];
            if (row < rend) {
                if (!xf32) {
                    if (!drop) {
                        xr[j] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(J.X) + (size_t)row * DIN + 8 * q);
                        continue;
                    }
                    Slice<bf16_t>::load(static_cast<const bf16_t*>(J.X) + (size_t)row * DIN + 8 * q, f);
                } else {
                    const float4 v0 = *reinterpret_cast<const float4*>(static_cast<const float*>(J.X) + (size_t)row * DIN + 8 * q);
                    const float4 v1 = *reinterpret_cast<const float4*>(static_cast<const float*>(J.X) + (size_t)row * DIN + 8 * q + 4);
                    f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
                }
                if (drop) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = feat_bits(salt, row, 8 * q + e) >= drop_thr ? f[e] * drop_scale : 0.f;
                }
                xr[j] = pack_bf16x8(f);
            } else {
                xr[j] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
#pragma unroll
        for (int j = 0; j < GI; ++j) {
            const int i = tid + 256 * j, r = i / (DOUT / 8), q = i % (DOUT / 8);
            const int row = row0 + r;
            if (row < rend) {
                gr[j] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(J.dH) + (size_t)row * DOUT + 8 * q);
                const uint4 hv = *reinterpret_cast<const uint4*>(Hg + (size_t)row * DOUT + 8 * q);
                const float sc = J.dS[(size_t)row * J.heads + (8 * q) / HW];
                const uint32_t w[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    da[2 * e] = fmaf(sc, bf16_to_f32(w[e] & 0xffffu), da[2 * e]);
                    da[2 * e + 1] = fmaf(sc, bf16_to_f32(w[e] >> 16), da[2 * e + 1]);
                }
            } else {
                gr[j] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    const int g4 = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int node_in_step = 8 * (g4 >> 1) + qq;
    const int col_in_blk = 16 * (g4 & 1) + 4 * pp;
    int tile = bl;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += nb) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int i = tid + 256 * j, r = i / (DIN / 8), q = i % (DIN / 8);
            *reinterpret_cast<uint4*>(Xs + r * SX + 16 * q) = xr[j];
        }
#pragma unroll
        for (int j = 0; j < GI; ++j) {
            const int i = tid + 256 * j, r = i / (DOUT / 8), q = i % (DOUT / 8);
            *reinterpret_cast<uint4*>(Gs + r * SG + 16 * q) = gr[j];
        }
        __syncthreads();
        if (tile + nb < ntiles) fetch(tile + nb);
#pragma unroll
        for (int st = 0; st < TN / 16; ++st) {
            const int offa = (16 * st + node_in_step) * SX + (32 * wk + col_in_blk) * 2;
            const int offb = (16 * st + node_in_step) * SG + (32 * wc + col_in_blk) * 2;
            const dw_bf16x8 af = tr_frag(Xs, offa, offa + 4 * SX), bf = tr_frag(Gs, offb, offb + 4 * SG);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
        }
    }
    float* wkp = a.work + ((size_t)jb * a.nb + bj) * (DIN * DOUT + DOUT);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = 32 * wk + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        wkp[(size_t)k * DOUT + 32 * wc + (lane & 31)] = acc[e];
    }
    {
        constexpr int QN = DOUT / 8, TG = 256 / QN;
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds_raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = da[e];
        __syncthreads();
        if (tid < DOUT) {
            const int q = tid >> 3, e = tid & 7;
            float sm = 0.f;
#pragma unroll 8
            for (int m = 0; m < TG; ++m) sm += red[(q + QN * m) * 8 + e];
            wkp[DIN * DOUT + tid] = sm;
        }
    }
}

// The same for fp32 storage (H, dH, X in fp32), exact on v_mfma_f32_16x16x4_f32: the contraction runs over nodes, 4 per MFMA;
// a 64-node tile of X (masked) and of dH sits row-major in LDS (68-float stride: the four node rows of a step start on banks
// 0, 4, 8, 12), wave w owns the 16 feature rows 16 w .. of the [64][64] result and all four 16-column blocks: per step one A read,
// four B reads, four MFMAs.  The per-stage fp32 launches it replaces: an FMA kernel with 1024 partial sums + an 18 us reduction.
typedef float dwf_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void transform_dw_batch_f32_kernel(const DwBatchArgs a) {
    constexpr int DIN = 64, DOUT = 64, TN = 64, ST = 68;
    __shared__ __attribute__((aligned(16))) float Xs[TN * ST];
    __shared__ __attribute__((aligned(16))) float Gs[TN * ST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jb = blockIdx.x / a.nb, bj = blockIdx.x % a.nb;
    const ngacf_dw_job& J = a.job[jb];
    const int HW = DOUT / J.heads;
    const uint32_t drop_thr = J.feat_drop_p > 0.f ? drop_threshold(J.feat_drop_p) : 0u;
    const float drop_scale = 1.f / (1.f - J.feat_drop_p);
    const bool drop = J.feat_drop_p > 0.f;
    const uint32_t salt = drop ? salt_of(J.seed + (a.seed_dev ? *a.seed_dev : 0ull), 0u) : 0u;
    const bool users = bj < a.nbu;
    const int nb = users ? a.nbu : a.nbi;
    const int bl = users ? bj : bj - a.nbu;
    const int rbeg = users ? 0 : a.U, rend = users ? a.U : a.U + a.I;
    const int ntiles = (rend - rbeg + TN - 1) / TN;
    const float* Xg = static_cast<const float*>(J.X);
    const float* Hg = static_cast<const float*>(J.H);
    const float* Dg = static_cast<const float*>(J.dH);

    dwf_f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = dwf_f32x4{0.f, 0.f, 0.f, 0.f};
    float da[4] = {0.f, 0.f, 0.f, 0.f};
    // a thread stages 4 float4 of X and 4 of dH per tile: item i = tid + 256 j -> node i / 16, columns 4 (i % 16) .. +3
    float4 xr[4], gr[4];
    auto fetch = [&](int tile) {
        const int row0 = rbeg + tile * TN;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j, r = i >> 4, q = i & 15;
            const int row = row0 + r;
            if (row < rend) {
                float4 v = *reinterpret_cast<const float4*>(Xg + (size_t)row * DIN + 4 * q);
                if (drop) {
                    v.x = feat_bits(salt, row, 4 * q + 0) >= drop_thr ? v.x * drop_scale : 0.f;
                    v.y = feat_bits(salt, row, 4 * q + 1) >= drop_thr ? v.y * drop_scale : 0.f;
                    v.z = feat_bits(salt, row, 4 * q + 2) >= drop_thr ? v.z * drop_scale : 0.f;
                    v.w = feat_bits(salt, row, 4 * q + 3) >= drop_thr ? v.w * drop_scale : 0.f;
                }
                xr[j] = v;
                gr[j] = *reinterpret_cast<const float4*>(Dg + (size_t)row * DOUT + 4 * q);
                const float4 hv = *reinterpret_cast<const float4*>(Hg + (size_t)row * DOUT + 4 * q);
                const float sc = J.dS[(size_t)row * J.heads + (4 * q) / HW];
                da[0] = fmaf(sc, hv.x, da[0]); da[1] = fmaf(sc, hv.y, da[1]); da[2] = fmaf(sc, hv.z, da[2]); da[3] = fmaf(sc, hv.w, da[3]);
            } else {
                xr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                gr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    const int n16 = lane & 15, g4 = lane >> 4;
    int tile = bl;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += nb) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j, r = i >> 4, q = i & 15;
            *reinterpret_cast<float4*>(Xs + r * ST + 4 * q) = xr[j];
            *reinterpret_cast<float4*>(Gs + r * ST + 4 * q) = gr[j];
        }
        __syncthreads();
        if (tile + nb < ntiles) fetch(tile + nb);
#pragma unroll 4
        for (int st = 0; st < TN / 4; ++st) {
            const int node = 4 * st + g4;
            const float av = Xs[node * ST + 16 * wave + n16];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Gs[node * ST + 16 * t + n16], acc[t], 0, 0, 0);
        }
    }
    // lane (n16, g4): acc[t][j] = dW[16 wave + 4 g4 + j][16 t + n16]
    float* wkp = a.work + ((size_t)jb * a.nb + bj) * (DIN * DOUT + DOUT);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) wkp[(size_t)(16 * wave + 4 * g4 + j) * DOUT + 16 * t + n16] = acc[t][j];
    {   // davec: the 16 threads that share a column group (tid & 15) fold through LDS
        __syncthreads();
        float* red = Xs;                                     // [256][4]
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tid * 4 + e] = da[e];
        __syncthreads();
        if (tid < DOUT) {
            const int q = tid >> 2, e = tid & 3;
            float sm = 0.f;
#pragma unroll
            for (int m = 0; m < 16; ++m) sm += red[(q + 16 * m) * 4 + e];
            wkp[DIN * DOUT + tid] = sm;
        }
    }
}

// dW[type], davec[type] of every job = the sums of that job's and type's workgroup partials, in workgroup order
__global__ __launch_bounds__(256) void transform_dw_batch_reduce_kernel(const DwBatchArgs a) {
    constexpr int wsz = 64 * 64 + 64, dwsz = 64 * 64, PER_JOB = (2 * wsz + 63) / 64;
    __shared__ float red[256];
    const int jb = blockIdx.x / PER_JOB, blk = blockIdx.x % PER_JOB;
    const int o = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int idx = blk * 64 + o;
    const bool ok = idx < 2 * wsz;
    const int type = ok ? idx / wsz : 0, k = ok ? idx % wsz : 0;
    const int b0 = type == 0 ? 0 : a.nbu, b1 = type == 0 ? a.nbu : a.nbu + a.nbi;
    const float* work = a.work + (size_t)jb * a.nb * wsz;
    float s = 0.f;
#pragma unroll 8
    for (int b = b0 + pg; b < b1; b += 4) s += work[(size_t)b * wsz + k];
    red[threadIdx.x] = s;
    __syncthreads();
    if (pg != 0 || !ok) return;
    s = (red[o] + red[64 + o]) + (red[128 + o] + red[192 + o]);
    if (k < dwsz) a.job[jb].dW[(size_t)type * dwsz + k] = s;
    else a.job[jb].davec[(size_t)type * 64 + (k - dwsz)] = s;
}

int transform_dx_mfma(const void* dH, int dh_dtype, const float* W, void* dX, int dx_dtype, int U, int I, int din, int dout, int heads,
                      int u_lo, int u_n, int i_lo, int i_n, int x_local,
                      float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t st);   // transform_mfma.hip

constexpr bool FUSE_DX_INTO_DW = true;     // 14.7 us fused against 10.2 (dW) + 9.4 (dX) us as two launches, bitwise the same dX

template <typename TX, typename TH, int DIN, int DOUT, int HEADS>
static int launch_transform_bwd(TransBwdArgs a, float* dW, float* davec, bool drop, hipStream_t st) {
    const int tu = (a.u_n + TM - 1) / TM, ti = (a.i_n + TM - 1) / TM;
    if (tu + ti == 0) return 0;
    // bf16 storage with both halves requested: dX comes out of the dW kernel (its dH tile is already in LDS)
    const bool fuse_dx = FUSE_DX_INTO_DW && sizeof(TH) == 2 && a.dX != nullptr && dW != nullptr;
    if (a.dX == nullptr || fuse_dx) {
        // nothing here: weight gradients only, or dX fused below
    } else if constexpr (sizeof(TH) == 2) {   // bf16 storage: dX on the matrix cores
        const int rc = transform_dx_mfma(a.dH, NGACF_BF16, a.W, a.dX, sizeof(TX) == 2 ? NGACF_BF16 : NGACF_F32, a.U, a.I, DIN, DOUT, HEADS, a.u_lo, a.u_n, a.i_lo, a.i_n, a.x_local,
                                         a.drop_p, a.seed, a.seed_dev, st);
        if (rc != 0) return rc;
    } else {   // dX
        constexpr size_t lds = (size_t)(TM * (DOUT + 4) + DOUT * (DIN + 4)) * sizeof(float);
        if (drop) {
            auto k = transform_bwd_dx_kernel<TX, DIN, DOUT, true>;
            if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(tu + ti), dim3(256), lds, st, a);
        } else {
            auto k = transform_bwd_dx_kernel<TX, DIN, DOUT, false>;
            if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(tu + ti), dim3(256), lds, st, a);
        }
        NGACF_CHECK_LAUNCH();
    }
    if (dW != nullptr) {   // dW, davec
        int nb = NGACF_TBWD_BLOCKS;
        if constexpr (sizeof(TH) == 2) {
            nb = 384;          // alone, 256 / 512 / 1024 workgroups measured 10.2+4.4 / 9.4+5.5 / 10.1+9.9 us (kernel + reduce); inside the
                               // train step (two chains sharing the CUs) 256 / 384 / 512 / 768 / 1024 -> 0.680 / 0.670 / 0.674 / 0.695 / 0.704 ms per step
        }
        if (tu + ti < nb) { a.nbu = tu; a.nbi = ti; }
        else {
            a.nbu = (int)((long long)nb * tu / (tu + ti));
            if (tu > 0 && a.nbu == 0) a.nbu = 1;
            if (ti > 0 && a.nbu == nb) a.nbu = nb - 1;
            a.nbi = ti > 0 ? nb - a.nbu : 0;
            if (tu == 0) a.nbu = 0;
        }
        constexpr size_t lds = (size_t)(TM * (DIN + 4) + TM * (DOUT + 4)) * sizeof(float);
        if constexpr (sizeof(TH) == 2) {
            constexpr size_t lm = (size_t)64 * (DIN * 2 + 64) + (size_t)64 * (DOUT * 2 + 64);
            static_assert(lm >= 256 * 8 * sizeof(float), "davec fold reuses the tile image");
            constexpr size_t lx = lm + (size_t)(DOUT / 16) * (DIN / 32) * 64 * 16;      // + W^T fragments of the fused dX
            if (fuse_dx) {
                auto k = drop ? transform_dw_mfma_kernel<TX, DIN, DOUT, HEADS, true, true> : transform_dw_mfma_kernel<TX, DIN, DOUT, HEADS, false, true>;
                if (lx > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lx);
                hipLaunchKernelGGL(k, dim3(a.nbu + a.nbi), dim3(256), lx, st, a);
            } else if (drop) hipLaunchKernelGGL((transform_dw_mfma_kernel<TX, DIN, DOUT, HEADS, true, false>), dim3(a.nbu + a.nbi), dim3(256), lm, st, a);
            else             hipLaunchKernelGGL((transform_dw_mfma_kernel<TX, DIN, DOUT, HEADS, false, false>), dim3(a.nbu + a.nbi), dim3(256), lm, st, a);
        } else if (drop) {
            auto k = transform_bwd_dw_kernel<TX, TH, DIN, DOUT, HEADS, true>;
            if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(a.nbu + a.nbi), dim3(256), lds, st, a);
        } else {
            auto k = transform_bwd_dw_kernel<TX, TH, DIN, DOUT, HEADS, false>;
            if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(a.nbu + a.nbi), dim3(256), lds, st, a);
        }
        NGACF_CHECK_LAUNCH();
        constexpr int wsz = DIN * DOUT + DOUT;
        hipLaunchKernelGGL(transform_bwd_reduce_kernel, dim3((2 * wsz + 63) / 64), dim3(256), 0, st,
                           a.work, dW, davec, a.nbu, a.nbi, wsz, DIN * DOUT);
        NGACF_CHECK_LAUNCH();
    }
    return 0;
}

template <typename TX, typename TH>
static int dispatch_transform_bwd(const TransBwdArgs& a, float* dW, float* davec, int din, int dout, int heads, bool drop, hipStream_t st) {
    if (din == 64 && dout == 64 && heads == 8) return launch_transform_bwd<TX, TH, 64, 64, 8>(a, dW, davec, drop, st);
    if (din == 64 && dout == 64 && heads == 1) return launch_transform_bwd<TX, TH, 64, 64, 1>(a, dW, davec, drop, st);
    if (din == 128 && dout == 64 && heads == 8) return launch_transform_bwd<TX, TH, 128, 64, 8>(a, dW, davec, drop, st);
    if (din == 128 && dout == 64 && heads == 1) return launch_transform_bwd<TX, TH, 128, 64, 1>(a, dW, davec, drop, st);
    if (din == 64 && dout == 128 && heads == 1) return launch_transform_bwd<TX, TH, 64, 128, 1>(a, dW, davec, drop, st);
    if (din == 64 && dout == 128 && heads == 16) return launch_transform_bwd<TX, TH, 64, 128, 16>(a, dW, davec, drop, st);
    if (din == 128 && dout == 128 && heads == 1) return launch_transform_bwd<TX, TH, 128, 128, 1>(a, dW, davec, drop, st);
    if (din == 128 && dout == 128 && heads == 16) return launch_transform_bwd<TX, TH, 128, 128, 16>(a, dW, davec, drop, st);
    return NGACF_E_SHAPE;
}

}  // namespace ngacf

using namespace ngacf;

extern "C" int ngacf_transform_bwd_rows(const void* X, int x_dtype, const float* W, const void* H, int h_dtype,
                                        const void* dH, const float* dS,
                                        void* dX, float* dW, float* davec, float* work,
                                        int U, int I, int din, int dout, int heads,
                                        int u_lo, int u_n, int i_lo, int i_n, int x_local,
                                        float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream) {
    if (U < 0 || I < 0 || u_lo < 0 || u_n < 0 || i_lo < 0 || i_n < 0 || u_lo + u_n > U || i_lo + i_n > I) return NGACF_E_ARG;
    if (!X || !W || !H || !dH || !dS) return NGACF_E_ARG;
    if (!dX && !dW) return NGACF_E_ARG;                          // one half may be left out (NULL), not both
    if (dW && (!davec || !work)) return NGACF_E_ARG;
    if (!(feat_drop_p >= 0.f && feat_drop_p < 1.f)) return NGACF_E_ARG;
    TransBwdArgs a;
    a.X = X; a.W = W; a.H = H; a.dH = dH; a.dS = dS; a.dX = dX; a.work = work; a.U = U; a.I = I;
    a.u_lo = u_lo; a.u_n = u_n; a.i_lo = i_lo; a.i_n = i_n; a.x_local = x_local;
    a.drop_thr = drop_threshold(feat_drop_p); a.drop_scale = 1.f / (1.f - feat_drop_p); a.drop_p = feat_drop_p;
    a.seed = seed; a.seed_dev = seed_dev; a.nbu = a.nbi = 0;
    const bool drop = feat_drop_p > 0.f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (u_n + i_n == 0) {                         // nothing owned: the sums are zero
        if (!dW) return 0;
        hipError_t e = hipMemsetAsync(dW, 0, sizeof(float) * 2 * din * dout, st);
        if (e == hipSuccess) e = hipMemsetAsync(davec, 0, sizeof(float) * 2 * dout, st);
        return static_cast<int>(e);
    }
    if (x_dtype == NGACF_F32 && h_dtype == NGACF_F32) return dispatch_transform_bwd<float, float>(a, dW, davec, din, dout, heads, drop, st);
    if (x_dtype == NGACF_F32 && h_dtype == NGACF_BF16) return dispatch_transform_bwd<float, bf16_t>(a, dW, davec, din, dout, heads, drop, st);
    if (x_dtype == NGACF_BF16 && h_dtype == NGACF_BF16) return dispatch_transform_bwd<bf16_t, bf16_t>(a, dW, davec, din, dout, heads, drop, st);
    if (x_dtype == NGACF_BF16 && h_dtype == NGACF_F32) return dispatch_transform_bwd<bf16_t, float>(a, dW, davec, din, dout, heads, drop, st);
    return NGACF_E_SHAPE;
}

extern "C" int ngacf_transform_bwd(const void* X, int x_dtype, const float* W, const void* H, int h_dtype,
                                   const void* dH, const float* dS,
                                   void* dX, float* dW, float* davec, float* work,
                                   int U, int I, int din, int dout, int heads,
                                   float feat_drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream) {
    return ngacf_transform_bwd_rows(X, x_dtype, W, H, h_dtype, dH, dS, dX, dW, davec, work, U, I, din, dout, heads,
                                    0, U, 0, I, 0, feat_drop_p, seed, seed_dev, stream);
}

extern "C" int ngacf_transform_dx_node(const float* dH, const float* W, float* dX, int U, int I,
                                       float feat_drop_p, uint64_t seed, const uint64_t* seed_dev,
                                       const ngacf_dx_prev* prev, void* stream) {
    if (U < 0 || I < 0 || !dH || !W || !dX || !prev) return NGACF_E_ARG;
    if (!prev->out || !prev->agg || !prev->R || !prev->S || !prev->Gn || !prev->ST || !prev->dS) return NGACF_E_ARG;
    if (!(feat_drop_p >= 0.f && feat_drop_p < 1.f) || !(prev->alpha > 0.f && prev->alpha < 1.f)) return NGACF_E_ARG;
    if (U + I == 0) return 0;
    TransBwdArgs a{};
    a.W = W; a.dH = dH; a.dX = dX; a.U = U; a.I = I; a.u_lo = 0; a.u_n = U; a.i_lo = 0; a.i_n = I; a.x_local = 0;
    a.drop_p = feat_drop_p; a.drop_thr = feat_drop_p > 0.f ? drop_threshold(feat_drop_p) : 0u; a.drop_scale = 1.f / (1.f - feat_drop_p);
    a.seed = seed; a.seed_dev = seed_dev;
    a.pv_out = static_cast<const float*>(prev->out); a.pv_agg = static_cast<const float*>(prev->agg); a.pv_R = prev->R; a.pv_S = prev->S;
    a.pv_Gn = static_cast<float*>(prev->Gn); a.pv_ST = reinterpret_cast<float2*>(prev->ST); a.pv_dS = prev->dS;
    a.pv_alpha = prev->alpha; a.pv_flags = prev->flags;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tu = (U + TM - 1) / TM, ti = (I + TM - 1) / TM;
    constexpr size_t lds = (size_t)(TM * (64 + 4) + 64 * (64 + 4)) * sizeof(float);
    if (feat_drop_p > 0.f) hipLaunchKernelGGL((transform_bwd_dx_kernel<float, 64, 64, true, true>), dim3(tu + ti), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((transform_bwd_dx_kernel<float, 64, 64, false, true>), dim3(tu + ti), dim3(256), lds, st, a);
    NGACF_CHECK_LAUNCH();
    return 0;
}

extern "C" int ngacf_transform_dw_batch(const ngacf_dw_job* jobs, int njobs, float* work, int work_blocks,
                                        int U, int I, const uint64_t* seed_dev, void* stream) {
    if (njobs < 0 || njobs > NGACF_DW_BATCH_MAX || U < 0 || I < 0) return NGACF_E_ARG;
    if (njobs == 0) return 0;
    if (!jobs || !work || work_blocks < 2) return NGACF_E_ARG;
    DwBatchArgs a;
    for (int k = 0; k < njobs; ++k) {
        const ngacf_dw_job& j = jobs[k];
        if (!j.X || !j.H || !j.dH || !j.dS || !j.dW || !j.davec) return NGACF_E_ARG;
        if (j.din != 64 || j.dout != 64 || (j.heads != 8 && j.heads != 1)) return NGACF_E_SHAPE;
        if (j.x_dtype != NGACF_F32 && j.x_dtype != NGACF_BF16) return NGACF_E_SHAPE;
        if (j.h_dtype != NGACF_F32 && j.h_dtype != NGACF_BF16) return NGACF_E_SHAPE;
        if (j.h_dtype != jobs[0].h_dtype) return NGACF_E_SHAPE;                   // one storage type per launch
        if (j.h_dtype == NGACF_F32 && j.x_dtype != NGACF_F32) return NGACF_E_SHAPE;    // fp32 storage: every table is fp32
        if (!(j.feat_drop_p >= 0.f && j.feat_drop_p < 1.f)) return NGACF_E_ARG;
        a.job[k] = j;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (U + I == 0) {
        for (int k = 0; k < njobs; ++k) {
            hipError_t e = hipMemsetAsync(jobs[k].dW, 0, sizeof(float) * 2 * 64 * 64, st);
            if (e == hipSuccess) e = hipMemsetAsync(jobs[k].davec, 0, sizeof(float) * 2 * 64, st);
            if (e != hipSuccess) return static_cast<int>(e);
        }
        return 0;
    }
    const int tu = (U + 63) / 64, ti = (I + 63) / 64;
    int nb = work_blocks;
    if (tu + ti < nb) { a.nbu = tu; a.nbi = ti; }
    else {
        a.nbu = (int)((long long)nb * tu / (tu + ti));
        if (tu > 0 && a.nbu == 0) a.nbu = 1;
        if (ti > 0 && a.nbu == nb) a.nbu = nb - 1;
        a.nbi = ti > 0 ? nb - a.nbu : 0;
        if (tu == 0) a.nbu = 0;
    }
    a.nb = a.nbu + a.nbi; a.njobs = njobs; a.work = work; a.U = U; a.I = I; a.seed_dev = seed_dev;
    if (jobs[0].h_dtype == NGACF_F32) hipLaunchKernelGGL(transform_dw_batch_f32_kernel, dim3(njobs * a.nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(transform_dw_batch_kernel, dim3(njobs * a.nb), dim3(256), 0, st, a);
    NGACF_CHECK_LAUNCH();
    constexpr int wsz = 64 * 64 + 64;
    hipLaunchKernelGGL(transform_dw_batch_reduce_kernel, dim3(njobs * ((2 * wsz + 63) / 64)), dim3(256), 0, st, a);
    NGACF_CHECK_LAUNCH();
    return 0;
}
