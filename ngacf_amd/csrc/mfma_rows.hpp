// Row transform on the matrix cores, the pieces shared by transform_mfma.hip (the stand-alone launch) and propagate.hip (the
// same transform riding in the traversal's epilogue, ngacf_prop_fwd_next): ONE statement of the arithmetic, so the two forms
// produce the same bits.
//
//   v_mfma_f32_32x32x16_bf16 with the NODE on the lane (B operand: 16 contiguous bytes of the node's row per k-step) and the
//   output column on the accumulator registers: lane (r = l & 31, h = l >> 5) ends up with acc[t][4g + j] = column
//   32t + 8g + 4h + j of node r.
//   Weight panel = A fragments: index ((y * KS + s) * OT + t) * 64 + l, KS = din / 16, OT = dout / 32, 8 bf16 each:
//   lane l holds W[y][16 s + 8 (l >> 5) + j][32 t + (l & 31)], j = 0..7.
#pragma once
#include <stdlib.h>

#include "ngacf_common.hpp"

namespace ngacf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// One (stage, node type) weight panel, fp32 W [2][din][dout] -> bf16 A fragments (256 threads).
__device__ __forceinline__ void pack_panel_job(const float* __restrict__ W, void* F, int din, int dout, int y, int tid) {
    const int KS = din / 16, OT = dout / 32;
    const float* M = W + (size_t)y * din * dout;
    uint4* out = reinterpret_cast<uint4*>(F) + (size_t)y * KS * OT * 64;
    for (int slot = tid; slot < KS * OT * 64; slot += 256) {
        const int l = slot & 63, t = (slot >> 6) % OT, sk = (slot >> 6) / OT;
        const int col = 32 * t + (l & 31), k0 = 16 * sk + 8 * (l >> 5);
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[j] = f32_to_bf16(M[(size_t)(k0 + 2 * j) * dout + col]) | (f32_to_bf16(M[(size_t)(k0 + 2 * j + 1) * dout + col]) << 16);
        out[slot] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// The same for the transposed product dX = dH @ W^T (ngacf_prop_bwd_dx, transform_dw_mfma_kernel<WITH_DX>): the contraction
// runs over dout, the output column over din:  slot ((y * DKS + s) * DOT + t) * 64 + l = W[y][32 t + (l & 31)][16 s + 8 (l >> 5) + j].
__device__ __forceinline__ void pack_panel_job_t(const float* __restrict__ W, void* F, int din, int dout, int y, int tid) {
    const int DKS = dout / 16, DOT = din / 32;
    const float* M = W + (size_t)y * din * dout;
    uint4* out = reinterpret_cast<uint4*>(F) + (size_t)y * DKS * DOT * 64;
    for (int slot = tid; slot < DKS * DOT * 64; slot += 256) {
        const int l = slot & 63, t = (slot >> 6) % DOT, sk = (slot >> 6) / DOT;
        const int i = 32 * t + (l & 31), k0 = 16 * sk + 8 * (l >> 5);
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[j] = f32_to_bf16(M[(size_t)i * dout + k0 + 2 * j]) | (f32_to_bf16(M[(size_t)i * dout + k0 + 2 * j + 1]) << 16);
        out[slot] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Epilogue of one 32-node tile: H rows (bf16, 16-byte stores) and the per-node scores S = H . a, formed from the ROUNDED H.
//   nd: the lane's node (row of H / S), ok: the node exists; h = lane >> 5; av = avec of the tile's node type [DOUT].
//   staged: the rounded H' goes through otile -- 32 x (DOUT / 8 + 1) uint4 of LDS private to this wave, tr: the
//   lane's row in it, row_of(tile row) -> row of H or -1 -- so that the tile leaves as whole 128-byte lines (a wave instruction
//   = 8 full rows) instead of 32 x 32-byte pieces per instruction.  Same values.  It pays where H streams to HBM (15 M rows:
//   the launch goes from 4.3-4.8 to 5.0-5.6 TB/s) and costs a little where the table is cache-resident and the launch is
//   latency-bound (Gowalla: +0.7 us per launch; as a run-time branch +1.7 us even when not taken), so the host picks the
//   instantiation by table size (stage_out_pays).
inline bool stage_out_pays(uint64_t n_rows, int dout) {
    static const char* force = getenv("NGACF_STAGE_OUT");           // "0" / "1": one form everywhere (tests compare the two)
    if (force && (force[0] == '0' || force[0] == '1')) return force[0] == '1';
    return n_rows * (uint64_t)dout * 2 >= (64ull << 20);
}

template <int OT, int HEADS, int DOUT, bool staged, typename RowOf>
__device__ __forceinline__ void mfma_fwd_epilogue(const f32x16 (&acc)[OT], int nd, bool ok, int h, const float* __restrict__ av,
                                                  void* Hout, float* __restrict__ S, uint4* otile, int tr, RowOf row_of) {
    constexpr int OROW = DOUT / 8 + 1;
    float head_part[OT * 4];
    float total = 0.f;
    char* Hrow = reinterpret_cast<char*>(Hout) + (size_t)nd * DOUT * 2;
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int gp = 0; gp < 4; gp += 2) {
            uint2 pk[2];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int g = gp + gg;
                const int c = 32 * t + 8 * g + 4 * h;
                const float4 a4 = *reinterpret_cast<const float4*>(av + c);
                uint32_t b[4];
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    b[j] = f32_to_bf16(acc[t][4 * g + j]);
                    v[j] = bf16_to_f32(b[j]);
                }
                float p = v[0] * a4.x;
                p = fmaf(v[1], a4.y, p); p = fmaf(v[2], a4.z, p); p = fmaf(v[3], a4.w, p);
                head_part[t * 4 + g] = p;
                total += p;
                pk[gg] = make_uint2(b[0] | (b[1] << 16), b[2] | (b[3] << 16));
            }
            // the two half-waves hold columns 8g..8g+3 / 8g+4..8g+7 of the node: one v_permlane32_swap per dword
            // turns two 8-byte stores per lane into one 16-byte store (lower half: group gp, upper half: group gp+1)
            const auto rx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
            const uint4 piece = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            if (staged) otile[tr * OROW + 4 * t + gp + h] = piece;
            else if (ok) *reinterpret_cast<uint4*>(Hrow + 64 * t + 16 * gp + 16 * h) = piece;
        }
    if (staged) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < DOUT / 16; ++j) {                  // 32 rows x DOUT bf16 = DOUT / 16 wave stores of 1 KB
            const int o = j * 1024 + lane * 16;
            const int orow = o / (DOUT * 2), cb = o % (DOUT * 2);
            const uint4 v = otile[orow * OROW + cb / 16];
            const int g = row_of(orow);
            if (g >= 0) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Hout) + (size_t)g * DOUT * 2 + cb) = v;
        }
    }
    if (HEADS == 1) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(total), __float_as_uint(total), false, false);
        total += __uint_as_float(h ? r[0] : r[1]);
        if (ok && h == 0) S[nd] = total;
    } else {                                   // 8-wide heads: head 4t+g = this lane's 4 columns + the other half's
        float p[OT * 4];
#pragma unroll
        for (int q = 0; q < OT * 4; ++q) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(head_part[q]), __float_as_uint(head_part[q]), false, false);
            p[q] = head_part[q] + __uint_as_float(h ? r[0] : r[1]);
        }
        // half 0 stores heads 0..3 (+8..11), half 1 heads 4..7 (+12..15): one 16-byte store per lane and group
#pragma unroll
        for (int q0 = 0; q0 < OT * 4; q0 += 8) {
            const float4 v = h == 0 ? make_float4(p[q0], p[q0 + 1], p[q0 + 2], p[q0 + 3])
                                    : make_float4(p[q0 + 4], p[q0 + 5], p[q0 + 6], p[q0 + 7]);
            if (ok) *reinterpret_cast<float4*>(S + (size_t)nd * HEADS + q0 + 4 * h) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same row transform in EXACT fp32 on v_mfma_f32_16x16x4_f32 (its accumulation is an fmaf chain, k ascending), for rows of 64
// fp32 columns: one workgroup (4 waves) and a tile of 16 rows.  Shared by the traversal epilogue (propagate.hip) and the
// stand-alone launch (transform.hip).
//   tile   [16 rows][64 fp32], row stride 68 floats: the B-operand reads (node n, feature 4t + g) hit 64 distinct banks;
//   panel  W'[type] as it stands, [64][NEXT] fp32, brought in by LDS-DMA with the 16-float column blocks of row k XOR-ed by
//          (k & 3) on the SOURCE side, so that the A-operand reads (four rows k = 4t + g of one column) spread over four bank
//          groups instead of one;
//   wave w owns the output columns 16 w .. +15 (and 64 + 16 w .. for NEXT = 128) of all 16 rows: 16 MFMAs per block, complete
//   k chains; lane (n = l & 15, g = l >> 4) ends up with columns 16 cb + 4 g .. +3 of row n: one 16-byte store of H'.
//   S': 4-column partial dot products, then a pairing tree (lane pairs, quads, and for a one-head stage the four column blocks,
//   which meet through LDS).
// ---------------------------------------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int XROW_F32 = 68;

// W [64][NEXT] fp32 (global) -> the swizzled LDS image; a 1 KB LDS-DMA piece = 256 consecutive floats of the image, the lane's 16
// bytes come from the column block the swizzle names.  256 threads; complete at the next barrier.
template <int NEXT>
__device__ __forceinline__ void load_panel_f32(float* panel_lds, const float* __restrict__ src) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int piece = threadIdx.x >> 6; piece < 64 * NEXT / 256; piece += 4) {
        const int p = piece * 256 + lane * 4;                // float position in the LDS image: row k, columns cc .. cc+3
        const int k = p / NEXT, cc = p % NEXT;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * NEXT + (cc ^ (16 * (k & 3)))),
                                         (__attribute__((address_space(3))) void*)(panel_lds + piece * 256), 16, 0, 0);
    }
}

// The tile step: tile, row numbers (xmeta[n] = row of H or -1) and panel are in LDS (a barrier lies behind them).  NONE: the stage
// has ONE head (sx: 64 floats of LDS; one more barrier inside).
template <int NEXT, bool NONE>
__device__ __forceinline__ void f32_tile_step(const float* xtile, const int* xmeta, const float* panel, const float* av, float* sx,
                                              void* Hout, float* __restrict__ S) {
    constexpr int CB = NEXT / 64;                        // column blocks of 16 per wave
    constexpr int NH = NONE ? 1 : NEXT / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int nrow = xmeta[n];
    const bool ok = nrow >= 0;
    f32x4_t acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int t = 0; t < 64 / 4; ++t) {
        const int k = 4 * t + g;
        const float xb = xtile[n * XROW_F32 + k];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int col = 64 * c + 16 * wave + n;
            const float wa = panel[k * NEXT + (col ^ (16 * g))];            // (k & 3) == g: the panel's source-side swizzle
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xb, acc[c], 0, 0, 0);
        }
    }
    float part[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int c0 = 64 * c + 16 * wave + 4 * g;
        const float4 a4 = *reinterpret_cast<const float4*>(av + c0);
        float p = 0.f;
        p = fmaf(acc[c][0], a4.x, p); p = fmaf(acc[c][1], a4.y, p); p = fmaf(acc[c][2], a4.z, p); p = fmaf(acc[c][3], a4.w, p);
        part[c] = p;
        if (ok) *reinterpret_cast<float4*>(static_cast<float*>(Hout) + (size_t)nrow * NEXT + c0) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    }
    auto x16 = [&](float v) { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                              return __uint_as_float((g & 1) ? r[0] : r[1]); };
    auto x32 = [&](float v) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                              return __uint_as_float((g & 2) ? r[0] : r[1]); };
    if constexpr (!NONE) {                                    // 8-wide heads: this lane's 4 columns + the 4 of the lane 16 away
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const float hsum = part[c] + x16(part[c]);
            if (ok && (g & 1) == 0) S[(size_t)nrow * NH + 8 * c + 2 * wave + (g >> 1)] = hsum;
        }
    } else {
        // one head: a thread's 4-column partials of the 64-column groups first, then a tree over 16 lanes' worth (pairs, quads,
        // eights, sixteen) -- pairs and quads across the lanes, the four column blocks of a group (= the four waves) through LDS
        float p = part[0];
#pragma unroll
        for (int c = 1; c < CB; ++c) p += part[c];
        p = p + x16(p);
        p = p + x32(p);
        if (g == 0) sx[wave * 16 + n] = p;
        __syncthreads();
        if (wave == 0 && g == 0 && ok) S[nrow] = (sx[n] + sx[16 + n]) + (sx[32 + n] + sx[48 + n]);
    }
}

}  // namespace ngacf
