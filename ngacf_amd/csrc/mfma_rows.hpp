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

}  // namespace ngacf
