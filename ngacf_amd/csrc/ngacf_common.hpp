// Shared device helpers for the gfx950 kernels: 16-byte row-slice I/O for fp32 and
// bf16 storage, wave64 cross-lane moves, the counter-hash dropout RNG.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/ngacf.h"

namespace ngacf {

constexpr int WAVE = 64;

// ---------------------------------------------------------------------------
// storage types.  A lane always moves 16 bytes: 4 fp32 or 8 bf16 columns.
// ---------------------------------------------------------------------------
struct bf16_t { uint16_t bits; };

template <typename T> struct Slice;

template <> struct Slice<float> {
    static constexpr int VEC = 4;
    static __device__ __forceinline__ void load(const float* p, float (&f)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&f)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    }
    static __device__ __forceinline__ float round(float x) { return x; }
};

__device__ __forceinline__ float bf16_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// round-to-nearest-even through the hardware convert (keeps NaN a NaN)
__device__ __forceinline__ uint32_t f32_to_bf16(float x) {
    const __bf16 h = static_cast<__bf16>(x);
    return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, h));
}

template <> struct Slice<bf16_t> {
    static constexpr int VEC = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&f)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j]     = __uint_as_float(w[j] << 16);
            f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = f32_to_bf16(f[2 * j]) | (f32_to_bf16(f[2 * j + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ float round(float x) { return bf16_to_f32(f32_to_bf16(x)); }
};

// ---------------------------------------------------------------------------
// wave64 cross-lane helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float xor_add(float v, int mask) { return v + __shfl_xor(v, mask, WAVE); }

// sum over the lanes that differ only in the bits [lo, hi) of the lane id
template <int LO, int HI>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = LO; m < HI; m <<= 1) v = xor_add(v, m);
    return v;
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// DPP moves run at VALU rate and need no LDS: quad_perm / row_half_mirror / row_mirror (gfx9 encodings)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int m = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(m);
}

// all-reduce (sum) over aligned groups of N consecutive lanes, N a power of two
template <int N>
__device__ __forceinline__ float lanes_sum(float v) {
    if constexpr (N >= 2) v = dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]
    if constexpr (N >= 4) v = dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
    if constexpr (N >= 8) v = dpp_add<0x141>(v);         // row_half_mirror
    if constexpr (N >= 16) v = dpp_add<0x140>(v);        // row_mirror
    if constexpr (N >= 32) v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (0x10 << 10) | 0x1f));
    if constexpr (N >= 64) v = xor_add(v, 32);
    return v;
}

// ---------------------------------------------------------------------------
// counter-hash dropout RNG (host restatement: ngacf_amd/rng.py; pinned by tests)
//   bits = lowbias32(a * K1 ^ (b * K2 + salt)),  keep iff bits >= floor(p * 2^32)
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
constexpr uint32_t RNG_K1 = 0x9E3779B1u, RNG_K2 = 0x85EBCA77u, RNG_K3 = 0xC2B2AE3Du;

__host__ __device__ __forceinline__ uint32_t salt_of(uint64_t seed, uint32_t stream) {
    return lowbias32(static_cast<uint32_t>(seed) ^ lowbias32(static_cast<uint32_t>(seed >> 32) + stream * RNG_K3));
}
// feature mask of element (node n, column c)
__host__ __device__ __forceinline__ uint32_t feat_bits(uint32_t salt, uint32_t n, uint32_t c) {
    return lowbias32((n * RNG_K1) ^ (c * RNG_K2 + salt));
}
// edge mask of (user u, item i as global id, head h); salt_h = salt_of(seed, 1 + h)
__host__ __device__ __forceinline__ uint32_t edge_bits(uint32_t salt_h, uint32_t u, uint32_t i) {
    return lowbias32((u * RNG_K1) ^ (i * RNG_K2 + salt_h));
}
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
    const double t = static_cast<double>(p) * 4294967296.0;
    return t >= 4294967295.0 ? 0xffffffffu : static_cast<uint32_t>(t);
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : (__expf(x) - 1.f); }

// Work order of one traversal launch (workgroups, in dispatch order):
//   [user hub rows][user long segments][user short rows][item hub rows][item long segments][item short rows]
// User rows gather item rows only and vice versa, so each half of the launch re-reads ONE of the two tables
// (5 MB instead of 9 MB at Gowalla scale, bf16) and that table largely stays in the 4 MB L2 of each XCD.  Hub rows
// (one workgroup per row) lead each half: the longest work starts first.
struct WorkOrder {
    int nseg, nseg_u, nshort, nshort_u, nhub, nhub_u;   // totals and the leading user parts (nshort_u is a multiple of 16)
    int b0, b1, b2, b3, b4;                             // prefix sums of the first five groups' workgroup counts
};
__host__ inline WorkOrder make_order(int nseg, int nseg_u, int nshort, int nshort_u, int nhub, int nhub_u, int epi, int* nblocks) {
    WorkOrder o;
    o.nseg = nseg; o.nseg_u = nseg_u; o.nshort = nshort; o.nshort_u = nshort_u; o.nhub = nhub; o.nhub_u = nhub_u;
    o.b0 = nhub_u;
    o.b1 = o.b0 + (nseg_u + 3) / 4;
    o.b2 = o.b1 + ((nshort_u + epi - 1) / epi + 3) / 4;
    o.b3 = o.b2 + (nhub - nhub_u);
    o.b4 = o.b3 + (nseg - nseg_u + 3) / 4;
    *nblocks = o.b4 + (((nshort - nshort_u) + epi - 1) / epi + 3) / 4;
    return o;
}
enum WorkKind { WORK_HUB = 0, WORK_LONG = 1, WORK_SHORT = 2 };
// -> kind, first item of this wave's group (hub-row index of the WORKGROUP, segment index or short-row index of the wave), end
__device__ __forceinline__ void locate(const WorkOrder& o, int block, int wave, int epi, int& kind, int& idx, int& end) {
    if (block < o.b0) { kind = WORK_HUB; idx = block; end = o.nhub_u; }
    else if (block < o.b1) { kind = WORK_LONG; idx = (block - o.b0) * 4 + wave; end = o.nseg_u; }
    else if (block < o.b2) { kind = WORK_SHORT; idx = ((block - o.b1) * 4 + wave) * epi; end = o.nshort_u; }
    else if (block < o.b3) { kind = WORK_HUB; idx = o.nhub_u + (block - o.b2); end = o.nhub; }
    else if (block < o.b4) { kind = WORK_LONG; idx = o.nseg_u + (block - o.b3) * 4 + wave; end = o.nseg; }
    else { kind = WORK_SHORT; idx = o.nshort_u + ((block - o.b4) * 4 + wave) * epi; end = o.nshort; }
}

// row flags of a scored batch: one bit per row (N / 8 bytes: the table of a 10^5-row graph sits in a CU's vector L1)
__device__ __forceinline__ bool flagged(const uint32_t* __restrict__ bits, int row) { return (bits[row >> 5] >> (row & 31)) & 1u; }

// The row-local half of the backward's dS (propagate_bwd.hip, header comment): P1 = (1 - alpha) (R (Gn . aggn)_h - Rn t) from the two
// head sums p = (Gn . agg)_h (= t) and pn = (Gn . aggn)_h; ONE explicit operation order (step 1 has three homes: prop_bwd_node_kernel,
// the edge backward's fused tail, the fp32 dX kernel's epilogue -- the same bits in all of them)
__device__ __forceinline__ float own_term(float alpha, float r, float rn, float p, float pn) {
    return (1.f - alpha) * fmaf(r, pn, -(rn * p));
}

// Kernel launch with optional dispatch-timestamp timing (timing.hip; tags: NGACF_TAG_* of include/ngacf.h).
bool timing_on();
int timing_events(hipEvent_t* a, hipEvent_t* b);
void timing_record(int tag, hipEvent_t a, hipEvent_t b);
template <typename K, typename... Args>
inline void launch_k(int tag, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    hipEvent_t ea, eb;
    if (timing_on() && timing_events(&ea, &eb) == 0) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, ea, eb, 0, args...);
        timing_record(tag, ea, eb);                     // (behind the launch: see timing.hip)
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
    }
}

#define NGACF_CHECK_LAUNCH()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return static_cast<int>(e__);   \
    } while (0)

}  // namespace ngacf
