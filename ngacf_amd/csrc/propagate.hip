// Attention propagation over the unified bipartite CSR (user rows then item rows),
// forward.  One wave64 per work segment; a lane owns a 16-byte slice of a feature row,
// so 64 lanes cover EPI = 64 / (d * sizeof(T) / 16) edges per gather instruction and every
// neighbour row is read as one contiguous 64..512-byte burst.
//
// Replaces graphattention/SPUIGACF.py:359-397 (see include/ngacf.h).
#include "ngacf_common.hpp"

namespace ngacf {

struct PropArgs {
    const int4* segs;
    int nseg;
    const int* col;
    const void* H;
    const float* S_dst;
    const float* S_src;
    const float* avec;     // [2][d]
    void* out;
    float* agg;
    float* R;
    float* part_acc;
    float* part_r;
    int U, N;
    float alpha;
    uint32_t drop_thr;
    float drop_scale;
    uint64_t seed;
    int flags;
};

// Shapes: T storage type, D row width, HEADS in {1, D/8}.
template <typename T, int D, int HEADS>
struct Shape {
    static constexpr int VEC = Slice<T>::VEC;
    static constexpr int LPR = D / VEC;            // lanes per row
    static constexpr int EPI = WAVE / LPR;         // edges per gather instruction
    static constexpr int HW = D / HEADS;           // head width in columns
    static constexpr int LPH = HW >= VEC ? HW / VEC : 1;   // lanes per head
    static_assert(HW % VEC == 0 || HW >= VEC, "a lane's slice must sit inside one head");
};

constexpr int UNROLL = 4;

template <typename T, int D, int HEADS, bool FROM_ROW, bool DROP>
__global__ __launch_bounds__(256) void prop_fwd_kernel(const PropArgs a) {
    using SH = Shape<T, D, HEADS>;
    constexpr int VEC = SH::VEC, LPR = SH::LPR, EPI = SH::EPI, LPH = SH::LPH;
    const int lane = threadIdx.x & 63;
    const int w = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= a.nseg) return;
    const int4 sg = a.segs[w];
    const int row = uniform(sg.x), beg = uniform(sg.y), end = uniform(sg.z), part = uniform(sg.w);

    const int slot = lane / LPR;                   // which of the EPI edges of a group
    const int q = lane % LPR;                      // 16-byte slice of the row
    const int head = (q * VEC) / SH::HW;
    const T* __restrict__ Hp = static_cast<const T*>(a.H);
    const bool dst_is_user = row < a.U;

    const float s_dst = a.S_dst[(size_t)row * HEADS + head];
    float av[VEC];
    if (FROM_ROW) {
        const float* ap = a.avec + (dst_is_user ? D : 0) + q * VEC;   // sources are of the other type
#pragma unroll
        for (int j = 0; j < VEC; ++j) av[j] = ap[j];
    }
    uint32_t salt_h = 0, row_key = 0;
    if (DROP) {
        salt_h = salt_of(a.seed, 1u + head);
        row_key = dst_is_user ? (uint32_t)row * RNG_K1 : (uint32_t)row * RNG_K2 + salt_h;
    }

    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    float rsum = 0.f;

    for (int base = beg; base < end; base += EPI * UNROLL) {
        int c[UNROLL];
        bool ok[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int e = base + u * EPI + slot;
            ok[u] = e < end;
            c[u] = a.col[ok[u] ? e : end - 1];
        }
        float hv[UNROLL][VEC];
        float ss[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            Slice<T>::load(Hp + (size_t)c[u] * D + q * VEC, hv[u]);
            if (!FROM_ROW) ss[u] = a.S_src[(size_t)c[u] * HEADS + head];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float s;
            if (FROM_ROW) {
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < VEC; ++j) p = fmaf(hv[u][j], av[j], p);
                s = group_sum<1, LPH>(p);
            } else {
                s = ss[u];
            }
            s += s_dst;
            const float lr = s > 0.f ? s : a.alpha * s;
            float ev = ok[u] ? __expf(-lr) : 0.f;
            rsum += ev;
            if (DROP) {
                const uint32_t other = dst_is_user ? (uint32_t)c[u] * RNG_K2 + salt_h : (uint32_t)c[u] * RNG_K1;
                const uint32_t bits = lowbias32(row_key ^ other);
                ev = bits >= a.drop_thr ? ev * a.drop_scale : 0.f;
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = fmaf(ev, hv[u][j], acc[j]);
        }
    }
    // fold the EPI edge slots together
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
    rsum = group_sum<LPR, WAVE>(rsum);
    if (slot != 0) return;

    if (part >= 0) {            // slice of a long row: leave partial sums for prop_fwd_split_kernel
        float* pa = a.part_acc + (size_t)part * D + q * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) pa[j] = acc[j];
        if ((q * VEC) % SH::HW == 0) a.part_r[(size_t)part * HEADS + head] = rsum;
        return;
    }
    float own[VEC], o[VEC];
    Slice<T>::load(Hp + (size_t)row * D + q * VEC, own);
    const float inv = rsum > 0.f ? 1.f / rsum : 0.f;   // isolated node: aggregate is 0 (SPUIGACF.py:389)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float g = acc[j] * inv;
        acc[j] = g;
        const float pre = (a.flags & NGACF_RESIDUAL) ? own[j] + g : g;
        o[j] = (a.flags & NGACF_ELU) ? (pre > 0.f ? pre : expm1f(pre)) : pre;
    }
    Slice<T>::store(static_cast<T*>(a.out) + (size_t)row * D + q * VEC, o);
    if (a.agg) {
        float* ga = a.agg + (size_t)row * D + q * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) ga[j] = acc[j];
    }
    if (a.R && (q * VEC) % SH::HW == 0) a.R[(size_t)row * HEADS + head] = rsum;
}

// Long rows: add the partial sums of the row's slices in slot order and finish the row.
template <typename T, int D, int HEADS>
__global__ __launch_bounds__(256) void prop_fwd_split_kernel(const PropArgs a, const int* __restrict__ split, int nsplit) {
    using SH = Shape<T, D, HEADS>;
    constexpr int VEC = SH::VEC, LPR = SH::LPR;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nsplit || lane >= LPR) return;
    const int row = split[3 * w], p0 = split[3 * w + 1], np = split[3 * w + 2];
    const int q = lane, head = (q * VEC) / SH::HW;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    float rsum = 0.f;
    for (int p = p0; p < p0 + np; ++p) {
        const float* pa = a.part_acc + (size_t)p * D + q * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += pa[j];
        rsum += a.part_r[(size_t)p * HEADS + head];
    }
    float own[VEC], o[VEC];
    Slice<T>::load(static_cast<const T*>(a.H) + (size_t)row * D + q * VEC, own);
    const float inv = rsum > 0.f ? 1.f / rsum : 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float g = acc[j] * inv;
        acc[j] = g;
        const float pre = (a.flags & NGACF_RESIDUAL) ? own[j] + g : g;
        o[j] = (a.flags & NGACF_ELU) ? (pre > 0.f ? pre : expm1f(pre)) : pre;
    }
    Slice<T>::store(static_cast<T*>(a.out) + (size_t)row * D + q * VEC, o);
    if (a.agg) {
        float* ga = a.agg + (size_t)row * D + q * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) ga[j] = acc[j];
    }
    if (a.R && (q * VEC) % SH::HW == 0) a.R[(size_t)row * HEADS + head] = rsum;
}

template <typename T, int D, int HEADS>
static int launch_fwd(const PropArgs& a, const int* split, int nsplit, bool from_row, bool drop, hipStream_t st) {
    if (a.nseg > 0) {
        const dim3 grid((a.nseg + 3) / 4), block(256);
        if (from_row) {
            if (drop) hipLaunchKernelGGL((prop_fwd_kernel<T, D, HEADS, true, true>), grid, block, 0, st, a);
            else      hipLaunchKernelGGL((prop_fwd_kernel<T, D, HEADS, true, false>), grid, block, 0, st, a);
        } else {
            if (drop) hipLaunchKernelGGL((prop_fwd_kernel<T, D, HEADS, false, true>), grid, block, 0, st, a);
            else      hipLaunchKernelGGL((prop_fwd_kernel<T, D, HEADS, false, false>), grid, block, 0, st, a);
        }
        NGACF_CHECK_LAUNCH();
    }
    if (nsplit > 0) {
        hipLaunchKernelGGL((prop_fwd_split_kernel<T, D, HEADS>), dim3((nsplit + 3) / 4), dim3(256), 0, st, a, split, nsplit);
        NGACF_CHECK_LAUNCH();
    }
    return 0;
}


// ===========================================================================
// backward (pull-only, no atomics): SURVEY.md §8(a) A4b
// ===========================================================================
struct PropBwdArgs {
    const int4* segs;
    int nseg;
    const int* col;
    const void* H;
    const float* S_dst;
    const float* S_src;
    const float* avec;
    const void* out;
    const float* agg;
    const float* R;
    const float* gout;
    float* G;
    float* Gn;
    float* t;
    float* dH;
    float* dS;
    float* part_acc;
    float* part_r;
    int U, N;
    float alpha;
    uint32_t drop_thr;
    float drop_scale;
    uint64_t seed;
    int flags;
};

// 4 consecutive columns of a T row as floats
template <typename T> struct Quad;
template <> struct Quad<float> {
    static __device__ __forceinline__ void load(const float* p, float (&f)[4]) { Slice<float>::load(p, f); }
};
template <> struct Quad<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&f)[4]) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    }
};

// Step 1, per node: g = gout * ELU'(out); Gn = g / R; t[n,h] = (g . agg)_h / R.   One lane per 4 columns.
template <typename T, int D, int HEADS>
__global__ __launch_bounds__(256) void prop_bwd_node_kernel(const PropBwdArgs a) {
    constexpr int LPR = D / 4, HW = D / HEADS, LPH = HW / 4;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = gid / LPR;
    const int q = (int)(gid % LPR);
    const bool ok = n < (size_t)a.N;
    const size_t nn = ok ? n : 0;
    const int head = (q * 4) / HW;
    float g[4], o[4], ag[4];
    Slice<float>::load(a.gout + nn * D + q * 4, g);
    Slice<float>::load(a.agg + nn * D + q * 4, ag);
    if (a.flags & NGACF_ELU) {
        Quad<T>::load(static_cast<const T*>(a.out) + nn * D + q * 4, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] *= o[j] > 0.f ? 1.f : o[j] + 1.f;   // ELU' = exp(pre) = out + 1 on the negative side
    }
    const float r = a.R[nn * HEADS + head];
    const float inv = r > 0.f ? 1.f / r : 0.f;                // isolated node: the aggregate is a constant 0
    float gn[4], p = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        gn[j] = g[j] * inv;
        p = fmaf(gn[j], ag[j], p);
    }
    p = group_sum<1, LPH>(p);
    if (!ok) return;
    Slice<float>::store(a.G + n * D + q * 4, g);
    Slice<float>::store(a.Gn + n * D + q * 4, gn);
    if ((q * 4) % HW == 0) a.t[n * HEADS + head] = p;
}

// Step 2, per directed edge (pull).  One lane per 4 columns, EPI = 64 / (D/4) edges per gather instruction.
template <typename T, int D, int HEADS, bool FROM_ROW, bool DROP>
__global__ __launch_bounds__(256) void prop_bwd_kernel(const PropBwdArgs a) {
    constexpr int LPR = D / 4, EPI = WAVE / LPR, HW = D / HEADS, LPH = HW / 4;
    const int lane = threadIdx.x & 63;
    const int w = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= a.nseg) return;
    const int4 sg = a.segs[w];
    const int row = uniform(sg.x), beg = uniform(sg.y), end = uniform(sg.z), part = uniform(sg.w);
    const int slot = lane / LPR, q = lane % LPR;
    const int head = (q * 4) / HW;
    const T* __restrict__ Hp = static_cast<const T*>(a.H);
    const bool dst_is_user = row < a.U;

    float h_own[4], gn_own[4];
    Quad<T>::load(Hp + (size_t)row * D + q * 4, h_own);
    Slice<float>::load(a.Gn + (size_t)row * D + q * 4, gn_own);
    const float s_dst = a.S_dst[(size_t)row * HEADS + head];
    const float t_own = a.t[(size_t)row * HEADS + head];
    float av[4];
    if (FROM_ROW) {
        const float* ap = a.avec + (dst_is_user ? D : 0) + q * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = ap[j];
    }
    uint32_t salt_h = 0, row_key = 0;
    if (DROP) {
        salt_h = salt_of(a.seed, 1u + head);
        row_key = dst_is_user ? (uint32_t)row * RNG_K1 : (uint32_t)row * RNG_K2 + salt_h;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float dsum = 0.f;

    for (int base = beg; base < end; base += EPI * UNROLL) {
        int c[UNROLL];
        bool ok[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int e = base + u * EPI + slot;
            ok[u] = e < end;
            c[u] = a.col[ok[u] ? e : end - 1];
        }
        float hv[UNROLL][4], gv[UNROLL][4], ss[UNROLL], tc[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            Quad<T>::load(Hp + (size_t)c[u] * D + q * 4, hv[u]);
            Slice<float>::load(a.Gn + (size_t)c[u] * D + q * 4, gv[u]);
            tc[u] = a.t[(size_t)c[u] * HEADS + head];
            if (!FROM_ROW) ss[u] = a.S_src[(size_t)c[u] * HEADS + head];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float dot = 0.f, s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dot = fmaf(gn_own[j], hv[u][j], dot);
                dot = fmaf(gv[u][j], h_own[j], dot);
                if (FROM_ROW) s = fmaf(hv[u][j], av[j], s);
            }
            dot = group_sum<1, LPH>(dot);
            s = FROM_ROW ? group_sum<1, LPH>(s) : ss[u];
            s += s_dst;
            const float lr = s > 0.f ? s : a.alpha * s;
            const float ev = ok[u] ? __expf(-lr) : 0.f;
            float m = 1.f;
            if (DROP) {
                const uint32_t other = dst_is_user ? (uint32_t)c[u] * RNG_K2 + salt_h : (uint32_t)c[u] * RNG_K1;
                m = lowbias32(row_key ^ other) >= a.drop_thr ? a.drop_scale : 0.f;
            }
            const float de = m * dot - t_own - tc[u];
            dsum -= ev * de * (s > 0.f ? 1.f : a.alpha);
            const float wgt = m * ev;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(wgt, gv[u][j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
    dsum = group_sum<LPR, WAVE>(dsum);
    if (slot != 0) return;

    if (part >= 0) {
        Slice<float>::store(a.part_acc + (size_t)part * D + q * 4, acc);
        if ((q * 4) % HW == 0) a.part_r[(size_t)part * HEADS + head] = dsum;
        return;
    }
    float g[4], o[4];
    if (a.flags & NGACF_RESIDUAL) Slice<float>::load(a.G + (size_t)row * D + q * 4, g);
    else g[0] = g[1] = g[2] = g[3] = 0.f;
    const float* ao = a.avec + (dst_is_user ? 0 : D) + q * 4;      // the node's own type
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = g[j] + acc[j] + dsum * ao[j];
    Slice<float>::store(a.dH + (size_t)row * D + q * 4, o);
    if ((q * 4) % HW == 0) a.dS[(size_t)row * HEADS + head] = dsum;
}

template <int D, int HEADS>
__global__ __launch_bounds__(256) void prop_bwd_split_kernel(const PropBwdArgs a, const int* __restrict__ split, int nsplit) {
    constexpr int LPR = D / 4, HW = D / HEADS;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nsplit || lane >= LPR) return;
    const int row = split[3 * w], p0 = split[3 * w + 1], np = split[3 * w + 2];
    const int q = lane, head = (q * 4) / HW;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, dsum = 0.f;
    for (int p = p0; p < p0 + np; ++p) {
        float v[4];
        Slice<float>::load(a.part_acc + (size_t)p * D + q * 4, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
        dsum += a.part_r[(size_t)p * HEADS + head];
    }
    float g[4], o[4];
    if (a.flags & NGACF_RESIDUAL) Slice<float>::load(a.G + (size_t)row * D + q * 4, g);
    else g[0] = g[1] = g[2] = g[3] = 0.f;
    const float* ao = a.avec + (row < a.U ? 0 : D) + q * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = g[j] + acc[j] + dsum * ao[j];
    Slice<float>::store(a.dH + (size_t)row * D + q * 4, o);
    if ((q * 4) % HW == 0) a.dS[(size_t)row * HEADS + head] = dsum;
}

template <typename T, int D, int HEADS>
static int launch_bwd(const PropBwdArgs& a, const int* split, int nsplit, bool from_row, bool drop, hipStream_t st) {
    {
        const size_t threads = (size_t)a.N * (D / 4);
        hipLaunchKernelGGL((prop_bwd_node_kernel<T, D, HEADS>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a);
        NGACF_CHECK_LAUNCH();
    }
    if (a.nseg > 0) {
        const dim3 grid((a.nseg + 3) / 4), block(256);
        if (from_row) {
            if (drop) hipLaunchKernelGGL((prop_bwd_kernel<T, D, HEADS, true, true>), grid, block, 0, st, a);
            else      hipLaunchKernelGGL((prop_bwd_kernel<T, D, HEADS, true, false>), grid, block, 0, st, a);
        } else {
            if (drop) hipLaunchKernelGGL((prop_bwd_kernel<T, D, HEADS, false, true>), grid, block, 0, st, a);
            else      hipLaunchKernelGGL((prop_bwd_kernel<T, D, HEADS, false, false>), grid, block, 0, st, a);
        }
        NGACF_CHECK_LAUNCH();
    }
    if (nsplit > 0) {
        hipLaunchKernelGGL((prop_bwd_split_kernel<D, HEADS>), dim3((nsplit + 3) / 4), dim3(256), 0, st, a, split, nsplit);
        NGACF_CHECK_LAUNCH();
    }
    return 0;
}

// ===========================================================================
// static-weight propagation (GPLayer, SPUIGACF.py:178-185)
// ===========================================================================
template <typename T, int D>
__global__ __launch_bounds__(256) void spmm_fwd_kernel(const int4* __restrict__ segs, int nseg, const int* __restrict__ col,
                                                       const float* __restrict__ val, const T* __restrict__ X,
                                                       T* __restrict__ out, float* __restrict__ part_acc, int self_loop) {
    constexpr int VEC = Slice<T>::VEC, LPR = D / VEC, EPI = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int w = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= nseg) return;
    const int4 sg = segs[w];
    const int row = uniform(sg.x), beg = uniform(sg.y), end = uniform(sg.z), part = uniform(sg.w);
    const int slot = lane / LPR, q = lane % LPR;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int base = beg; base < end; base += EPI * UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int e = base + u * EPI + slot;
            const bool ok = e < end;
            const int ee = ok ? e : end - 1;
            float hv[VEC];
            Slice<T>::load(X + (size_t)col[ee] * D + q * VEC, hv);
            const float wv = ok ? val[ee] : 0.f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = fmaf(wv, hv[j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = group_sum<LPR, WAVE>(acc[j]);
    if (slot != 0) return;
    if (part >= 0) {
        float* pa = part_acc + (size_t)part * D + q * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) pa[j] = acc[j];
        return;
    }
    if (self_loop) {
        float own[VEC];
        Slice<T>::load(X + (size_t)row * D + q * VEC, own);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += own[j];
    }
    Slice<T>::store(out + (size_t)row * D + q * VEC, acc);
}

template <typename T, int D>
__global__ __launch_bounds__(256) void spmm_split_kernel(const int* __restrict__ split, int nsplit, const T* __restrict__ X,
                                                         T* __restrict__ out, const float* __restrict__ part_acc, int self_loop) {
    constexpr int VEC = Slice<T>::VEC, LPR = D / VEC;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nsplit || lane >= LPR) return;
    const int row = split[3 * w], p0 = split[3 * w + 1], np = split[3 * w + 2], q = lane;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int p = p0; p < p0 + np; ++p)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += part_acc[(size_t)p * D + q * VEC + j];
    if (self_loop) {
        float own[VEC];
        Slice<T>::load(X + (size_t)row * D + q * VEC, own);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += own[j];
    }
    Slice<T>::store(out + (size_t)row * D + q * VEC, acc);
}

template <typename T, int D>
static int launch_spmm(const int32_t* segs, int nseg, const int32_t* split, int nsplit, const int32_t* col, const float* val,
                       const void* X, void* out, float* part_acc, int self_loop, hipStream_t st) {
    hipLaunchKernelGGL((spmm_fwd_kernel<T, D>), dim3((nseg + 3) / 4), dim3(256), 0, st, reinterpret_cast<const int4*>(segs), nseg,
                       col, val, static_cast<const T*>(X), static_cast<T*>(out), part_acc, self_loop);
    NGACF_CHECK_LAUNCH();
    if (nsplit > 0) {
        hipLaunchKernelGGL((spmm_split_kernel<T, D>), dim3((nsplit + 3) / 4), dim3(256), 0, st, split, nsplit,
                           static_cast<const T*>(X), static_cast<T*>(out), part_acc, self_loop);
        NGACF_CHECK_LAUNCH();
    }
    return 0;
}

}  // namespace ngacf

using namespace ngacf;

extern "C" int ngacf_prop_fwd(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                              const int32_t* col, const void* H, int dtype,
                              const float* S_dst, const float* S_src, const float* avec,
                              void* out, float* agg, float* R, float* part_acc, float* part_r,
                              int U, int N, int d, int heads, float alpha,
                              float drop_p, uint64_t seed, int flags, void* stream) {
    if (nseg < 0 || nsplit < 0 || U < 0 || N < U) return NGACF_E_ARG;
    if (nseg == 0) return 0;
    if (!segs || !col || !H || !S_dst || !out) return NGACF_E_ARG;
    const bool from_row = (flags & NGACF_SCORE_ROW) != 0;
    if (from_row ? !avec : !S_src) return NGACF_E_ARG;
    if (nsplit > 0 && (!split || !part_acc || !part_r)) return NGACF_E_ARG;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return NGACF_E_ARG;
    PropArgs a;
    a.segs = reinterpret_cast<const int4*>(segs); a.nseg = nseg; a.col = col; a.H = H;
    a.S_dst = S_dst; a.S_src = S_src; a.avec = avec; a.out = out; a.agg = agg; a.R = R;
    a.part_acc = part_acc; a.part_r = part_r; a.U = U; a.N = N; a.alpha = alpha;
    a.drop_thr = drop_threshold(drop_p); a.drop_scale = 1.f / (1.f - drop_p); a.seed = seed; a.flags = flags;
    const bool drop = drop_p > 0.f;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NGACF_CASE(T, D, HD) return launch_fwd<T, D, HD>(a, split, nsplit, from_row, drop, st)
    if (dtype == NGACF_F32) {
        if (d == 64 && heads == 8) NGACF_CASE(float, 64, 8);
        if (d == 64 && heads == 1) NGACF_CASE(float, 64, 1);
        if (d == 128 && heads == 1) NGACF_CASE(float, 128, 1);
        if (d == 128 && heads == 16) NGACF_CASE(float, 128, 16);
    } else if (dtype == NGACF_BF16) {
        if (d == 64 && heads == 8) NGACF_CASE(bf16_t, 64, 8);
        if (d == 64 && heads == 1) NGACF_CASE(bf16_t, 64, 1);
        if (d == 128 && heads == 1) NGACF_CASE(bf16_t, 128, 1);
        if (d == 128 && heads == 16) NGACF_CASE(bf16_t, 128, 16);
    }
#undef NGACF_CASE
    return NGACF_E_SHAPE;
}

extern "C" int ngacf_prop_bwd(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                              const int32_t* col, const void* H, int dtype,
                              const float* S_dst, const float* S_src, const float* avec,
                              const void* out, const float* agg, const float* R, const float* gout,
                              float* G, float* Gn, float* t, float* dH, float* dS,
                              float* part_acc, float* part_r,
                              int U, int N, int d, int heads, float alpha,
                              float drop_p, uint64_t seed, int flags, void* stream) {
    if (nseg < 0 || nsplit < 0 || U < 0 || N < U) return NGACF_E_ARG;
    if (N == 0) return 0;
    if (!segs || !col || !H || !S_dst || !avec || !out || !agg || !R || !gout || !G || !Gn || !t || !dH || !dS) return NGACF_E_ARG;
    const bool from_row = (flags & NGACF_SCORE_ROW) != 0;
    if (!from_row && !S_src) return NGACF_E_ARG;
    if (nsplit > 0 && (!split || !part_acc || !part_r)) return NGACF_E_ARG;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return NGACF_E_ARG;
    PropBwdArgs a;
    a.segs = reinterpret_cast<const int4*>(segs); a.nseg = nseg; a.col = col; a.H = H;
    a.S_dst = S_dst; a.S_src = S_src; a.avec = avec; a.out = out; a.agg = agg; a.R = R; a.gout = gout;
    a.G = G; a.Gn = Gn; a.t = t; a.dH = dH; a.dS = dS; a.part_acc = part_acc; a.part_r = part_r;
    a.U = U; a.N = N; a.alpha = alpha;
    a.drop_thr = drop_threshold(drop_p); a.drop_scale = 1.f / (1.f - drop_p); a.seed = seed; a.flags = flags;
    const bool drop = drop_p > 0.f;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NGACF_CASE(T, D, HD) return launch_bwd<T, D, HD>(a, split, nsplit, from_row, drop, st)
    if (dtype == NGACF_F32) {
        if (d == 64 && heads == 8) NGACF_CASE(float, 64, 8);
        if (d == 64 && heads == 1) NGACF_CASE(float, 64, 1);
        if (d == 128 && heads == 1) NGACF_CASE(float, 128, 1);
        if (d == 128 && heads == 16) NGACF_CASE(float, 128, 16);
    } else if (dtype == NGACF_BF16) {
        if (d == 64 && heads == 8) NGACF_CASE(bf16_t, 64, 8);
        if (d == 64 && heads == 1) NGACF_CASE(bf16_t, 64, 1);
        if (d == 128 && heads == 1) NGACF_CASE(bf16_t, 128, 1);
        if (d == 128 && heads == 16) NGACF_CASE(bf16_t, 128, 16);
    }
#undef NGACF_CASE
    return NGACF_E_SHAPE;
}

extern "C" int ngacf_spmm_fwd(const int32_t* segs, int nseg, const int32_t* split, int nsplit,
                              const int32_t* col, const float* val, const void* X, int dtype,
                              void* out, float* part_acc, int N, int d, int self_loop, void* stream) {
    if (nseg < 0 || nsplit < 0 || N < 0) return NGACF_E_ARG;
    if (nseg == 0) return 0;
    if (!segs || !col || !val || !X || !out) return NGACF_E_ARG;
    if (nsplit > 0 && (!split || !part_acc)) return NGACF_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == NGACF_F32 && d == 64) return launch_spmm<float, 64>(segs, nseg, split, nsplit, col, val, X, out, part_acc, self_loop, st);
    if (dtype == NGACF_F32 && d == 128) return launch_spmm<float, 128>(segs, nseg, split, nsplit, col, val, X, out, part_acc, self_loop, st);
    if (dtype == NGACF_BF16 && d == 64) return launch_spmm<bf16_t, 64>(segs, nseg, split, nsplit, col, val, X, out, part_acc, self_loop, st);
    if (dtype == NGACF_BF16 && d == 128) return launch_spmm<bf16_t, 128>(segs, nseg, split, nsplit, col, val, X, out, part_acc, self_loop, st);
    return NGACF_E_SHAPE;
}
