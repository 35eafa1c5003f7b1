// Static-weight propagation (GPLayer, graphattention/SPUIGACF.py:178-185).
#include "ngacf_common.hpp"

namespace ngacf {

constexpr int UNROLL = 4;
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
    // 256 columns: the reference's default embedSize of the dense-attention models (GACFmodel1.py:43, GACFmodel6.py:47): a row = a whole wave (fp32)
    if (dtype == NGACF_F32 && d == 256) return launch_spmm<float, 256>(segs, nseg, split, nsplit, col, val, X, out, part_acc, self_loop, st);
    if (dtype == NGACF_BF16 && d == 256) return launch_spmm<bf16_t, 256>(segs, nseg, split, nsplit, col, val, X, out, part_acc, self_loop, st);
    return NGACF_E_SHAPE;
}
