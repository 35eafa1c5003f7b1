// Adam step of the BPR train loop (train_eval_Gowalla.py:137-138: optim.step() of torch.optim.Adam with L2 weight decay,
// run_Gowalla.py:114) for ALL parameter tensors in one launch.  Pure streaming: per element it reads p, g, m, v and writes
// p, m, v (28 B) -- 126 MB per step at Gowalla scale, i.e. bound by the HBM/fabric copy rate; a lane moves 16 bytes of each
// of the four streams, four such groups in flight.
//
// Arithmetic = torch's fused Adam (fp32 opmath, bias corrections in double from the per-tensor step counter):
//     g (+= g2) += wd * p;  m += (1 - b1) (g - m);  v = b2 v + (1 - b2) g g
//     p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "ngacf_common.hpp"

namespace ngacf {

struct AdamTable {
    float* p[NGACF_ADAM_MAX_TENSORS];
    const float* g[NGACF_ADAM_MAX_TENSORS];
    const float* g2[NGACF_ADAM_MAX_TENSORS];       // optional second gradient of the tensor (NULL: none): g = g + g2
    float* m[NGACF_ADAM_MAX_TENSORS];
    float* v[NGACF_ADAM_MAX_TENSORS];
    const float* step[NGACF_ADAM_MAX_TENSORS];     // the step count t AFTER this update (the caller increments first)
    long long n[NGACF_ADAM_MAX_TENSORS];
    int first_block[NGACF_ADAM_MAX_TENSORS + 1];   // workgroup range of every tensor
    int count;
    float lr, b1, b2, eps, wd;
};

constexpr int ADAM_ILP = 4;                        // float4 groups per lane
constexpr int ADAM_CHUNK = 256 * 4 * ADAM_ILP;     // elements per workgroup

__global__ __launch_bounds__(256) void adam_kernel(const AdamTable a) {
    int t = 0;
#pragma unroll 1
    while (t + 1 < a.count && (int)blockIdx.x >= a.first_block[t + 1]) ++t;
    __shared__ float corr[2];
    if (threadIdx.x == 0) {
        const double st = (double)*a.step[t];
        corr[0] = (float)((double)a.lr / (1.0 - pow((double)a.b1, st)));
        corr[1] = (float)sqrt(1.0 - pow((double)a.b2, st));
    }
    __syncthreads();
    const float step_size = corr[0], bc2_sqrt = corr[1];
    float* __restrict__ P = a.p[t];
    const float* __restrict__ G = a.g[t];
    const float* __restrict__ G2 = a.g2[t];
    float* __restrict__ M = a.m[t];
    float* __restrict__ V = a.v[t];
    const long long n = a.n[t];
    const long long base = (long long)((int)blockIdx.x - a.first_block[t]) * ADAM_CHUNK + threadIdx.x * 4;
    const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2;
    float4 p[ADAM_ILP], g[ADAM_ILP], m[ADAM_ILP], v[ADAM_ILP];
    bool full[ADAM_ILP];
#pragma unroll
    for (int k = 0; k < ADAM_ILP; ++k) {
        const long long e = base + (long long)k * 1024;
        full[k] = e + 4 <= n;
        if (full[k]) {
            p[k] = *reinterpret_cast<const float4*>(P + e); g[k] = *reinterpret_cast<const float4*>(G + e);
            m[k] = *reinterpret_cast<const float4*>(M + e); v[k] = *reinterpret_cast<const float4*>(V + e);
        }
    }
    if (G2) {                                      // the two propagations of a BPR step hand in their gradients separately
#pragma unroll
        for (int k = 0; k < ADAM_ILP; ++k) {
            if (!full[k]) continue;
            const float4 h = *reinterpret_cast<const float4*>(G2 + base + (long long)k * 1024);
            g[k].x += h.x; g[k].y += h.y; g[k].z += h.z; g[k].w += h.w;
        }
    }
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg = fmaf(a.wd, pp, gg);
        mm = fmaf(omb1, gg - mm, mm);
        vv = fmaf(omb2 * gg, gg, a.b2 * vv);
        pp -= step_size * mm / (sqrtf(vv) / bc2_sqrt + a.eps);
    };
#pragma unroll
    for (int k = 0; k < ADAM_ILP; ++k) {
        const long long e = base + (long long)k * 1024;
        if (full[k]) {
            upd(p[k].x, g[k].x, m[k].x, v[k].x); upd(p[k].y, g[k].y, m[k].y, v[k].y);
            upd(p[k].z, g[k].z, m[k].z, v[k].z); upd(p[k].w, g[k].w, m[k].w, v[k].w);
            *reinterpret_cast<float4*>(P + e) = p[k]; *reinterpret_cast<float4*>(M + e) = m[k]; *reinterpret_cast<float4*>(V + e) = v[k];
        } else {
            for (long long j = e; j < n && j < e + 4; ++j) {       // ragged tail of a tensor whose size is not a multiple of 4
                float pp = P[j], mm = M[j], vv = V[j];
                upd(pp, G2 ? G[j] + G2[j] : G[j], mm, vv);
                P[j] = pp; M[j] = mm; V[j] = vv;
            }
        }
    }
}

// The per-step scalar bookkeeping of a train step in ONE launch: int64 counters += their strides (dropout seed counters,
// the sampler's row offset and seed) and float counters += 1 (Adam's step counts) -- each a one-element torch add otherwise,
// 4-5 us apiece on the serial part of the step.
struct TickTable {
    long long* ic[NGACF_TICK_MAX];
    long long istride[NGACF_TICK_MAX];
    float* fc[NGACF_TICK_MAX];
    int ni, nf;
};
__global__ void tick_kernel(const TickTable t) {
    const int k = threadIdx.x;
    if (k < t.ni) *reinterpret_cast<unsigned long long*>(t.ic[k]) += (unsigned long long)t.istride[k];   // (wraps, like torch's add_)
    if (k < t.nf) *t.fc[k] += 1.f;
}

}  // namespace ngacf

using namespace ngacf;

extern "C" int ngacf_adam_step(float* const* params, const float* const* grads, const float* const* grads2, float* const* exp_avg, float* const* exp_avg_sq,
                               const float* const* steps, const int64_t* numel, int count,
                               float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (count < 0 || count > NGACF_ADAM_MAX_TENSORS) return NGACF_E_ARG;
    if (count == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !steps || !numel) return NGACF_E_ARG;
    if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f) || !(lr >= 0.f) || !(weight_decay >= 0.f)) return NGACF_E_ARG;
    AdamTable a;
    long long blocks = 0;
    for (int t = 0; t < count; ++t) {
        if (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t] || !steps[t] || numel[t] < 0) return NGACF_E_ARG;
        if ((reinterpret_cast<uintptr_t>(params[t]) | reinterpret_cast<uintptr_t>(grads[t]) | reinterpret_cast<uintptr_t>(exp_avg[t]) |
             reinterpret_cast<uintptr_t>(exp_avg_sq[t])) & 15) return NGACF_E_ARG;                      // 16-byte row slices
        a.g2[t] = grads2 ? grads2[t] : nullptr;
        if (reinterpret_cast<uintptr_t>(a.g2[t]) & 15) return NGACF_E_ARG;
        a.p[t] = params[t]; a.g[t] = grads[t]; a.m[t] = exp_avg[t]; a.v[t] = exp_avg_sq[t]; a.step[t] = steps[t]; a.n[t] = numel[t];
        a.first_block[t] = (int)blocks;
        blocks += (numel[t] + ADAM_CHUNK - 1) / ADAM_CHUNK;
        if (blocks > 0x7fffffffLL) return NGACF_E_SHAPE;
    }
    a.first_block[count] = (int)blocks;
    a.count = count; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    NGACF_CHECK_LAUNCH();
    return 0;
}

extern "C" int ngacf_tick(int64_t* const* int_counters, const int64_t* int_strides, int n_int, float* const* float_counters, int n_float,
                          void* stream) {
    if (n_int < 0 || n_float < 0 || n_int > NGACF_TICK_MAX || n_float > NGACF_TICK_MAX) return NGACF_E_ARG;
    if (n_int + n_float == 0) return 0;
    if ((n_int && (!int_counters || !int_strides)) || (n_float && !float_counters)) return NGACF_E_ARG;
    TickTable t;
    t.ni = n_int; t.nf = n_float;
    for (int k = 0; k < n_int; ++k) {
        if (!int_counters[k]) return NGACF_E_ARG;
        t.ic[k] = reinterpret_cast<long long*>(int_counters[k]); t.istride[k] = int_strides[k];
    }
    for (int k = 0; k < n_float; ++k) {
        if (!float_counters[k]) return NGACF_E_ARG;
        t.fc[k] = float_counters[k];
    }
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), t);
    NGACF_CHECK_LAUNCH();
    return 0;
}
