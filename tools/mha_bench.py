"""Time the dense multi-head attention kernels (csrc/mha.hip) at the reference's sizes: ml100k, all N = 2625 nodes against
all nodes under the adjacency mask (ATTLayer_mask), and users x items / items x users (ATTLayer).  Reports time, flop/s
against the dense matrix-core peak of the operand type, and exponentials/s against the v_exp_f32 issue rate -- the bound
that actually applies at d_k = 8 (32 flop per score against one exp + ~10 VALU of scale/mask/max/sum).
    python tools/mha_bench.py [--dim 64]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ngacf_amd import data as D, ops  # noqa: E402
from ngacf_amd.modules import attention_exps, attention_flops  # noqa: E402

MFMA_PEAK = {"fp32": 157.3e12, "bf16": 2.5e15}          # MI355X_MICROARCH.md: dense peaks
# v_exp_f32: 8 cycles per wave-instruction per SIMD (guide, cycle constants) -> 64 lanes / 8 cyc x 4 SIMD x 256 CU x 2.4 GHz
EXP_PEAK = 64 / 8 * 4 * 256 * 2.4e9


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--heads", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    u, i, _ = D.load_ml100k()
    U, I = 943, 1682
    N = U + I
    mask = torch.zeros(N, N, dtype=torch.uint8, device=dev)
    ut, it = torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev) + U
    mask[ut, it] = 1; mask[it, ut] = 1
    mask[torch.arange(N), torch.arange(N)] = 1
    mask = ops.pack_mask_bits(mask)                        # packed once, as MultiHeadAttention's cache does
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(N, a.dim, device=dev, generator=g)
    G = torch.randn(N, a.dim, device=dev, generator=g)
    out = []
    for name, (q, k, m) in {"all x all, adjacency mask (ATTLayer_mask)": (X, X, mask), "all x all, no mask": (X, X, None),
                            "items x users (ATTLayer)": (X[U:].contiguous(), X[:U].contiguous(), None)}.items():
        Nq, Nk = q.shape[0], k.shape[0]
        O, LSE = ops.mha_fwd(q, k, k, a.heads, m)
        tf = timeit(lambda: ops.mha_fwd(q, k, k, a.heads, m))
        tb = timeit(lambda: ops.mha_bwd(q, k, k, O, LSE, G[:Nq].contiguous(), a.heads, m))
        fl, ex = attention_flops(Nq, Nk, a.dim), attention_exps(Nq, Nk, a.heads)
        peak = MFMA_PEAK["fp32"]
        out.append({"case": name, "Nq": Nq, "Nk": Nk, "dim": a.dim, "heads": a.heads, "operands": "fp32",
                    "fwd_us": round(tf * 1e6, 1), "bwd_us": round(tb * 1e6, 1),
                    "fwd_TFLOPs": round(fl / tf / 1e12, 2), "fwd_frac_of_mfma_peak": round(fl / tf / peak, 5),
                    "fwd_Gexp_per_s": round(ex / tf / 1e9, 1), "fwd_frac_of_exp_issue_peak": round(ex / tf / EXP_PEAK, 4),
                    "bwd_TFLOPs": round(2.5 * fl / tb / 1e12, 2), "bwd_frac_of_mfma_peak": round(2.5 * fl / tb / peak, 5),
                    "reference_score_tensor_MB": round(2 * a.heads * Nq * Nk * 4 / 1e6, 1)})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
