"""Times the stand-alone forward transform (ngacf_transform_fwd, bf16 storage) at Gowalla size and at 15 M rows, fp32 and bf16
input rows: kernel time from the dispatch timestamps, algorithmic bytes per second.  python tools/transform_probe.py"""
import sys, json, torch
sys.path.insert(0, ".")
from ngacf_amd import ops
dev = torch.device("cuda:0")
for (U, I) in ((29858, 40981), (10_000_000, 5_000_000)):
    for xdt in (torch.float32, torch.bfloat16):
        N = U + I
        X = torch.randn(N, 64, device=dev).to(xdt)
        W = torch.randn(2, 64, 64, device=dev) * 0.1
        a = torch.randn(2, 64, device=dev)
        for _ in range(3):
            H, S = ops.transform_fwd(X, W, a, U, I, 8, h_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        with ops.KernelTimer("t") as kt:
            for _ in range(20):
                H, S = ops.transform_fwd(X, W, a, U, I, 8, h_dtype=torch.bfloat16)
            torch.cuda.synchronize()
        d = kt.durations_ms(ops.KernelTimer.TAG_TRANSFORM_FWD)
        byt = N * (64 * X.element_size() + 128 + 32)
        print(json.dumps({"N": N, "x": str(xdt), "us": round(1e3 * sum(d) / len(d), 2), "GBps": round(byt / (sum(d) / len(d) * 1e-3) / 1e9, 1)}))
        del X, H, S
