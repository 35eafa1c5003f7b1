"""One replayed train step as a timeline: every kernel of the step with its queue, start and end (us from the previous step's
optimizer launch), from a rocprofv3 kernel trace of tools/train_profile.py.
    python tools/step_timeline.py gpurun_out/trainprof/t_kernel_trace.csv [> profiles/rNN_trainstep_timeline.txt]"""
import csv
import re
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    i0, i1 = adam[-3], adam[-2]
    t0 = int(rows[i0]["End_Timestamp"])

    def short(n):
        n = re.sub(r"void |ngacf::", "", n)
        m = re.match(r"([a-z_A-Z0-9:]+)(<[^>]*>)?", n)
        return (m.group(1) + (m.group(2) or ""))[:70] if m else n[:70]      # (RCCL's kernels: names of another shape)
    busy = 0.0
    for r in rows[i0 + 1:i1 + 1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        busy += (e - s) / 1e3
        print("q%s %8.1f -> %8.1f (%6.1f)  %s" % (r["Queue_Id"], s / 1e3, e / 1e3, (e - s) / 1e3, short(r["Kernel_Name"])))
    print("step: %.1f us from optimizer to optimizer, %.1f us of kernel time" % ((int(rows[i1]["End_Timestamp"]) - t0) / 1e3, busy))


if __name__ == "__main__":
    main(sys.argv[1])
