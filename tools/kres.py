"""Resource usage per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks (build container, no GPU):
    hipcc ... -c file.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2> remarks.txt;  python tools/kres.py remarks.txt [filter]"""
import re
import subprocess
import sys


def main():
    t = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    blocks = t.split("Function Name: ")[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    print("VGPR AGPR scratch occ   LDS  kernel")
    for b, d in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        d = d.replace("ngacf::", "").replace("void ", "")
        if flt and not re.search(flt, d):
            continue
        print("%4d %4d %7d %3d %6d  %s" % (g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                                           g(r"LDS Size \[bytes/block\]"), d[:150]))


if __name__ == "__main__":
    main()
