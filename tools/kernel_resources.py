#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), no GPU
needed:   python tools/kernel_resources.py 3dfacerecon_amd/csrc/fr_decode_q.hip [substring]"""
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-mllvm",
         "-amdgpu-atomic-optimizer-strategy=None", "-Rpass-analysis=kernel-resource-usage"]


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as td:
        p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", os.path.join(td, "x.so"), src], capture_output=True, text=True)
    rows, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark: (.*)", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip().split()[0]
    names = [r["name"] for r in rows]
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    except OSError:
        dem = names
    print("%-92s %5s %5s %7s %4s %6s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "sspill", "vspill"))
    for r, d in zip(rows, dem):
        d = re.sub(r"\(.*", "", d).replace("void ", "")
        if want and want not in d:
            continue
        print("%-92s %5s %5s %7s %4s %6s %6s" % (d[:92], r.get("VGPRs", "?"), r.get("SGPRs", "?"),
                                                  r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"),
                                                  r.get("SGPRs Spill", "?"), r.get("VGPRs Spill", "?")))
    if p.returncode:
        sys.stderr.write(p.stderr[-2000:])
    return p.returncode


if __name__ == "__main__":
    sys.exit(main())
