"""VGPRs / scratch / occupancy of the kernels of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kernel_regs.py gcr.hip build_lean_kernel [--rev HEAD]"""
import re
import subprocess
import sys
import tempfile
import os

CS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mgpreconditionedgcr_amd", "csrc")


def main():
    fn, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    src = os.path.join(CS, fn)
    if "--rev" in sys.argv:
        rev = sys.argv[sys.argv.index("--rev") + 1]
        text = subprocess.check_output(["git", "show", "%s:mgpreconditionedgcr_amd/csrc/%s" % (rev, fn)], cwd=CS)
        src = os.path.join(CS, "_rev_" + fn)
        open(src, "wb").write(text)
    try:
        with tempfile.TemporaryDirectory() as td:
            out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                                  "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(td, "x.o")],
                                 capture_output=True, text=True, cwd=CS).stderr
    finally:
        if "--rev" in sys.argv:
            os.unlink(src)
    cur = None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.check_output(["c++filt", m.group(1)], text=True).strip()
            cur = re.sub(r"\(.*", "", cur).replace("void mgcr::", "")
            vals = {}
        for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur:
                vals[key] = int(m.group(1))
                if key.startswith("LDS") and pat in cur:
                    print("%-60s vgpr %3d  scratch %3d  occupancy %d  lds %d" % (cur, vals.get("VGPRs", -1), vals.get("ScratchSize [bytes/lane]", -1),
                                                                         vals.get("Occupancy [waves/SIMD]", -1), vals[key]))


if __name__ == "__main__":
    main()
