"""Scan the gfx950 ISA of the HIP sources for the two compiler habits that drain an LDS-DMA pipeline:
scratch (spill) traffic inside a loop that contains MFMAs, and `s_waitcnt vmcnt(N)` the COMPILER put there (the kernels'
own counted waits sit inside inline-asm blocks and are not listed).

    python tools/isa_scan.py [file.hip ...]        (default: every vision_ft_amd/csrc/*.hip; needs hipcc, no GPU)

Per kernel: registers, spilled registers, and for every INNERMOST loop with at least 8 MFMAs the number of scratch instructions and the
compiler's vmcnt waits.  A register-staged kernel (global load -> registers -> LDS, one tile ahead) shows one `vmcnt(0)` per
iteration by design; anything else in a kernel that keeps DMAs in flight across its barriers is a finding."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only", "-S", "--cuda-device-only"]


def scan(asm_path: str) -> None:
    kern, in_asm, cur = None, False, None
    loops: dict = {}
    meta: dict = {}
    name = None
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, cur = m.group(1), None
            loops[kern] = {}
            continue
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count):\s+(\d+)", line)
        if m and name:
            meta.setdefault(name, {})[m.group(1)] = int(m.group(2))
        if kern is None:
            continue
        if re.match(r"^\.LBB\d+_\d+:", line) or line.startswith("; %bb."):
            m3 = re.search(r"Loop Header: Depth=(\d+)", line)
            m2 = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", line)
            if m3:
                cur = (line.split(":")[0].lstrip(".L"), int(m3.group(1)))
            elif m2:
                cur = (m2.group(1), int(m2.group(2)))
            else:
                cur = None
        if "ASMSTART" in line:
            in_asm = True
        if "ASMEND" in line:
            in_asm = False
        if cur:
            d = loops[kern].setdefault(cur, {"mfma": 0, "scratch": 0, "waits": []})
            if "v_mfma" in line:
                d["mfma"] += 1
            if "scratch_" in line:
                d["scratch"] += 1
            if not in_asm and re.search(r"s_waitcnt.*vmcnt\(", line):
                d["waits"].append(re.search(r"vmcnt\((\d+)\)", line).group(1))
    for k, ls in loops.items():
        mm = meta.get(k, {})
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", k)[:70]
        head = f"{short}: vgprs {mm.get('vgpr_count', '?')} spilled {mm.get('vgpr_spill_count', '?')} (sgpr spills {mm.get('sgpr_spill_count', '?')})"
        deepest = max((lp[1] for lp in ls), default=0)  # innermost loops only: an outer (persistent-tile) loop legitimately waits
        rows = [(lp, d) for lp, d in ls.items() if lp[1] == deepest and d["mfma"] >= 8 and (d["scratch"] or d["waits"])]
        if mm.get("vgpr_spill_count", 0) or rows:
            print(head)
            for (lbl, depth), d in rows:
                print(f"    loop {lbl} depth {depth}: {d['mfma']} MFMAs, {d['scratch']} scratch instructions, compiler vmcnt waits {d['waits'][:16]}")


def main() -> None:
    srcs = [os.path.abspath(a) for a in sys.argv[1:]] or sorted(glob.glob(os.path.join(ROOT, "vision_ft_amd", "csrc", "*.hip")))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        for src in srcs:
            out = os.path.join(tmp, os.path.basename(src) + ".s")
            r = subprocess.run([hipcc, *FLAGS, src, "-o", out], cwd=os.path.dirname(src), capture_output=True, text=True)
            if r.returncode != 0:
                print(f"{src}: hipcc failed\n{r.stderr[-2000:]}")
                continue
            print(f"== {os.path.basename(src)}")
            scan(out)


if __name__ == "__main__":
    main()
