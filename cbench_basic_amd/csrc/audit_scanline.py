#!/usr/bin/env python3
"""Build-time audit of scanline_batched_kernel (csrc/scanline.hip): its 256 resident weight registers are accumulator registers
named literally in inline asm, which is only sound while the COMPILER itself never touches the accumulator file in that kernel
(cdna_hip_programming.md 5.7 item 4).  Fails unless, for both instantiations: no VGPR spill, no scratch, every v_accvgpr_* and
every v_mfma_* sits inside an ;;#ASMSTART / ;;#ASMEND pair, and the kernel descriptor allocates all 256 accumulator registers."""
import re
import subprocess
import sys

src, hipcc, arch = sys.argv[1], sys.argv[2], sys.argv[3]
asm = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", f"--offload-arch={arch}", "-fno-gpu-rdc", "--cuda-device-only", "-S", src, "-o", "-"],
                     check=True, capture_output=True, text=True).stdout
bad = []
kernels = re.findall(r"^(_ZN\S*scanline_batched_kernel\S*):", asm, flags=re.M)
if len(kernels) != 2:
    bad.append(f"expected two scanline_batched_kernel instantiations, found {len(kernels)}")
for name in kernels:
    body = asm[asm.index("\n" + name + ":"):]
    body = body[:body.index("s_endpgm")]
    inside = False
    n_acc = n_mfma = 0
    for line in body.split("\n"):
        if "#ASMSTART" in line:
            inside = True
        elif "#ASMEND" in line:
            inside = False
        elif re.search(r"\bv_accvgpr_|\bv_mfma_", line):
            if not inside:
                bad.append(f"{name}: compiler-issued accumulator access / MFMA: {line.strip()}")
            n_acc += "v_accvgpr" in line
            n_mfma += "v_mfma" in line
        elif re.search(r"\bscratch_(load|store)", line):
            bad.append(f"{name}: scratch access: {line.strip()}")
    items = [it for it in asm.split("\n  - ") if re.search(r"\.name:\s+" + re.escape(name) + r"\s", it)]
    meta = items[0] if items else ""
    m_agpr = re.search(r"\.agpr_count:\s+(\d+)", meta)
    m_spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", meta)
    if not m_agpr or int(m_agpr.group(1)) != 256:
        bad.append(f"{name}: .agpr_count is {m_agpr.group(1) if m_agpr else '?'}, not 256")
    if not m_spill or int(m_spill.group(1)) != 0:
        bad.append(f"{name}: .vgpr_spill_count is {m_spill.group(1) if m_spill else '?'}")
    print(f"audit {name[:60]}...: {n_acc} accumulator writes, {n_mfma} MFMAs, all inside asm statements" if not bad else f"audit {name}: FAILED")
if bad:
    print("\n".join(bad[:20]), file=sys.stderr)
    sys.exit(1)
