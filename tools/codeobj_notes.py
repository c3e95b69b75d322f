#!/usr/bin/env python3
"""Per-kernel resource figures from the code objects inside libxaac_amd.so (the AMDGPU metadata note: VGPRs, AGPRs, SGPRs,
LDS, scratch, spills) and the waves per SIMD they allow on gfx950 (512 VGPRs per SIMD lane, 160 KB of LDS per CU) --
rocprofv3's `vgpr=` column is a half count on this chip, these are the numbers the hardware schedules by.

  python tools/codeobj_notes.py [library.so] > profiles/rNN_codeobj_notes.txt     (runs without a GPU)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_of(code_object):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", code_object], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", "\n" + txt)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda key, d="0": (re.search(r"\.%s:\s+'?([^\n']+)'?" % re.escape(key), blk) or [None, d])[1]
        name = g("name", "?")
        if name.endswith(".kd"):
            name = name[:-3]
        out.append(dict(name=name, vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")),
                        lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size")),
                        vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count")),
                        wg=int(g("max_flat_workgroup_size"))))
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [re.sub(r"\(.*", "", s) for s in p.stdout.split("\n")]


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "libxaac_amd", "libxaac_amd.so")
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, capture_output=True)
        ks = []
        for f in sorted(os.listdir(tmp)):
            if "hipv4-amdgcn" in f:
                ks += kernels_of(os.path.join(tmp, f))
    finally:
        shutil.rmtree(tmp)
    for k, n in zip(ks, demangle([k["name"] for k in ks])):
        k["name"] = n
    print("# %s: code-object metadata (llvm-readelf --notes); waves/SIMD = min(8, 512 // alloc(vgpr: the unified count, agpr included), LDS limit)" % os.path.basename(lib))
    print("%-58s %5s %5s %5s %8s %8s %7s %7s %5s %11s" % ("kernel", "vgpr", "agpr", "sgpr", "lds_B", "scratch", "v_spill", "s_spill", "wg", "waves/SIMD"))
    for k in sorted(ks, key=lambda k: k["name"]):
        regs = k["vgpr"]  # gfx90a and later: .vgpr_count is the unified total, the AccVGPRs (.agpr_count) included
        alloc = max(8, (regs + 7) // 8 * 8)
        by_regs = min(8, 512 // alloc)
        waves_wg = max(1, k["wg"] // 64)
        wgs_cu = (160 * 1024) // k["lds"] if k["lds"] else 1 << 20
        by_lds = wgs_cu * waves_wg / 4.0
        # a workgroup's waves spread over the CU's four SIMDs
        lim = min(by_regs, by_lds)
        print("%-58s %5d %5d %5d %8d %8d %7d %7d %5d %6.1f (%s)" % (k["name"][:58], k["vgpr"], k["agpr"], k["sgpr"], k["lds"],
                                                                  k["scratch"], k["vspill"], k["sspill"], k["wg"], lim,
                                                                  "regs" if by_regs <= by_lds else "lds"))


if __name__ == "__main__":
    main()
