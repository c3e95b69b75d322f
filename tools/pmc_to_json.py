#!/usr/bin/env python3
"""profiles/<tag>_sbr_pmc_hbm.txt + <tag>_sbr_pmc_sq.txt (rocprof_summary.py pmc tables of the C4 probe) ->
profiles/pmc_latest.json, which bench.py reads for roofline.traffic and roofline.valu:
  python tools/pmc_to_json.py <tag>
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB; the x2 is the gfx950 correction MI355X_MICROARCH.md prescribes,
calibrated in round 1 on a 256 MiB copy); wave-instruction counts are SQ_INSTS_* per launch; active lanes per VALU
instruction = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU (63.3 for the IMDCT kernel, whose lanes are all busy)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C4_KERNELS = ("xaac_imdct_ola_kernel", "xaac_qmf_analysis_hq_kernel", "xaac_sbr_core_kernel", "xaac_sbr_core_list_kernel",
              "xaac_ps_kernel", "xaac_qmf_synthesis_pair_kernel")


def table(path):
    rows = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+((?:SQ_|FETCH|WRITE)\w+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            rows.setdefault(m.group(1).strip(), {})[m.group(2)] = (float(m.group(4)), int(m.group(3)), float(m.group(5)))
    return rows


def main(tag):
    hbm = table(os.path.join(ROOT, "profiles", tag + "_sbr_pmc_hbm.txt"))
    sq = table(os.path.join(ROOT, "profiles", tag + "_sbr_pmc_sq.txt"))
    out = {"source": "profiles/%s_sbr_pmc_hbm.txt, profiles/%s_sbr_pmc_sq.txt: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_* "
                     "in separate runs) over tools/pmc_probe_sbr.py, one C4 step = 8192 stream-frames" % (tag, tag), "kernels": {}}
    # the library the counters were taken on: tools/profile_r06.sh writes xaac_version()'s build id beside the passes
    try:
        out["build_id"] = open(os.path.join(ROOT, "profiles", tag + "_build_id.txt")).read().strip() or None
    except OSError:
        out["build_id"] = None
    tot = {"bytes": 0, "valu": 0, "salu": 0, "lds": 0}
    for name in sorted(set(hbm) | set(sq)):
        if not any(name.replace("void ", "").startswith(k) for k in C4_KERNELS):
            continue
        h, s = hbm.get(name, {}), sq.get(name, {})
        e = {}
        if "FETCH_SIZE" in h and "WRITE_SIZE" in h:
            e["hbm_bytes"] = int((2 * h["FETCH_SIZE"][0] + h["WRITE_SIZE"][0]) * 1024)
            e["avg_us"] = h["FETCH_SIZE"][2]
            tot["bytes"] += e["hbm_bytes"]
        if "SQ_INSTS_VALU" in s:
            v = s["SQ_INSTS_VALU"][0]
            e.update(valu=int(v), salu=int(s.get("SQ_INSTS_SALU", (0,))[0]), lds=int(s.get("SQ_INSTS_LDS", (0,))[0]),
                     active_lanes=round(s["SQ_THREAD_CYCLES_VALU"][0] / v, 1) if v and "SQ_THREAD_CYCLES_VALU" in s else None,
                     wait_frac=round(s["SQ_WAIT_ANY"][0] / s["SQ_WAVE_CYCLES"][0], 3) if "SQ_WAVE_CYCLES" in s and s["SQ_WAVE_CYCLES"][0] else None,
                     busy_frac=round(s["SQ_ACTIVE_INST_ANY"][0] / s["SQ_WAVE_CYCLES"][0], 3) if "SQ_WAVE_CYCLES" in s and s["SQ_WAVE_CYCLES"][0] else None)
            tot["valu"] += e["valu"]; tot["salu"] += e["salu"]; tot["lds"] += e["lds"]
        out["kernels"][name.replace("void ", "")] = e
    out["c4"] = {"bytes_per_step": tot["bytes"], "valu_wave_instr": tot["valu"], "salu_wave_instr": tot["salu"], "lds_wave_instr": tot["lds"]}
    dst = os.path.join(ROOT, "profiles", "pmc_latest.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["c4"]), "->", dst)


if __name__ == "__main__":
    main(sys.argv[1])
