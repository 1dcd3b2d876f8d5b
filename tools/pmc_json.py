#!/usr/bin/env python3
"""summary.txt of tools/pmc_score.sh -> the entry of profiles/pmc_traffic.json that bench.py reads.

usage: pmc_json.py <summary.txt> <json-key> <particles> <beams> <source-name> [<out.json>]

  hbm_bytes_per_launch  (2 x FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE reports half of the streamed bytes on gfx950
                        (MI355X_MICROARCH.md); separate --pmc passes, means over the dispatches of the kernel
  issue_frac            SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs over the kernel's cycles (GRBM_GUI_ACTIVE is summed
                        over the 8 XCDs): the share of the kernel's duration in which a SIMD's VALU is issuing
"""
import json
import sys


def parse(path):
    d, kernel = {}, None
    for line in open(path):
        if not line.startswith(" "):
            if kernel is not None and d:
                break  # first kernel only
            kernel = line.strip()
            continue
        name, _, rest = line.strip().partition(" ")
        d[name] = float(rest.split("mean=")[1])
    return kernel, d


def main():
    path, key, particles, beams, source = sys.argv[1:6]
    out = sys.argv[6] if len(sys.argv) > 6 else None
    kernel, d = parse(path)
    rec = {"kernel": kernel, "source": source, "particles": int(particles), "beams": int(beams)}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        rec.update(FETCH_SIZE_KB=d["FETCH_SIZE"], WRITE_SIZE_KB=d["WRITE_SIZE"],
                   hbm_bytes_per_launch=(2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0)
    if d.get("SQ_ACTIVE_INST_VALU") and d.get("GRBM_GUI_ACTIVE"):
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        rec["issue_frac"] = d["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc
        rec["issue_note"] = ("SQ_ACTIVE_INST_VALU %.4g x 4 cycles / 1024 SIMDs = %.4g cycles of the kernel's %.4g "
                             "(GRBM_GUI_ACTIVE / 8 XCDs)" % (d["SQ_ACTIVE_INST_VALU"],
                                                             d["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0, cyc))
        rec["bound"] = "valu_issue" if rec["issue_frac"] >= 0.5 else "latency"
    if d.get("TCP_TOTAL_CACHE_ACCESSES_sum") and d.get("SQ_INSTS_VMEM_RD"):
        rec["l1_accesses_per_gather"] = d["TCP_TOTAL_CACHE_ACCESSES_sum"] / d["SQ_INSTS_VMEM_RD"]
    # (the headline record's `bound` / `bound_note` were set by hand from the measurements in
    # profiles/r03_score_field_variants.md: counters of one workload alone do not tell the L1's access rate from issue)
    if d.get("SQ_INSTS_VALU"):
        rec["valu_wave_instructions_per_launch"] = d["SQ_INSTS_VALU"]
    text = json.dumps({key: rec}, indent=1)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
