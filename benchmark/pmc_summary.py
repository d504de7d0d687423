"""Summarise `rocprofv3 --pmc ... --output-format csv` counter dumps per kernel name.

    python benchmark/pmc_summary.py <dir with *_counter_collection.csv> [...] > summary.json

Every dispatch contributes one row per counter; rows are summed per (kernel, counter) and a few derived figures are
added: MFMA busy share of the busy cycles, LDS bank-conflict share of the LDS-active cycles, bytes fetched."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("void bigdl::", "")


def main():
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add((f, row["Dispatch_Id"]))
    out = []
    for k, c in agg.items():
        rec = {"kernel": k, "dispatches": len(calls[k]), **{n: v for n, v in sorted(c.items())}}
        if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # SQ_BUSY_CYCLES is summed over the shader engines' SQs; the MFMA counter over SIMDs: report the raw ratio
            rec["mfma_busy_over_sq_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if "FETCH_SIZE" in c:
            rec["fetch_MiB_per_dispatch_x2"] = round(2 * c["FETCH_SIZE"] / 1024 / max(1, len(calls[k])), 3)
        out.append(rec)
    out.sort(key=lambda r: -r.get("SQ_WAVE_CYCLES", r.get("FETCH_SIZE", 0)))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
