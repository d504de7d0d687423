"""Summarise `rocprofv3 --pmc ... --output-format csv` counter dumps per kernel launch shape.

    python benchmark/pmc_summary.py <dir with *_counter_collection.csv> [...] > summary.json

Every dispatch contributes one row per counter; rows are summed per (kernel, grid size, workgroup size) -- one kernel
template serves several projections of a layer (qkv, o_proj, down_proj launch the same instantiation with different
grids), and averaging them into one line hides how many bytes each one really moves -- and a few derived figures are
added: MFMA busy share of the busy cycles, LDS bank-conflict share of the LDS-active cycles, bytes fetched.

Bytes: the derived FETCH_SIZE metric prices a read request at the request sizes of older parts, so on this GPU its
unit is not a KiB.  Instead of guessing the factor it is calibrated: `bench.py` run with BIGDL_PMC_CALIBRATE=1 ends
with three sums over a 1 GiB fp32 tensor (a `reduce_kernel` that reads exactly 2**30 bytes), and every kernel's
FETCH_SIZE is scaled by (1 GiB / the calibration dispatch's FETCH_SIZE).  Without a calibration dispatch in the dump
the raw value is reported and marked as uncalibrated."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

CAL_BYTES = float(1 << 30)


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("void bigdl::", "")


def main():
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    per_dispatch_fetch = defaultdict(float)           # (file, dispatch) -> FETCH_SIZE, for the calibration kernel
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = (short(row["Kernel_Name"]), row.get("Grid_Size", "?"), row.get("Workgroup_Size", "?"))
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add((f, row["Dispatch_Id"]))
                if row["Counter_Name"] == "FETCH_SIZE" and "reduce_kernel" in k[0]:
                    per_dispatch_fetch[(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    # the calibration dispatches are the largest reduce_kernel launches of the run
    factor = None
    if per_dispatch_fetch:
        top = sorted(per_dispatch_fetch.values())[-3:]
        if top[0] > 0 and top[-1] / top[0] < 1.2:       # three equal launches: that is the calibration tail
            factor = CAL_BYTES / (sum(top) / len(top))
    out = []
    for k, c in agg.items():
        n = max(1, len(calls[k]))
        rec = {"kernel": k[0], "grid": k[1], "workgroup": k[2], "dispatches": len(calls[k]),
               **{name: v for name, v in sorted(c.items())}}
        if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # SQ_BUSY_CYCLES is summed over the shader engines' SQs; the MFMA counter over SIMDs: report the raw ratio
            rec["mfma_busy_over_sq_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if "FETCH_SIZE" in c:
            if factor is not None:
                rec["fetch_MiB_per_dispatch"] = round(c["FETCH_SIZE"] * factor / n / 2 ** 20, 3)
            else:
                rec["fetch_size_raw_per_dispatch_uncalibrated"] = round(c["FETCH_SIZE"] / n, 3)
        out.append(rec)
    out.sort(key=lambda r: -r.get("SQ_WAVE_CYCLES", r.get("FETCH_SIZE", 0)))
    json.dump({"fetch_bytes_per_FETCH_SIZE_unit": factor, "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
