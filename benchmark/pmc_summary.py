"""Summarise `rocprofv3 --pmc ... --output-format csv` counter dumps per kernel launch shape.

    python benchmark/pmc_summary.py <dir with *_counter_collection.csv> [...] > summary.json

Every dispatch contributes one row per counter; rows are summed per (kernel, grid size, workgroup size) -- one kernel
template serves several projections of a layer (qkv, o_proj, down_proj launch the same instantiation with different
grids), and averaging them into one line hides how many bytes each one really moves -- and a few derived figures are
added: MFMA busy share of the busy cycles, LDS bank-conflict share of the LDS-active cycles, bytes fetched.

Bytes: ``TCC_EA0_RDREQ_DRAM_32B_sum`` counts the read traffic between L2 and memory in 32-byte units (a 128-byte
request counts 4), so bytes = 32 x counter.  The derived FETCH_SIZE metric is NOT used: its expression prices every
request that is not flagged as a 32-byte one at 64 bytes and takes the 128-byte requests from TCC_BUBBLE, which stays 0
on this GPU, where all requests of a streaming kernel are 128 bytes wide -- it reads low by 2x.  Per-launch figures divide by the launches of the pass that carried the counter (the SQ and TCC passes are separate
runs; dividing by the launches of both halved every per-launch byte count).  Check built into the
dump: `bench.py` run with BIGDL_PMC_CALIBRATE=1 ends with three sums over a 1 GiB fp32 tensor; the summary reports what
the counter says for those launches (`calibration_GiB_measured`, expected 1.0).  Measured on MI355X for the decode step
of Llama-2-7B sym_int4: qkv 27.2 MiB, o_proj 9.9, gate+up 48.6, down 24.5 per launch -- the weight bytes (+ x and the
attention partials for o_proj)."""
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
    calls = defaultdict(set)                          # launches seen in any pass
    per_counter = defaultdict(lambda: defaultdict(set))    # launches that reported a given counter: the passes are
    #                                                        separate runs, each with its own launches of every kernel
    per_dispatch_fetch = defaultdict(float)           # (file, dispatch) -> FETCH_SIZE, for the calibration kernel
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = (short(row["Kernel_Name"]), row.get("Grid_Size", "?"), row.get("Workgroup_Size", "?"))
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add((f, row["Dispatch_Id"]))
                per_counter[k][row["Counter_Name"]].add((f, row["Dispatch_Id"]))
                if row["Counter_Name"] == "TCC_EA0_RDREQ_DRAM_32B_sum" and "reduce_kernel" in k[0]:
                    per_dispatch_fetch[(f, row["Dispatch_Id"])] += 32.0 * float(row["Counter_Value"])
    # the calibration dispatches are the largest reduce_kernel launches of the run
    check = None
    if per_dispatch_fetch:
        top = sorted(per_dispatch_fetch.values())[-3:]
        if top[0] > 0 and top[-1] / top[0] < 1.2:       # three equal launches: that is the calibration tail
            check = round(sum(top) / len(top) / CAL_BYTES, 4)
    out = []
    for k, c in agg.items():
        def launches(counter):
            return max(1, len(per_counter[k][counter]))
        rec = {"kernel": k[0], "grid": k[1], "workgroup": k[2], "dispatches": max(len(v) for v in per_counter[k].values()),
               **{name: v for name, v in sorted(c.items())}}
        if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # SQ_BUSY_CYCLES is summed over the shader engines' SQs; the MFMA counter over SIMDs: report the raw ratio
            rec["mfma_busy_over_sq_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if "TCC_EA0_RDREQ_DRAM_32B_sum" in c:
            rec["fetch_MiB_per_dispatch"] = round(32.0 * c["TCC_EA0_RDREQ_DRAM_32B_sum"] / launches("TCC_EA0_RDREQ_DRAM_32B_sum") / 2 ** 20, 3)
        elif "TCC_EA0_RDREQ_sum" in c:
            rec["fetch_MiB_per_dispatch"] = round(128.0 * c["TCC_EA0_RDREQ_sum"] / launches("TCC_EA0_RDREQ_sum") / 2 ** 20, 3)   # all requests 128 B wide
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
            rec["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        out.append(rec)
    out.sort(key=lambda r: -r.get("SQ_WAVE_CYCLES", r.get("TCC_EA0_RDREQ_sum", 0)))
    json.dump({"calibration_GiB_measured": check, "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
