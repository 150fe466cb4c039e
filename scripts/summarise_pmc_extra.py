#!/usr/bin/env python3
"""Mean per-launch value of every counter of scripts/collect_pmc_extra.sh / collect_r03_configs.sh per kernel, plus derived
ratios.  `summarise_pmc_extra.py DIR [name-filter ...]`: kernels whose name contains one of the filters (default: the splat
kernel of the headline)."""
import collections
import csv
import glob
import json
import os
import sys


def main(out, filters=("splat_kernel<",)):
    res = {}
    for sub in ("a", "b"):
        files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if any(f in r["Kernel_Name"] for f in filters):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                agg[(name[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            res.setdefault(k, {})[c] = sum(v) / len(v)
            res[k]["launches_sampled"] = len(v)
    for k, c in res.items():
        d = {}
        if "SQ_INSTS_VMEM_RD" in c and c.get("SQ_WAVES"):
            d["load_instructions_per_wave"] = c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"]
        if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAVES"):
            d["wave_cycles_per_wave"] = c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
            d["valu_instructions_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
            d["salu_instructions_per_wave"] = c.get("SQ_INSTS_SALU", 0) / c["SQ_WAVES"]
            d["lds_instructions_per_wave"] = c.get("SQ_INSTS_LDS", 0) / c["SQ_WAVES"]
            d["store_instructions_per_wave"] = c.get("SQ_INSTS_VMEM_WR", 0) / c["SQ_WAVES"]
        if c.get("SQ_WAVE_CYCLES"):
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"):
                if name in c:
                    d[name.lower() + "_share_of_wave_cycles"] = c[name] / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
        c["derived"] = d
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], tuple(sys.argv[2:]) or ("splat_kernel<",))
