#!/usr/bin/env python3
"""profiles/pmc_sq.json from the two SQ counter passes of tools/pmc_sq.sh:  tools/pmc_sq_json.py gpurun_out/sq profiles/pmc_sq.json
Raw values are per launch (mean over the profiled launches, first quarter dropped); derived figures as quoted in DESIGN 5:
SQ_* cycle counters are summed over the chip's 256 CUs (SQ_BUSY_CU_CYCLES, SQ_ACTIVE_INST_*: per CU; SQ_VALU_MFMA_BUSY_CYCLES:
per SIMD, 1024 of them; GRBM_GUI_ACTIVE: per XCD, 8 of them); one v_mfma_f32_16x16x4_f32 is 512 MOPS_F32 units x ... = 2048 FLOP
(SQ_INSTS_VALU_MFMA_MOPS_F32 counts 512 FLOP each)."""
import collections, csv, glob, json, sys

def means(d):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in rows.items():
        out[k] = {c: (lambda v: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]))(v) for c, v in cs.items()}
    return out

def main():
    src, dst = sys.argv[1], sys.argv[2]
    m = means(src)
    res = {}
    for short in ("mel_power_rp_kernel", "core_fused_kernel"):
        raw = {}
        for k, cs in m.items():
            if short in k:
                raw.update(cs)
        if not raw:
            continue
        cu_busy = raw["SQ_BUSY_CU_CYCLES"]
        wave = raw["SQ_WAVE_CYCLES"]
        d = {
            "gui_active_cycles_per_xcd": raw["GRBM_GUI_ACTIVE"] / 8,
            "cu_busy_cycles_per_cu": cu_busy / 256,
            "mfma_busy_cycles_per_simd": raw["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024,
            "mfma_util_of_gui_active": round(raw["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (raw["GRBM_GUI_ACTIVE"] / 8), 4),
            "mfma_util_of_cu_busy": round(raw["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (cu_busy / 256), 4),
            "mfma_flops_executed": raw["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512,
            "wave_cycle_split": {"issuing": round(raw["SQ_ACTIVE_INST_ANY"] / wave, 3),
                                 "waiting_waitcnt_or_barrier": round(raw["SQ_WAIT_ANY"] / wave, 3),
                                 "issue_stalled": round(1 - (raw["SQ_ACTIVE_INST_ANY"] + raw["SQ_WAIT_ANY"]) / wave, 3)},
            "valu_active_frac_of_cu_busy": round(raw["SQ_ACTIVE_INST_VALU"] / cu_busy, 3),
            "lds_active_frac_of_cu_busy": round(raw["SQ_LDS_IDX_ACTIVE"] / cu_busy, 3),
            "lds_bank_conflict_frac_of_cu_busy": round(raw["SQ_LDS_BANK_CONFLICT"] / cu_busy, 3),
        }
        res[short] = {"raw_per_launch": {k: float(f"{v:.4g}") for k, v in raw.items()}, "derived": d}
    res["_source"] = "tools/pmc_sq.sh (two rocprofv3 --pmc passes over `python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-split`), tools/pmc_sq_json.py"
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in res.items():
        if k != "_source":
            print(k, json.dumps(v["derived"]))

main()
