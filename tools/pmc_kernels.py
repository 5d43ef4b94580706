"""Hardware counters per kernel, one `rocprofv3 --pmc` pass per counter group (the program goes directly after `--`: no shell,
no wrapper).  Aggregates the rocpd databases into one JSON: counter sums and dispatch counts per kernel short name.

  python3 tools/pmc_kernels.py <out.json> <name-filter[|name-filter...]> -- python3 tools/one_layer.py 64 64 96 320

Derived per kernel (MI355X_MICROARCH.md "rocprofv3 PMC slots" / per-instruction constants):
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES x 4 SIMDs)   fraction of SIMD-cycles the matrix pipe is busy
  wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES, issue_stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB  (gfx950: FETCH_SIZE counts 128-B requests as 64 B; WRITE_SIZE uncalibrated)
Attainable ("composite") bound per kernel: on gfx950 the f32-input MFMA executes on the vector ALU's lanes, so a
kernel cannot be faster than the SUM of its matrix cycles and its other vector instructions at their full-occupancy issue rate
(2.44 cycles per wave64 instruction and SIMD: tools/micro/role_split.hip, row VVVV; the guide's per-instruction table says 2):
  valu_frac      = (SQ_INSTS_VALU - SQ_INSTS_MFMA) x 2.44 / (avg_us x 2.4 GHz x 1024 SIMDs)
  composite_frac = mfma_busy + valu_frac   (f32 MFMA kernels; kernels whose matrix work is bf16: max of the two), and never below
                   hbm bytes / 8 TB/s / avg_us or the LDS pipe's busy share
  floor_us       = composite_frac x avg_us  -- what the kernel's own instruction mix allows at perfect overlap of everything else."""
import glob
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys

CLK_HZ, N_SIMD, VALU_ISSUE_CYC, HBM_BPS = 2.4e9, 1024, 2.44, 8.0e12

GROUPS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
     "SQ_WAIT_INST_LDS", "SQ_WAVES"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS",
     "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU_MFMA_MOPS_F32"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INST_CYCLES_VMEM_RD", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
     "SQ_INSTS_MFMA"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["TCC_HIT_sum", "TCC_MISS_sum"],
]


# PMC_EXTRA="A,B;C,D": additional counter groups (one pass each), e.g. the TCP / TCC latency counters for one kernel
if os.environ.get("PMC_EXTRA"):
    GROUPS = GROUPS + [g.split(",") for g in os.environ["PMC_EXTRA"].split(";") if g]


def read_commit_stamp(root):
    """tools/.commit: written in the build container right before the snapshot goes to the GPU box (tools/stamp_commit.sh; the box
    has no .git).  "<hash>" or "<hash>+dirty"."""
    try:
        return open(os.path.join(root, "tools", ".commit")).read().strip() or "unknown"
    except OSError:
        return "unknown"


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    return name.split("(")[0].strip()


def main():
    out, flt = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [os.path.join(root, c) if c.endswith(".py") and not os.path.isabs(c) else c for c in cmd]    # rocprofv3 runs in /tmp
    res = {}
    for gi, grp in enumerate(GROUPS):
        d = "/tmp/pmck_%d" % gi
        shutil.rmtree(d, ignore_errors=True)
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + grp + ["-d", d, "--"] + cmd, cwd="/tmp", env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        dbs = glob.glob(d + "/**/*.db", recursive=True)
        if not dbs:
            print("pass %d failed: %s" % (gi, r.stderr[-400:]))
            continue
        cur = sqlite3.connect(dbs[0]).cursor()
        # pmc_events holds one row per dispatch, counter AND hardware instance (shader engine / XCD): values are summed over
        # the instances; the number of dispatches comes from the kernel trace of the same pass
        for name, counter, val in cur.execute("select name, counter_name, sum(counter_value) from pmc_events group by name, counter_name"):
            k = short(name)
            if not any(f in k for f in flt.split('|')):
                continue
            res.setdefault(k, {"dispatches": 0})[counter] = val
        for name, dur, n in cur.execute("select name, sum(end-start), count(*) from kernels group by name"):
            k = short(name)
            if k in res:
                res[k]["dispatches"] = n
                res[k].setdefault("avg_us_profiled", []).append(dur / n / 1e3)
    for k, e in res.items():
        if "avg_us_profiled" in e:
            e["avg_us_profiled"] = sum(e["avg_us_profiled"]) / len(e["avg_us_profiled"])
        g = e.get
        if g("SQ_BUSY_CU_CYCLES"):
            e["mfma_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4.0 * e["SQ_BUSY_CU_CYCLES"])
        if g("SQ_WAVE_CYCLES"):
            e["wait_frac"] = g("SQ_WAIT_ANY", 0) / e["SQ_WAVE_CYCLES"]
            e["issue_stall_frac"] = g("SQ_WAIT_INST_ANY", 0) / e["SQ_WAVE_CYCLES"]
            e["active_frac"] = g("SQ_ACTIVE_INST_ANY", 0) / e["SQ_WAVE_CYCLES"]
            e["lds_issue_stall_frac"] = g("SQ_WAIT_INST_LDS", 0) / e["SQ_WAVE_CYCLES"]
        if g("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"]
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            e["hbm_bytes_per_dispatch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024 / max(e["dispatches"], 1)
        if g("TCC_HIT_sum") is not None:
            e["l2_hit"] = e["TCC_HIT_sum"] / max(e["TCC_HIT_sum"] + g("TCC_MISS_sum", 0), 1)
        us, n = e.get("avg_us_profiled", 0), max(e["dispatches"], 1)
        if us and g("SQ_INSTS_VALU") is not None and "mfma_busy" in e:
            n_mfma = g("SQ_INSTS_MFMA", 0) or 0
            simd_cycles = us * 1e-6 * CLK_HZ * N_SIMD                       # SIMD-cycles of one dispatch at the nominal clock
            e["valu_insts_per_dispatch"] = (e["SQ_INSTS_VALU"] - n_mfma) / n
            e["mfma_insts_per_dispatch"] = n_mfma / n
            e["valu_frac"] = e["valu_insts_per_dispatch"] * VALU_ISSUE_CYC / simd_cycles
            f32_mfma = (g("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) or 0) >= (g("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) or 0)
            comp = e["mfma_busy"] + e["valu_frac"] if f32_mfma else max(e["mfma_busy"], e["valu_frac"])
            e["hbm_frac"] = e.get("hbm_bytes_per_dispatch", 0) / HBM_BPS / (us * 1e-6)
            if g("SQ_LDS_IDX_ACTIVE") and g("SQ_BUSY_CU_CYCLES"):
                e["lds_busy"] = e["SQ_LDS_IDX_ACTIVE"] / e["SQ_BUSY_CU_CYCLES"]         # one LDS pipe per CU
            e["waves_per_simd"] = g("SQ_WAVE_CYCLES", 0) / max(g("SQ_BUSY_CU_CYCLES", 1), 1)
            e["composite_frac"] = min(1.0, max(comp, e["hbm_frac"], e.get("lds_busy", 0.0)))
            e["floor_us"] = e["composite_frac"] * us
            e["bound_by"] = ("mfma+valu" if f32_mfma else "max(mfma, valu)") if e["composite_frac"] == min(1.0, comp) else (
                "hbm" if e["composite_frac"] == e["hbm_frac"] else "lds")
    steps = int(os.environ.get("PMC_STEPS", "0"))            # train steps the command ran (warm-up + timed): per-step totals
    summary = {}
    if steps:
        summary = {"steps_profiled": steps,
                   "hbm_bytes_per_step": sum(e.get("hbm_bytes_per_dispatch", 0) * e["dispatches"] for e in res.values()) / steps,
                   "dispatches_per_step": sum(e["dispatches"] for e in res.values()) / steps,
                   "mfma_busy_time_weighted": (sum(e.get("mfma_busy", 0) * e.get("avg_us_profiled", 0) * e["dispatches"] for e in res.values())
                                               / max(sum(e.get("avg_us_profiled", 0) * e["dispatches"] for e in res.values()), 1e-9)),
                   # attainable bound of the profiled kernels together: sum of their floors over the sum of their times
                   "composite_frac_time_weighted": (sum(e.get("floor_us", 0) * e["dispatches"] for e in res.values())
                                                    / max(sum(e.get("avg_us_profiled", 0) * e["dispatches"] for e in res.values() if "floor_us" in e), 1e-9)),
                   "floor_ms_per_step": sum(e.get("floor_us", 0) * e["dispatches"] for e in res.values()) / steps / 1e3,
                   "kernel_ms_per_step": sum(e.get("avg_us_profiled", 0) * e["dispatches"] for e in res.values()) / steps / 1e3}
    import hashlib
    json.dump({"command": " ".join(cmd), "filter": flt, "summary": summary, "kernels": res,
               # provenance (VERDICT r2 hygiene): the commit the passes were made from (DCD_COMMIT: the GPU box has no .git) and a
               # hash of the kernel list, so a bench line can say which build its counter evidence belongs to
               "commit": os.environ.get("DCD_COMMIT") or read_commit_stamp(root),
               "kernel_list_sha1": hashlib.sha1("\n".join(sorted(res)).encode()).hexdigest(),
               "notes": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md "
                        "HBM); WRITE_SIZE uncalibrated; Infinity-Cache hits are included in both.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                        "(4 * SQ_BUSY_CU_CYCLES).  Each counter group is its own pass of the same command."},
              open(out, "w"), indent=1, sort_keys=True)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        print("%-34s us %7.1f mfma %.3f valu %.3f floor %6.1f us (%.2f, %s) wait %.2f stall %.2f (lds %.2f) active %.2f ldsconf %.2f l2hit %.2f hbmMB %.1f waves/simd %.2f" % (
            k[:34], e.get("avg_us_profiled", 0), e.get("mfma_busy", 0), e.get("valu_frac", 0), e.get("floor_us", 0), e.get("composite_frac", 0),
            e.get("bound_by", "-"), e.get("wait_frac", 0), e.get("issue_stall_frac", 0),
            e.get("lds_issue_stall_frac", 0), e.get("active_frac", 0), e.get("lds_conflict_frac", 0), e.get("l2_hit", 0),
            e.get("hbm_bytes_per_dispatch", 0) / 1e6,
            # resident waves per SIMD, time average (as this rocprofv3 reports the two: one-wave-per-SIMD kernels read 1.00, the
            # forward tile kernels with __launch_bounds__(256, 2) read 2.0)
            e.get("SQ_WAVE_CYCLES", 0) / max(e.get("SQ_BUSY_CU_CYCLES", 1), 1)))


if __name__ == "__main__":
    main()
