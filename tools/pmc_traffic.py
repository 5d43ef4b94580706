"""HBM traffic of the DCN kernels from two `rocprofv3 --pmc` passes (FETCH_SIZE and WRITE_SIZE need separate passes:
3 + 2 of the 4 TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d A -o f -- python tools/dcn_one_pass.py 8 0.5
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d B -o w -- python tools/dcn_one_pass.py 8 0.5
  python tools/pmc_traffic.py A B profiles/dcn_traffic.json 8

Units and corrections as the guide prescribes (MI355X_MICROARCH.md "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes actually fetched for wide coalesced reads, so it is doubled (the factor is calibrated for
16 B/lane streams; our dword gathers are not separately calibrated -- stated in the output)."""
import glob, json, sqlite3, sys

def total(dirname, counter):
    f = glob.glob(dirname + "/*.db")[0]
    cur = sqlite3.connect(f).cursor()
    rows = cur.execute("select name, sum(counter_value), count(*) from pmc_events where counter_name = ? and name like '%dcn_%' "
                       "group by name", (counter,)).fetchall()
    return {r[0].split("(")[0][-60:]: (r[1], r[2]) for r in rows}

fetch_dir, write_dir, out, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
REPS = 2
fe, wr = total(fetch_dir, "FETCH_SIZE"), total(write_dir, "WRITE_SIZE")
fetch_kib = sum(v[0] for v in fe.values()) / REPS
write_kib = sum(v[0] for v in wr.values()) / REPS
res = {
    "bytes_per_step_batch%d" % batch: int((2.0 * fetch_kib + write_kib) * 1024),
    "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
    "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 128-B requests as 64 B)",
    "scope": "all dcn_* kernels of one forward+backward over the 16 DCN layers, batch %d, offsets 0.5*randn px" % batch,
    "per_kernel_fetch_kib": {k: v[0] / REPS for k, v in fe.items()},
    "per_kernel_write_kib": {k: v[0] / REPS for k, v in wr.items()},
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in list(res)[:3]}))
