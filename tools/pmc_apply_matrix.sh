# HBM traffic of the general ApplyMatrix kernel at 256^3 from the PMC counters (two separate passes, each with --kernel-trace only, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes); run on the GPU box from the repo root, prints the JSON for profiles/
export TMPDIR=/tmp
OUT=gpurun_out/pmc_am; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/prof_kernels.py apply_matrix 10 > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/prof_kernels.py apply_matrix 10 > $OUT/write.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, json, statistics
def vals(d, name):
    f = glob.glob("gpurun_out/pmc_am/%s/*/*counter_collection.csv" % d)[0]
    out = []
    for r in csv.DictReader(open(f)):
        if "k_apply_matrix_v5" in r["Kernel_Name"] and r["Counter_Name"] == name:
            out.append(float(r["Counter_Value"]))
    return out
fe, wr = vals("fetch", "FETCH_SIZE"), vals("write", "WRITE_SIZE")
fk, wk = statistics.median(fe), statistics.median(wr)
raw = (fk + wk) * 1024
corr = (2 * fk + wk) * 1024
print(json.dumps({"kernel": "k_apply_matrix_v5<false,true,2> 256^3 (non-temporal streams, round 3)", "FETCH_SIZE_KiB_median": fk, "WRITE_SIZE_KiB_median": wk,
                  "hbm_bytes_per_launch_raw": raw, "hbm_bytes_per_launch": corr, "algorithmic_bytes_per_launch": 28 * 256 ** 3,
                  "launches_sampled": len(fe),
                  "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for 16-B/lane streaming reads -> read side doubled; WRITE_SIZE exact; separate --pmc passes"}, indent=1))
PY
