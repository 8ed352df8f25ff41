# Round profile: kernel-trace stats + HBM traffic counters (separate passes, as the microarch guide prescribes) for one bench run.
export TMPDIR=/tmp
R=${1:-r01}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err; cat gpurun_out/bench_$R.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o $R -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$R.trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o $R -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_$R.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o $R -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_$R.write.log 2>&1
python - <<PY
import csv, json, glob
out = {}
for name in ("fetch", "write"):
    tot = 0.0; n = 0
    for f in glob.glob("gpurun_out/prof_$R/%s/*counter_collection.csv" % name):
        for r in csv.DictReader(open(f)):
            if "mlrt_render_kernel" in r["Kernel_Name"]:
                tot += float(r["Counter_Value"]); n += 1
    out[name + "_size_kb_per_launch"] = tot
# gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM): the scene blob is staged with 16 B/lane loads
out["hbm_bytes_per_launch"] = (2.0 * out["fetch_size_kb_per_launch"] + out["write_size_kb_per_launch"]) * 1024.0
out["note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, one launch of the C2 render; FETCH_SIZE doubled per the gfx950 correction"
json.dump(out, open("gpurun_out/hbm_traffic_$R.json", "w"), indent=1); print(out)
PY
