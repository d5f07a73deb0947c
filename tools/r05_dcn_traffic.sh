#!/bin/bash
# HBM traffic of the stand-alone modulated-DCN operator from rocprofv3 counters CALIBRATED for its access width (GPU box, repo root):
# (1) tools/micro/fetch_calib.hip reads / writes a known byte count with the operator's pattern (27 coalesced dword streams per
# thread) under --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes); (2) the same two passes over tools/dcn_bench.py; (3) the
# operator's bytes = its counter x (known bytes / calibration counter).  -> gpurun_out/${TAG}_dcn_traffic.json + the csvs
set -u
TAG=${TAG:-r05}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
hipcc -O3 --offload-arch=gfx950 "$root/tools/micro/fetch_calib.hip" -o /tmp/fetch_calib || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_calib_$c
  rm -rf "$out"; mkdir -p "$out"
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $c --output-format csv -d "$out" -o t -- /tmp/fetch_calib > "$root/gpurun_out/${TAG}_calib_$c.log" 2>&1)
  f=$(find "$out" -name '*counter_collection.csv' | tail -1)
  [ -n "$f" ] && cp "$f" "$root/gpurun_out/${TAG}_calib_$c.csv"
  rm -rf "$out"
done
bash tools/pmc.sh ${TAG}_dcn "FETCH_SIZE" "WRITE_SIZE" -- tools/dcn_bench.py
TAG=$TAG python3 - <<'PY'
import csv, json, os
tag = os.environ["TAG"]
def vals(path, sub, counter):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
def mean(v):
    return sum(v) / len(v) if v else None
total = 27 * 2 * 1024 * 1024 * 4
cal = {}
for name, sub in (("dword", "calib_read<float>"), ("8B", "__vector(2)>"), ("16B", "__vector(4)>")):
    f = mean(vals("gpurun_out/%s_calib_FETCH_SIZE.csv" % tag, sub, "FETCH_SIZE"))
    w = mean(vals("gpurun_out/%s_calib_WRITE_SIZE.csv" % tag, sub, "WRITE_SIZE"))
    cal[name] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "bytes_read": total, "bytes_written": total // 27,
                 "fetch_factor": total / (f * 1024) if f else None, "write_factor": (total // 27) / (w * 1024) if w else None}
f = mean(vals("gpurun_out/%s_dcn_pmc1.csv" % tag, "mdcn_fwd_kernel", "FETCH_SIZE"))
w = mean(vals("gpurun_out/%s_dcn_pmc2.csv" % tag, "mdcn_fwd_kernel", "WRITE_SIZE"))
ff, wf = cal["dword"]["fetch_factor"], cal["dword"]["write_factor"]
out = {"calibration": cal,
       "mdcn_fwd_17x96x72_x16": {"kernel": "mdcn_fwd_kernel<17,1,true>, 16 clips, one dilation (tools/dcn_bench.py)",
                                 "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "fetch_factor": ff, "write_factor": wf,
                                 "hbm_bytes_per_launch": (f * 1024 * ff + w * 1024 * wf) if None not in (f, w, ff, wf) else None,
                                 "algorithmic_bytes_per_launch": 13630464 * 16,
                                 "note": "FETCH_SIZE / WRITE_SIZE of the operator (separate --pmc passes) times the factors measured by "
                                         "tools/micro/fetch_calib.hip for 27 coalesced dword streams per thread - the operator's own access pattern "
                                         "(MI355X_MICROARCH.md: widths other than 16 B per lane must be calibrated)"}}
json.dump(out, open("gpurun_out/%s_dcn_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
