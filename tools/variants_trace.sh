#!/bin/bash
# Developer tool (GPU box): per-kernel durations of the step-only bench under each library variant built by tools/build_variant.sh
# usage: tools/variants_trace.sh <tag> <variant|default>...   -> gpurun_out/<tag>/<variant>.txt
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset IMPACT_VOXEL_HIP_LIB; else export IMPACT_VOXEL_HIP_LIB=$PWD/impact_amd/lib/var_$v/libimpact_voxel_hip.so; fi
  rocprofv3 --kernel-trace --stats -d "$out/trace_$v" -o trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pile --plain $BENCH_ARGS > "$out/bench_$v.json" 2> "$out/stderr_$v.log"
  python tools/rocpd_stats.py "$(ls "$out"/trace_$v/*.db | tail -1)" > "$out/kernel_stats_$v.csv"
  rm -rf "$out/trace_$v"
  echo "== $v" | tee "$out/$v.txt"
  python - "$out/kernel_stats_$v.csv" "$out/bench_$v.json" <<'PY' | tee -a "$out/$v.txt"
import csv, sys, re, json
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    print(f"{name:28s} {int(r['Calls']):4d} {float(r['AverageNs'])/1000:9.1f} us")
try:
    d = json.load(open(sys.argv[2])); print("ms_per_step", round(d["ms_per_step"], 4), "sdf_sample", d["stage_ms"]["sdf_sample"])
except Exception as e:
    print("bench line unreadable:", e)
PY
done
