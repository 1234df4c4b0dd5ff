#!/bin/bash
# What differs between the mesher's two run-to-run modes? Per fresh process on one box:
#   part 1: clocks / power / temperature before and after the all-surface bench leg + the launch's min / median / max over its steps
#   part 2: rocprofv3 --kernel-trace --pmc passes (one counter group per process, nothing else traced) — each process lands in one mode
#           or the other; the table pairs k_step_emit's duration in that process with its counters per launch
# usage (GPU box): tools/emit_modes2.sh <tag> [plain processes] [processes per counter group]
tag=${1:-modes}
n_plain=${2:-8}
n_pmc=${3:-4}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-pile --plain --workload dense"
smi() {  # one line: sclk / mclk / fclk / socclk, power, temperature (whatever this box's tools report to an ordinary user)
  (rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|power\|Temperature (Sensor junction)\|Temperature (Sensor memory)" | sed 's/^.*GPU\[0\]\s*:\s*//' | tr '\n' ';') || true
}
echo "== part 1: $n_plain fresh processes" | tee "$out/modes_plain.log"
for i in $(seq 1 $n_plain); do
  before=$(smi)
  python3 bench.py $ARGS 2>/dev/null | tail -1 > /tmp/em.json
  after=$(smi)
  python3 - "$i" "$before" "$after" <<'P' | tee -a "$out/modes_plain.log"
import json, sys
d = json.load(open("/tmp/em.json"))
print(f"process {sys.argv[1]}: ms/step {d['ms_per_step']:.4f} emit {d['stage_ms']['emit']:.4f} derive {d['stage_ms']['derive']:.4f} sample {d['stage_ms']['sdf_sample']:.4f}")
print("   before:", sys.argv[2][:400])
print("   after: ", sys.argv[3][:400])
P
done
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_WRREQ_STALL\|TCC_EA0_WRREQ\b\|TCC_TAG_STALL\|TCP_PENDING_STALL_CYCLES\|TCP_UTCL1_TRANSLATION_MISS\|TCP_UTCL1_TRANSLATION_HIT\|GRBM_GUI_ACTIVE\|TCC_EA0_RDREQ\b\|TCC_HIT\b\|TCC_MISS\b\|TCC_EA0_WRREQ_64B\|TCC_BUSY\|TCP_TCC_WRITE_REQ\b\|TCC_EA0_ATOMIC\b\|TCC_NORMAL_WRITEBACK\|TCC_ALL_TC_OP_WB_WRITEBACK" | sort | uniq -c > "$out/counters_available.txt"
echo "== part 2: counter passes" | tee "$out/modes_pmc.log"
groups=("GRBM_GUI_ACTIVE TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_TAG_STALL" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ TCC_HIT TCC_MISS TCC_EA0_WRREQ_64B")
gi=0
for grp in "${groups[@]}"; do
  gi=$((gi+1))
  for i in $(seq 1 $n_pmc); do
    rm -rf "$out/p"
    rocprofv3 --kernel-trace --pmc $grp -d "$out/p" -o p -- python3 bench.py $ARGS > "$out/p.log" 2>&1
    db=$(ls "$out"/p/*.db 2>/dev/null | tail -1)
    if [ -z "$db" ]; then echo "group $gi process $i: no database ($(tail -1 "$out/p.log" | cut -c1-200))" | tee -a "$out/modes_pmc.log"; continue; fi
    python3 - "$db" "$gi" "$i" <<'PY' | tee -a "$out/modes_pmc.log"
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
dur = list(db.execute("select avg(end-start), min(end-start), max(end-start), count(*) from kernels where name like '%k_step_emit%' and (end-start) > 200000"))[0]
rows = db.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%k_step_emit%' group by counter_name")
c = {k: v for k, v in rows}
gui = c.get("GRBM_GUI_ACTIVE")
extra = f" clock~{gui / 8 / dur[0]:.3f} GHz" if gui and dur[0] else ""
print(f"group {sys.argv[2]} process {sys.argv[3]}: k_step_emit avg {dur[0] / 1e3:.1f} us (min {dur[1] / 1e3:.1f}, max {dur[2] / 1e3:.1f}, n {dur[3]}){extra} | " + " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
PY
  done
done
rm -rf "$out/p"
