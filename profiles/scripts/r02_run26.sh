export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r2ah; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_line_under_rocprof.json 2> $out/bench_under_rocprof.err; echo "trace rc=$?"
f=$(find $out/bench_trace -name "*kernel_trace.csv" | head -1); python3 $R/profiles/summarize_trace.py $f ig_accum 0 3 > $out/bench_timed_region.txt 2>&1; head -8 $out/bench_timed_region.txt
python3 - <<PY
import csv, json
rows = [r for r in csv.DictReader(open("$f")) if "ig_accum_stream" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("rocprofv3 durations of ig_accum_stream_kernel, in launch order (us):", [round(v, 1) for v in d])
print("timed steps (launches 2-4):", round(sum(d[1:4]) / 3, 2))
line = json.load(open("$out/bench_line_under_rocprof.json"))
print("bench.py, same process: kernel-stamped events", round(line["roofline"]["avg_launch_ms"] * 1e3, 2), "us; bracketing events", round(line["roofline"]["avg_launch_ms_events_bracketing_the_launch"] * 1e3, 2), "us; frac", round(line["roofline"]["frac"], 4))
PY
cp $(find $out/bench_trace -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv; rm -rf $out/bench_trace
