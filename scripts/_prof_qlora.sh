cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PYTHONPATH=.
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qlora_prof -o q -- python benchmark/qlora_bench.py --steps 4 --warmup 2 --full-targets --graph > gpurun_out/qlora_prof.log 2>&1
tail -2 gpurun_out/qlora_prof.log
f=$(find gpurun_out/qlora_prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms',tot/1e6)
for r in rows[:25]:
    print(f"{r['Name'][:110]:110s} {r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):5.1f}%")
PY
find gpurun_out/qlora_prof -name "*kernel_trace.csv" -delete
