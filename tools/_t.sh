python3 -m pytest tests -x -q -m gpu -k "lightglue or lg_ or two_view or matcher" 2>&1 | tail -3 || exit 1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for p in 1 8; do
rocprofv3 --kernel-trace -d $R/gpurun_out/lgk$p -o lg -- python3 $R/tools/bench_lightglue.py --pairs $p --steps 20 --warmup 3 > $R/gpurun_out/lgk$p.log 2>&1
echo "== pairs $p (under rocprofv3)"
python3 $R/tools/lg_kernel_table.py $R/gpurun_out/lgk$p/lg_results.db
done
cd $R
python3 tools/bench_lightglue.py --pairs 8 --steps 50 --warmup 5 > gpurun_out/lg_bench.jsonl 2>/dev/null
python3 tools/bench_lightglue.py --pairs 1 --steps 50 --warmup 5 >> gpurun_out/lg_bench.jsonl 2>/dev/null
cat gpurun_out/lg_bench.jsonl
