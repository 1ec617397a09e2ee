set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-checks > gpurun_out/r04_c8_bench$i.json 2> gpurun_out/r04_c8_bench$i.err
echo "run $i rc=$? $(tail -c 300 gpurun_out/r04_c8_bench$i.json | head -c 5)"; python -c "
import json
try:
    j=json.loads(open('gpurun_out/r04_c8_bench$i.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['frac'])
except Exception as e: print('FAILED', e)"
tail -5 gpurun_out/r04_c8_bench$i.err
done
python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard ms_per_step', j['ms_per_step'], j['config']['host']['step_ms'])"
SH=1:16:16,1:16:32,2:32:32,2:96:32,4:48:48,1:64:64:up,2:16:16:down,4:32:32:down
rm -f gpurun_out/r04_c8_sweep.txt
for V in 0,0,0 32,1,0 64,1,0 32,3,0 64,3,0; do
PP_CONV_VARIANT=$V python profiles/conv_one.py 64 $SH 5 2>&1 | grep "ts=" | sed "s/^/$V: /" >> gpurun_out/r04_c8_sweep.txt
done
cat gpurun_out/r04_c8_sweep.txt
