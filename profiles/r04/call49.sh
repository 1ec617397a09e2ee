set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_train_final; mkdir -p $O
python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_fp32.json
PP_CONV_DTYPE=bf16 python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_bf16.json
python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_fp32_b.json
bash profiles/kt_train.sh r04_train_final_kt > /dev/null 2>&1
cp gpurun_out/r04_train_final_kt/* $O/ 2>/dev/null
python profiles/train_sync_trace.py 1 2>&1 | grep -v amdgpu.ids > $O/train_sync_trace.txt
python profiles/infer_sync_trace.py 2>&1 | grep -v amdgpu.ids > $O/infer_sync_trace_shard.txt
for f in train_fp32 train_fp32_b train_bf16; do python -c "import json; j=json.load(open('$O/$f.json')); print('$f', [v['ms_per_step'] for v in j['modes'].values()])"; done
ls $O
