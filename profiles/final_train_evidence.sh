set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_final2; mkdir -p $O
cd $R
python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_fp32.json
PP_CONV_DTYPE=bf16 python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_bf16.json
PP_ADAM=foreach python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_fp32_foreach_adam.json
bash profiles/kt_train.sh r03_final2_train > /dev/null 2>&1
cp $R/gpurun_out/r03_final2_train/* $O/ 2>/dev/null
cat $O/pytest_gpu.txt $O/smoke.txt
