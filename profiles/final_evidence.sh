set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_final; mkdir -p $O
cd $R
python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_fp32.json
PP_CONV_DTYPE=bf16 python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_bf16.json
PP_ADAM=foreach python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_fp32_foreach_adam.json
python profiles/wgrad_one.py 1:16:16,1:32:32,1:16:32,2:32:32,2:64:64,4:64:64,4:96:96,8:112:112 2>/dev/null | grep ts= > $O/wgrad_one.txt
bash profiles/kt_train.sh r04_final_train > /dev/null 2>&1
cp $R/gpurun_out/r04_final_train/* $O/ 2>/dev/null
bash profiles/collect.sh r04_final > $O/collect.log 2>&1
python profiles/config_microbench.py --out $O > $O/configs.log 2>&1
ls $O
