cd $GRAFT_REPO_ROOT
ulimit -c 0
bash profiles/ab_x3_libs.sh r06_x3f_setprio.txt "4:64:64,2:32:32,2:96:32,1:64:16,8:64:64,4:128:48,1:64:64:up" g_base g_prio g_base g_prio > /dev/null 2>&1
grep -v "^ts" gpurun_out/r06_x3f_setprio.txt | paste - - | head
