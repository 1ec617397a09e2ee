#!/bin/bash
# round 6: gathered rows one step ahead (r6ahead1 = the tree before) vs two steps ahead (r6ahead2), layer by layer
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/profiles/ab_x3_libs.sh r06_x3_ab_ahead.txt "4:48:48,8:64:64,4:128:48,1:64:64:up,8:64:64:down,4:64:64,2:32:32,2:96:32,2:64:32,16:80:80,4:96:96:up" r6ahead1 r6ahead2 r6ahead1 r6ahead2
