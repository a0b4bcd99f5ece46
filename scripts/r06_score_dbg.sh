# k_res_score timing ablations (wrong results): 0 as is, 1 no item-row fetch, 2 no user-row reads from LDS, 3 neither
cd $GRAFT_REPO_ROOT
for d in 0 1 2 3; do
  MFM_RES_SCORE_DBG=$d bash scripts/prof_cfg.sh scoredbg$d --steps 10 --warmup 2 --no-other-configs --long-seconds 0 | grep -E "k_res_score" | cut -c1-110 | sed "s/^/dbg=$d /"
done
MFM_RES_SCORE_PROF=3 python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --fit-iters 0 --no-other-configs --long-seconds 0 2>&1 | grep "k_res_score" | head -3
