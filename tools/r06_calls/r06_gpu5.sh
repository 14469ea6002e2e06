cd $GRAFT_REPO_ROOT
ok=0; bad=0
for s in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 120 python tools/exp_exit_race.py $s > /tmp/o.txt 2>&1; rc=$?; if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "seed $s rc=$rc: $(tail -2 /tmp/o.txt | tr '\n' ' ' | cut -c1-200)"; fi; done
echo "exit race: $ok clean exits, $bad bad"
for f in 0 0 0; do PST_EXPR_FUSE=$f timeout 300 python tools/exp_expr_fused.py 20000000 2>&1 | grep -v amdgpu.ids | grep -c "dumped\|Segmentation\|Abort"; echo "rc=$?"; done
