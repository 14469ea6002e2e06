cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp ROUND=r06 COMMIT=4242455
# every profile of the round again on the final tree (no kernel changed after 9e91299 / 0b8f7ff; the files then carry this commit)
echo '{}' > profiles/hbm_traffic.json
SPECS_FILE=tools/r06_specs.txt bash tools/run_profiles.sh 2>&1 | grep -v "simple_timer" | tail -60
