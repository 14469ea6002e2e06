cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp ROUND=r06 COMMIT=4375169
# (fresh traffic table: every entry of hbm_traffic.json comes from this run)
echo '{}' > profiles/hbm_traffic.json
SPECS_FILE=tools/r06_specs.txt bash tools/run_profiles.sh 2>&1 | grep -v "simple_timer" | tail -40
