# Round 4, host: summaries of gpurun_out/prof_* and gpurun_out/final_r04 into profiles/r04 (committed)
#   tools/collect_r04.sh [tags...]     default: every tag that has a gpurun_out/prof_<tag>/trace
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/r04
TAGS="$@"
if [ -z "$TAGS" ]; then
  for d in gpurun_out/prof_*; do t=${d#gpurun_out/prof_}; [ -f $d/trace/t_results.db ] && TAGS="$TAGS $t"; done
fi
for t in $TAGS; do
  case $t in
    *_direct)  # one kernel-trace pass of the direct-launch leg alone (tools/prof_direct.sh)
      python tools/rocprof_db_stats.py gpurun_out/prof_$t/trace/t_results.db > profiles/r04/${t}_kernel_stats.csv
      python tools/rocprof_db_stats.py gpurun_out/prof_$t/trace/t_results.db --window 20:220 > profiles/r04/${t}_timed_window.json
      echo "$t: $(sed -n 2p profiles/r04/${t}_kernel_stats.csv | cut -d, -f2-4)";;
    *) [ -d gpurun_out/prof_$t/pmc_fetch ] && python tools/summarize_prof.py r04 $t;;
  esac
done
cp gpurun_out/final_r04/*.json gpurun_out/final_r04/*.txt profiles/r04/ 2>/dev/null || true
ls profiles/r04 | wc -l
