# Round 3, host: summaries of gpurun_out/prof_* and gpurun_out/final_r03 into profiles/r03 (committed)
set -e
cd "$(dirname "$0")/.."
TAGS="default c2 c3 c4 c5 c6 c7 c8 c9 c10 c11 c12 qp_c2 qp_c3 qp_c4 qp_c5full qp_c5full_wave_polish0 c2_batch16m"
python tools/summarize_prof.py r03 $TAGS
cp gpurun_out/final_r03/*.json gpurun_out/final_r03/*.txt profiles/r03/ 2>/dev/null || true
ls profiles/r03 | wc -l
