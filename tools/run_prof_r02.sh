# all profile passes of the round (GPU box): tools/prof_cfg.sh per command, summaries made afterwards on the host by
# tools/summarize_prof.py r02 <tags>
for c in 2 3 4 5; do bash tools/prof_cfg.sh c$c --config $c; done
bash tools/prof_cfg.sh c2polish0 --config 2 --polish 0
bash tools/prof_cfg.sh c2polish1 --config 2 --polish 1
for sh in c2 c3 c4 c5full; do bash tools/prof_cfg.sh qp_$sh --config qp --shape $sh; done
bash tools/prof_cfg.sh qp_c5full_wave_polish0 --config qp --shape c5full --lanes 64 --polish 0
