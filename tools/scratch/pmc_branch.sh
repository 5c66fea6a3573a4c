# Scratch (GPU box): branch and scalar-memory instruction counts per wave of a bench command.  tools/scratch/pmc_branch.sh <tag> <bench args>
T=$1; shift; OUT=gpurun_out/pmcb_$T; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/$OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pcie "$@" > $GRAFT_REPO_ROOT/$OUT/log.txt 2>&1 || echo "pass failed"
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv,glob,collections
rows=[]
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows: agg[r["Kernel_Name"][:70]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    w=v.get("SQ_WAVES",1) or 1
    print("$T", k, {c: round(x/w,1) for c,x in v.items() if c!="SQ_WAVES"}, "waves", int(w))
PY
