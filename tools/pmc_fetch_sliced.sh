cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for sl in 4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    out=gpurun_out/pmc2_s${sl}_${c}; mkdir -p $out
    SG_GATHER_SLICES_FORCE=$sl timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o run -- tools/prof_harness 69878 10677 10000000 10 256 1 > $out/log.txt 2>&1
    echo "$c rc=$?"; ls $out
  done
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc2_s4_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0][:60], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print("%-50s %-14s %.4g per dispatch (%d)" % (k, c, v / n, n))
PY
