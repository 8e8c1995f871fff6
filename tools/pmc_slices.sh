cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for sl in 1 4; do
  for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE"; do
    tag=$(echo $grp | cut -d' ' -f1)
    out=gpurun_out/pmc_s${sl}_${tag}
    mkdir -p $out
    SG_GATHER_SLICES_FORCE=$sl timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o run -- tools/prof_harness 69878 10677 10000000 10 256 3 > $out/log.txt 2>&1
  done
done
python - <<'PY'
import csv, glob, collections
for sl in (1, 4):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("gpurun_out/pmc_s%d_*/**/*counter_collection.csv" % sl, recursive=True):
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0][:60], row["Counter_Name"])
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    print("== slices", sl)
    for (k, c), (v, n) in sorted(agg.items()):
        if "gather_kernel" in k:
            print("%-50s %-14s %.4g per dispatch (%d)" % (k, c, v / n, n))
PY
