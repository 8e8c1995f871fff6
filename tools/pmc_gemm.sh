cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU"; do
  i=$((i+1)); out=gpurun_out/pmc_gemm_$i; mkdir -p $out
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o run -- tools/prof_harness 69878 10677 10000000 10 256 1 > $out/log.txt 2>&1
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_gemm_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0][:48] + " grid " + row.get("Grid_Size", "?"), row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    if "gemm" in k:
        print("%-70s %-26s %.4g" % (k, c, v / n))
PY
