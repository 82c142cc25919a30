# PMC passes over the DMA-ring weight-gradient kernel (gemm_tn3.cuh) at one probe shape: L2 hit rate / fabric reads, then SQ counters
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
shape="${1:-dec pw1}"
for pmc in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
rm -rf gpurun_out/tn3pmc2
ONLY="$shape" timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d gpurun_out/tn3pmc2 -o p --output-format csv -- python tools/wgrad_probe.py > /dev/null 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/tn3pmc2/p_counter_collection.csv")))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"][:34]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, d in acc.items():
    if "tn3" in k or "tn2" in k:
        print("$shape", k, "  ".join(f"{c}={v / n[k][c]:.3g}" for c, v in d.items()))
PY
done
