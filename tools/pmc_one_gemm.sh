# SQ counters of one GEMM shape under the kernels selected by the environment: bash tools/pmc_one_gemm.sh <tag> M N K [stats]
R=$GRAFT_REPO_ROOT; TAG=$1; shift; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/one_gemm.py $1 $2 $3 3 $4 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/one_gemm.py $1 $2 $3 3 $4 > $O/b.log 2>&1
cd $R
python - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for sub in ("a", "b"):
    fs = glob.glob(f"{O}/{sub}/*/*counter_collection.csv")
    if not fs:
        print(sub, "no counters", open(f"{O}/{sub}.log").read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:60]
        if "gemm" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in acc.items():
        print(k)
        for c, x in v.items(): print(f"   {c:<32} {x/3:16.0f}")
PY
