set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_util
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM -d $OUT -o pmc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1 || { tail -3 $OUT/log.txt; exit 1; }
python3 - $OUT <<'PY'
import sqlite3, glob, sys
con=sqlite3.connect(glob.glob(sys.argv[1]+"/*.db")[0])
q="select kernel_name, dispatch_id, counter_name, value from counters_collection where kernel_name like '%rt::k_%' order by dispatch_id"
d={}
for k,i,c,v in con.execute(q):
    d.setdefault((i,k[:34]),{})[c]=v
for (i,k),c in list(d.items())[3:20]:
    lanes = c['SQ_THREAD_CYCLES_VALU']/max(1,c["SQ_ACTIVE_INST_VALU"]) if 'SQ_THREAD_CYCLES_VALU' in c else -1
    print(i,k, "VALU insts %.3g thread_cycles %.3g active %.3g -> lanes/instr %.1f ; wave_cycles %.3g wait_any %.0f%% vmem_rd %.3g vmem_wr %.3g"%(c['SQ_INSTS_VALU'],c['SQ_THREAD_CYCLES_VALU'],c['SQ_ACTIVE_INST_VALU'],lanes,c['SQ_WAVE_CYCLES'],100*c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES'],c['SQ_INSTS_VMEM_RD'],c['SQ_INSTS_VMEM_WR']))
PY
