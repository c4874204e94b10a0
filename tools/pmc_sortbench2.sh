#!/bin/bash
# SQ counters of one sortbench2 run (both scatter generations), summarised per kernel name.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sb2
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY -d $OUT --output-format csv -- $GRAFT_REPO_ROOT/tools/sortbench2 6000000 27 2 > $OUT/run.log 2>&1
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_sb2'
f=glob.glob(out+'/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
disp=set()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        disp.add((k,r['Dispatch_Id']))
for k,_ in disp: n[k]+=1
for k in acc:
    print(k, 'dispatches', n[k])
    for c,v in sorted(acc[k].items()): print('   %-24s %14.0f per dispatch'%(c, v/n[k]))
PY
