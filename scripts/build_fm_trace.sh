#!/bin/bash
# Trace build of the FM step's first launch for scripts/trace_fm_launch1.py: scratch copies of fm.hip, fm_body.h and
# table_sort.h under scripts/exp/l1 get time stamps (forward: start, ids known, rows summed, dense added, done; sort: the
# phases table_sort.h marks + three inside the first radix pass), built with -DTS_TRACE and linked with the in-tree objects
# into scripts/exp/libmi_tstrace.so.  Run `make` in recommendation_amd/csrc first.
set -e
mkdir -p $(dirname $0)/exp/l1
cd $(dirname $0)/exp/l1
cp ../../../recommendation_amd/csrc/{fm.hip,fm_body.h,table_sort.h} .
python3 - <<'P'
s=open('table_sort.h').read()
a=s.index('#ifdef TS_TRACE'); b=s.index('#endif', a)+len('#endif')
blk=s[a:b]; s=s[:a]+s[b:]
old="  __syncthreads();  // the caller's fill\n"
assert old in s; s=s.replace(old,old+"  TS_MARK(13);\n",1)
old="""      loc[j] = old + before;
    }
    __syncthreads();
    if (w == 0) {"""
new="""      loc[j] = old + before;
    }
    if (shift == 0) TS_MARK(14);
    __syncthreads();
    if (shift == 0) TS_MARK(15);
    if (w == 0) {"""
assert old in s; s=s.replace(old,new,1)
open('table_sort.h','w').write(s)
g=open('fm.hip').read()
old="  fm_fwd_body<IDX, DIM>(a, blockIdx.x, TS_THREADS);\n"
assert old in g; g=g.replace(old,"  TS_MARK(0);\n"+old+"  TS_MARK(8);\n",1)
open('fm.hip','w').write(g)
f=open('fm_body.h').read()
f=f.replace('#include "common.h"\n', '#include "common.h"\n'+blk+'\n',1)
f=f.replace("    float4 v[UF];\n    float w[UF];\n","    TS_MARK(1);\n    float4 v[UF];\n    float w[UF];\n",1)
f=f.replace("  // combine the field phases (lanes that own the same columns)\n","  TS_MARK(2);\n  // combine the field phases (lanes that own the same columns)\n",1)
f=f.replace("  float tot = pair + lin;\n","  float tot = pair + lin;\n  TS_MARK(3);\n",1)
open('fm_body.h','w').write(f)
P
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -DTS_TRACE -I. -I../../../recommendation_amd/csrc -c fm.hip -o fm_ts.o
cd ../../../recommendation_amd/csrc
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scripts/exp/libmi_tstrace.so $(ls *.o | grep -v '^fm.o$') ../../scripts/exp/l1/fm_ts.o -ldl
echo built
