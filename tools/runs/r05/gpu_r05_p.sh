#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cu_state.py -x -q -s -k "engine" 2>&1 | grep -v amdgpu.ids | tail -4
for sn in 1 4; do
ENGINE_STREAMS=$sn timeout 1800 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48,128 --no-check 2>gpurun_out/m.err | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['chains'].items(): print('streams $sn k', k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
"
done
tail -2 gpurun_out/m.err
rm -rf /tmp/prof_e; mkdir -p /tmp/prof_e
ENGINE_STREAMS=4 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_e -- python tools/cu_state_walk.py --mode engine --states 1500 --k 128 --no-check > gpurun_out/p_prof.json 2>gpurun_out/p_prof.err
f=$(find /tmp/prof_e -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in rows]
ev.sort()
# the engine's part: from the first cs_seg kernel to the last
segs=[e for e in ev if 'cs_seg' in e[2]]
t0,t1=segs[0][0],max(e[1] for e in segs)
busy=0; cur_s=cur_e=None
for s,e,_ in segs:
    if cur_e is None or s>cur_e:
        if cur_e is not None: busy+=cur_e-cur_s
        cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print('engine span ms',(t1-t0)/1e6,'device busy (union) frac',busy/(t1-t0),'launches',len(segs))
by=collections.defaultdict(lambda:[0,0])
for s,e,n in segs:
    n=n.split('(')[0]; by[n][0]+=1; by[n][1]+=e-s
tot=sum(v[1] for v in by.values())
print('sum of kernel time / span', tot/(t1-t0))
for n,v in sorted(by.items(),key=lambda x:-x[1][1]): print('%-60s %6d %8.1f us avg %5.1f%%'%(n[:60],v[0],v[1]/v[0]/1e3,100*v[1]/tot))
PY
